// 1-D Winograd F(2, 5) / F(4, 5) convolution launcher (kernel: conv_wino1d.h).
#include "conv_wino1d.h"

constexpr bool RAFT_WINO1D_CK2_DEFAULT = true;   // 32 channels per barrier: +2-4 % on the GRU layers (profiles/r03u)

template <int AXIS, int TNW, int CK, int TM, int MO = 2>
static int launch_wino1d(const ConvArgs &a, int epi, int grid, hipStream_t s) {
    switch (epi) {
        case EPI_LINEAR: conv_wino1d_kernel<AXIS, TNW, EPI_LINEAR, CK, TM, MO><<<grid, 256, 0, s>>>(a); break;
        case EPI_RELU: conv_wino1d_kernel<AXIS, TNW, EPI_RELU, CK, TM, MO><<<grid, 256, 0, s>>>(a); break;
        case EPI_GRU_ZR: conv_wino1d_kernel<AXIS, TNW, EPI_GRU_ZR, CK, TM, MO><<<grid, 256, 0, s>>>(a); break;
        case EPI_GRU_Q: conv_wino1d_kernel<AXIS, TNW, EPI_GRU_Q, CK, TM, MO><<<grid, 256, 0, s>>>(a); break;
        default: return RAFT_E_UNSUPPORTED;
    }
    return raft_launch_status();
}

template <int AXIS, int TM>
static int launch_wino1d_tm(const ConvArgs &a, int epi, int grid, int tnw, bool ck2, hipStream_t s) {
    if (ck2) return tnw == 2 ? launch_wino1d<AXIS, 2, 2, TM>(a, epi, grid, s) : launch_wino1d<AXIS, 1, 2, TM>(a, epi, grid, s);
    return tnw == 2 ? launch_wino1d<AXIS, 2, 1, TM>(a, epi, grid, s) : launch_wino1d<AXIS, 1, 1, TM>(a, epi, grid, s);
}

int raft_launch_conv_wino1d(const ConvArgs &a, int kh, int kw, int epi, hipStream_t s, int mo, int tnw_hint) {
    if (mo != 2 && mo != 4) return RAFT_E_UNSUPPORTED;
    if (mo == 4 && (a.c0 % 32 || a.c1 % 32)) return RAFT_E_UNSUPPORTED;
    if (!((kh == 1 && kw == 5) || (kh == 5 && kw == 1))) return RAFT_E_UNSUPPORTED;
    if (a.c0 <= 0 || a.c0 % 16 || a.c1 < 0 || a.c1 % 16 || a.npad <= 0 || a.npad % 32) return RAFT_E_UNSUPPORTED;
    if (a.lda0 % 4 || (a.c1 && a.lda1 % 4)) return RAFT_E_ALIGN;
    if (!raft_aligned16(a.a0) || !raft_aligned16(a.wp) || (a.c1 && !raft_aligned16(a.a1))) return RAFT_E_ALIGN;
    if (a.pre_scale || a.stats || (a.Hi && (a.Hi != a.H || a.Wi != a.W))) return RAFT_E_UNSUPPORTED;
    {   // 32-bit buffer offsets: every operand must span < 2 GiB
        const int64_t M = (int64_t)a.B * a.H * a.W, lim = (int64_t)1 << 31;
        if (((M - 1) * a.lda0 + a.c0) * 4 >= lim || (a.c1 && ((M - 1) * a.lda1 + a.c1) * 4 >= lim)) return RAFT_E_UNSUPPORTED;
        if (M * a.ldo0 * 4 >= lim || (a.o1 && M * a.ldo1 * 4 >= lim) || (a.e0 && M * a.lde0 * 4 >= lim) ||
            (a.e1 && M * a.lde1 * 4 >= lim) || (a.init && M * a.ldi * 4 >= lim))
            return RAFT_E_UNSUPPORTED;
        if ((int64_t)(mo + 4) * (a.c0 + a.c1) * a.npad * 4 >= lim) return RAFT_E_UNSUPPORTED;
    }
    const int axis = kh == 5 ? 1 : 0;
    const int forced = raft_opt(RAFT_OPT_WINO_TNW, tnw_hint);   // tuning / test overrides (raft_set_option), else the caller's hint
    const int tm_forced = raft_opt(RAFT_OPT_WINO1D_TM, 0);
    const bool ck2 = a.c0 % 32 == 0 && a.c1 % 32 == 0 && raft_opt(RAFT_OPT_WINO_CK, RAFT_WINO1D_CK2_DEFAULT ? 2 : 1) == 2;
    auto tiles_of = [&](int tm) {
        return axis == 0 ? a.B * ((a.H + 2 * tm - 1) / (2 * tm)) * ((a.W + 31) / 32)
                         : a.B * ((a.H + 4 * tm - 1) / (4 * tm)) * ((a.W + 15) / 16);
    };
    // 64-channel workgroups (a transformed input feeds two column blocks) wherever the channel count allows; full-height
    // tiles (TM = 2) when they still give about two workgroups per CU, half-height tiles otherwise (gru_q at B = 4:
    // 224 -> 448 workgroups)
    if (mo == 4) {
        // F(4, 5): a workgroup owns 2 rows x 64 columns (1x5) or 8 rows x 16 columns (5x1); 64-channel workgroups while
        // that leaves about two per CU (gru_q at B = 4 stand-alone: 224 workgroups of 64 channels 28.3 us against 31.0 us
        // for 448 of 32 channels, but inside the three-stream loop with the GRU epilogue 34.5 / 38.6 us against 33.5 / 34.3)
        const int tiles = axis == 0 ? a.B * ((a.H + 1) / 2) * ((a.W + 63) / 64) : a.B * ((a.H + 7) / 8) * ((a.W + 15) / 16);
        int tnw = (a.npad % 64 == 0 && (int64_t)tiles * (a.npad / 64) * raft_concurrency() >= 400) ? 2 : 1;
        if (forced == 1 || (forced == 2 && a.npad % 64 == 0)) tnw = forced;
        const int grid = tiles * (a.npad / (32 * tnw));
        if (axis == 0) return tnw == 2 ? launch_wino1d<0, 2, 2, 1, 4>(a, epi, grid, s) : launch_wino1d<0, 1, 2, 1, 4>(a, epi, grid, s);
        return tnw == 2 ? launch_wino1d<1, 2, 2, 1, 4>(a, epi, grid, s) : launch_wino1d<1, 1, 2, 1, 4>(a, epi, grid, s);
    }
    // F(2, 5): 64-channel workgroups only where that still leaves enough of them (a single 448 x 512 pair: 112 against 448
    // workgroups for gru_zr -- 7.90 -> 7.48 ms per forward with the 32-channel ones, profiles/r06b_b1_options.txt)
    int tnw = (a.npad % 64 == 0 && (int64_t)tiles_of(2) * (a.npad / 64) * raft_concurrency() >= 400) ? 2 : 1;
    if (forced == 1 || (forced == 2 && a.npad % 64 == 0)) tnw = forced;
    int tm = (int64_t)tiles_of(2) * (a.npad / (32 * tnw)) * raft_concurrency() >= 400 ? 2 : 1;
    if (tm_forced == 1 || tm_forced == 2) tm = tm_forced;
    const int grid = tiles_of(tm) * (a.npad / (32 * tnw));
    if (axis == 0) return tm == 2 ? launch_wino1d_tm<0, 2>(a, epi, grid, tnw, ck2, s) : launch_wino1d_tm<0, 1>(a, epi, grid, tnw, ck2, s);
    return tm == 2 ? launch_wino1d_tm<1, 2>(a, epi, grid, tnw, ck2, s) : launch_wino1d_tm<1, 1>(a, epi, grid, tnw, ck2, s);
}
