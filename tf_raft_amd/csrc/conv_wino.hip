// Winograd F(2x2, 3x3) convolution launcher (kernel: conv_wino.h).
#include "conv_wino.h"

template <int TNW, int SB, int CK>
static int launch_wino(const ConvArgs &a, int epi, int grid, hipStream_t s) {
    const bool pre = a.pre_scale != nullptr, stats = a.stats != nullptr;
    if (epi == EPI_LINEAR && stats && pre)
        conv_wino_kernel<TNW, EPI_LINEAR, 1, 1, 0, 1><<<grid, 256, 0, s>>>(a);
    else if (epi == EPI_LINEAR && stats)
        conv_wino_kernel<TNW, EPI_LINEAR, 0, 1, SB, CK><<<grid, 256, 0, s>>>(a);
    else if (pre || stats)
        return RAFT_E_UNSUPPORTED;
    else if (epi == EPI_LINEAR)
        conv_wino_kernel<TNW, EPI_LINEAR, 0, 0, SB, CK><<<grid, 256, 0, s>>>(a);
    else if (epi == EPI_RELU)
        conv_wino_kernel<TNW, EPI_RELU, 0, 0, SB, CK><<<grid, 256, 0, s>>>(a);
    else if (epi == EPI_RES)
        conv_wino_kernel<TNW, EPI_RES, 0, 0, SB, CK><<<grid, 256, 0, s>>>(a);
    else if (epi == EPI_GRU_ZR)
        conv_wino_kernel<TNW, EPI_GRU_ZR, 0, 0, SB, CK><<<grid, 256, 0, s>>>(a);
    else if (epi == EPI_GRU_Q)
        conv_wino_kernel<TNW, EPI_GRU_Q, 0, 0, SB, CK><<<grid, 256, 0, s>>>(a);
    else
        return RAFT_E_UNSUPPORTED;
    return raft_launch_status();
}

// split-K inside the workgroup (conv_wino.h, KS = 2): 512 threads, plain epilogues only; CK = 32 or 64 channels per stage
template <int CK>
static int launch_wino_ks2(const ConvArgs &a, int epi, int grid, hipStream_t s) {
    if (epi == EPI_LINEAR)
        conv_wino_kernel<1, EPI_LINEAR, 0, 0, 1, CK, 2><<<grid, 512, 0, s>>>(a);
    else if (epi == EPI_RELU)
        conv_wino_kernel<1, EPI_RELU, 0, 0, 1, CK, 2><<<grid, 512, 0, s>>>(a);
    else if (epi == EPI_RES)
        conv_wino_kernel<1, EPI_RES, 0, 0, 1, CK, 2><<<grid, 512, 0, s>>>(a);
    else if (epi == EPI_GRU_ZR)
        conv_wino_kernel<1, EPI_GRU_ZR, 0, 0, 1, CK, 2><<<grid, 512, 0, s>>>(a);
    else if (epi == EPI_GRU_Q)
        conv_wino_kernel<1, EPI_GRU_Q, 0, 0, 1, CK, 2><<<grid, 512, 0, s>>>(a);
    else
        return RAFT_E_UNSUPPORTED;
    return raft_launch_status();
}

int raft_launch_conv_wino(const ConvArgs &a, int epi, hipStream_t s, int decide_npad) {
    if (a.c0 <= 0 || a.c0 % 16 || a.c1 < 0 || a.c1 % 16 || a.npad <= 0 || a.npad % 32) return RAFT_E_UNSUPPORTED;
    if (a.lda0 % 4 || (a.c1 && a.lda1 % 4)) return RAFT_E_ALIGN;
    if (!raft_aligned16(a.a0) || !raft_aligned16(a.wp) || (a.c1 && !raft_aligned16(a.a1))) return RAFT_E_ALIGN;
    if (a.init || (a.Hi && (a.Hi != a.H || a.Wi != a.W))) return RAFT_E_UNSUPPORTED;
    if ((epi == EPI_GRU_ZR || epi == EPI_GRU_Q) && (a.e0 == nullptr || (epi == EPI_GRU_Q && a.e1 == nullptr))) return RAFT_E_NULL;
    if (epi == EPI_RES && (a.e0 == nullptr || (int64_t)a.B * a.H * a.W * a.lde0 * 4 >= ((int64_t)1 << 31))) return RAFT_E_UNSUPPORTED;
    {   // 32-bit buffer offsets: every operand must span < 2 GiB
        const int64_t M = (int64_t)a.B * a.H * a.W, lim = (int64_t)1 << 31;
        if (((M - 1) * a.lda0 + a.c0) * 4 >= lim || (a.c1 && ((M - 1) * a.lda1 + a.c1) * 4 >= lim)) return RAFT_E_UNSUPPORTED;
        if (M * a.ldo0 * 4 >= lim || (int64_t)16 * (a.c0 + a.c1) * a.npad * 4 >= lim) return RAFT_E_UNSUPPORTED;
    }
    // channel blocks of 64 per workgroup when that still leaves >= 2 workgroups per CU, else blocks of 32
    const int tiles = a.B * ((a.H + 3) / 4) * ((a.W + 31) / 32);
    const int forced = raft_opt(RAFT_OPT_WINO_TNW, 0);   // tuning / test override (raft_set_option)
    const int npad_d = (decide_npad > 0 && decide_npad % 32 == 0) ? decide_npad : a.npad;   // the layer whose launch decides the variant
    int tnw = (a.npad % 64 == 0 && npad_d % 64 == 0 && (int64_t)tiles * (npad_d / 64) * raft_concurrency() >= 512) ? 2 : 1;
    if (forced == 1 || (forced == 2 && a.npad % 64 == 0)) tnw = forced;
    const int grid = tiles * (a.npad / (32 * tnw));
    const int grid_d = tiles * (npad_d / (32 * tnw));
    // pinned weight prefetch (SB): always at TNW = 2; at TNW = 1 only when two workgroups per CU hold the whole grid
    const bool sb = raft_opt(RAFT_OPT_WINO_SB, (tnw == 2 || grid_d <= 512) ? 1 : 0) != 0;   // tuning override: 0 / 1
    // 32 channels per barrier at TNW = 1 when the channel counts allow it (RAFT_WINO_CK = 1 / 2 overrides)
    const bool ck2_ok = a.c0 % 32 == 0 && a.c1 % 32 == 0;
    const bool ck2 = ck2_ok && raft_opt(RAFT_OPT_WINO_CK, grid_d <= 512 ? 2 : 1) == 2;   // 58 KB of LDS: two workgroups per CU
    // fewer wave-tasks than SIMDs (grid * 4 < 1024): split K between two wave sets of a 512-thread workgroup
    // (RAFT_WINO_KS = 1 / 2 overrides)
    const bool plain = a.pre_scale == nullptr && a.stats == nullptr;
    const int ks = raft_opt(RAFT_OPT_WINO_KS, (tnw == 1 && grid_d * raft_concurrency() <= 224) ? 2 : 1);
    if (ks == 2 && tnw == 1 && ck2_ok && plain) {
        // 64 channels per stage where the channel counts allow: the stages of these launches are latency, not work
        const bool ck4 = a.c0 % 64 == 0 && a.c1 % 64 == 0 && raft_opt(RAFT_OPT_WINO_CK, 4) == 4;
        return ck4 ? launch_wino_ks2<4>(a, epi, grid, s) : launch_wino_ks2<2>(a, epi, grid, s);
    }
    if (tnw == 2) return sb ? launch_wino<2, 1, 1>(a, epi, grid, s) : launch_wino<2, 0, 1>(a, epi, grid, s);
    if (ck2) return sb ? launch_wino<1, 1, 2>(a, epi, grid, s) : launch_wino<1, 0, 2>(a, epi, grid, s);
    return sb ? launch_wino<1, 1, 1>(a, epi, grid, s) : launch_wino<1, 0, 1>(a, epi, grid, s);
}
