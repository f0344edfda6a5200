// Winograd F(4x4, 3x3) convolution launcher (kernel: conv_wino4.h).
#include "conv_wino4.h"

int raft_launch_conv_wino4(const ConvArgs &a, int epi, hipStream_t s, int decide_npad, int ks_hint) {
    if (a.c0 <= 0 || a.c0 % 16 || a.c1 < 0 || a.c1 % 16 || a.npad <= 0 || a.npad % 64) return RAFT_E_UNSUPPORTED;
    if (a.lda0 % 4 || (a.c1 && a.lda1 % 4)) return RAFT_E_ALIGN;
    if (!raft_aligned16(a.a0) || !raft_aligned16(a.wp) || (a.c1 && !raft_aligned16(a.a1))) return RAFT_E_ALIGN;
    if (a.init || (a.Hi && (a.Hi != a.H || a.Wi != a.W))) return RAFT_E_UNSUPPORTED;
    if ((a.pre_scale || a.stats) && (epi != EPI_LINEAR || a.stats == nullptr || a.c1 != 0)) return RAFT_E_UNSUPPORTED;
    if (a.pre_scale && a.pre_shift == nullptr) return RAFT_E_NULL;
    if (epi == EPI_RES && a.e0 == nullptr) return RAFT_E_NULL;
    {   // 32-bit buffer offsets: every operand must span < 2 GiB
        const int64_t M = (int64_t)a.B * a.H * a.W, lim = (int64_t)1 << 31;
        if (((M - 1) * a.lda0 + a.c0) * 4 >= lim || (a.c1 && ((M - 1) * a.lda1 + a.c1) * 4 >= lim)) return RAFT_E_UNSUPPORTED;
        if (M * a.ldo0 * 4 >= lim || (int64_t)36 * (a.c0 + a.c1) * a.npad * 4 >= lim) return RAFT_E_UNSUPPORTED;
        if (epi == EPI_RES && M * a.lde0 * 4 >= lim) return RAFT_E_UNSUPPORTED;
    }
    // Two row blocks (8 x 64 pixels) x 64 channels per workgroup; when that leaves fewer workgroups than ~3/4 of the chip's CUs
    // (fewer than 128: convc2 84, conv 56, fh1 112 at 4 pairs; conv 112 at 8 -- profiles/r07i_wino4_bench.txt), one row block per workgroup with K split between two wave sets instead --
    // twice the workgroups, half the K loop each (RAFT_WINO4_KS = 1 / 2 overrides).  The split needs an even number of
    // 16-channel chunks in each source.
    const int nt = a.npad / 64;
    const int grid1 = a.B * ((a.H + 7) / 8) * ((a.W + 63) / 64) * nt;
    // decide_npad: choose the variant as a layer of that many output channels would (flow_head.conv1 alone must round exactly
    // like its half of the fused flow / mask head: RAFT.predict_step returns the bits of flow_predictions[-1])
    const int grid_decide = decide_npad > 0 ? grid1 / nt * (decide_npad / 64) : grid1;
    const bool ks2_ok = (a.c0 % 32 == 0) && (a.c1 % 32 == 0);
    int ks = raft_opt(RAFT_OPT_WINO4_KS, (ks_hint == 1 || ks_hint == 2) ? ks_hint : (grid_decide * raft_concurrency() < 128 ? 2 : 1));
    if (ks != 2 || !ks2_ok || a.stats) ks = 1;
    if (a.stats) {   // instance-norm encoder: moments of the raw output, optionally the producer's normalisation + relu on the input
        if (a.pre_scale)
            conv_wino4_kernel<EPI_LINEAR, 1, 1, 1><<<grid1, 256, 0, s>>>(a);
        else
            conv_wino4_kernel<EPI_LINEAR, 1, 0, 1><<<grid1, 256, 0, s>>>(a);
        return raft_launch_status();
    }
    if (ks == 2) {
        const int grid = a.B * ((a.H + 3) / 4) * ((a.W + 63) / 64) * nt;
        if (epi == EPI_LINEAR)
            conv_wino4_kernel<EPI_LINEAR, 2><<<grid, 256, 0, s>>>(a);
        else if (epi == EPI_RELU)
            conv_wino4_kernel<EPI_RELU, 2><<<grid, 256, 0, s>>>(a);
        else if (epi == EPI_RES)
            conv_wino4_kernel<EPI_RES, 2><<<grid, 256, 0, s>>>(a);
        else
            return RAFT_E_UNSUPPORTED;
        return raft_launch_status();
    }
    if (epi == EPI_LINEAR)
        conv_wino4_kernel<EPI_LINEAR, 1><<<grid1, 256, 0, s>>>(a);
    else if (epi == EPI_RELU)
        conv_wino4_kernel<EPI_RELU, 1><<<grid1, 256, 0, s>>>(a);
    else if (epi == EPI_RES)
        conv_wino4_kernel<EPI_RES, 1><<<grid1, 256, 0, s>>>(a);
    else
        return RAFT_E_UNSUPPORTED;
    return raft_launch_status();
}
