// Winograd F(2x2, 3x3) stride-1 'same' convolution on fp32 MFMA (v_mfma_f32_16x16x4_f32) for gfx950.
//
// Y = A^T [ (G g G^T) (.) (B^T d B) ] A with the standard F(2,3) matrices: a 2x2 output tile costs 16 multiplies per
// (input channel, output channel) instead of 36 -- 2.25x fewer MACs than the direct 3x3 kernel of conv_halo.h, in
// fp32 arithmetic throughout (the input and output transforms are additions, the weight transform G g G^T is done
// once on the host in float64 and rounded to fp32: tf_raft_amd/packing.py).  This is the algorithm cuDNN runs for the
// reference's Conv2D on a GPU; it is NOT bit-identical to an fmaf chain (measured deviation of a layer ~1e-6 relative,
// tests/test_gpu_kernels.py), so the direct kernel stays selectable (RAFT_CONV_WINO=0).
//
//   * GEMM view per winograd tap t = (ty, tx) of 16: D_t[tile][n] = sum_k V_t[tile][k] * U_t[k][n]; an MFMA row is
//     one 2x2 output tile, an MFMA row block 16 tiles along x (32 x 2 output pixels).  A workgroup owns 4 x 32 output
//     pixels (two row blocks) x BN = 32*TNW output channels; wave w -> row block w & 1, channel group w >> 1, TNW
//     column blocks of 16 channels: 16 x TNW accumulators of 4 registers (one per tap: the taps are only combined in
//     the epilogue).
//   * K is walked in 16-channel chunks: the 6 x 34 halo tile of the chunk is staged ONCE in LDS (pixel stride 20
//     floats: conflict-free ds_read_b128 for lanes stepping two pixels, tools/bank_check.py; out-of-image pixels are
//     zeros through the buffer bounds check).  Lane (tile m, k-quad G) reads the rows of its 4 x 4 patch as b128 =
//     four channels, applies B^T . B with 8 vector adds per tap row + 4 per tap, and feeds the four channels as the
//     four k-steps of the tap's MFMAs -- the transformed tile never exists in memory.
//   * weights: the packed layout [tap][k/4][npad][4] of the direct kernel with 16 "taps"; fragments come straight from
//     L2 two taps ahead through a 4-slot register ring (the two waves of a channel group share them in L1).
//   * epilogue: A^T . A over the 16 accumulators of a column block (lane-local: 24 adds per tile), bias, relu, scale.
#pragma once
#include <stdlib.h>

#include "conv_mfma.h"

#ifndef RAFT_WINO_PF1
#define RAFT_WINO_PF1 3
#endif
#ifndef RAFT_WINO_PF2
#define RAFT_WINO_PF2 2
#endif
#ifndef RAFT_WINO_ABL
#define RAFT_WINO_ABL 0   // tools/ablate/wino_abl.hip builds this header with pieces of the main loop switched off
#endif

// PRE / STATS / EPI_RES serve the encoders exactly as in conv_halo.h: PRE applies relu(x * scale[b][c] + shift[b][c])
// while the halo tile is staged (instance norm + relu of the producer), STATS writes per-(workgroup, row block)
// (sum, sum of squares) of the raw output per channel [2 * pixel tile + rb][npad][2], EPI_RES is the ResBlock tail.
// SB = 1 pins the weight-fragment prefetch two taps ahead with a scheduling barrier (see the main loop); it costs
// registers (two workgroups per CU instead of three at TNW = 1), so the launcher uses it where two resident workgroups
// per CU cover the grid anyway.
// CK = 16-channel chunks staged per barrier (1 or 2): CK = 2 halves the barriers of the TNW = 1 kernels, whose chunks
// are only 64 MFMAs per wave (it needs 12 more staging registers, which the TNW = 2 kernels do not have).
// KS = 2 (TNW = 1, CK = 2 only): split K inside the workgroup.  512 threads; waves 0-3 take the even 16-channel chunks,
// waves 4-7 the odd ones (the two chunks of a staged stage, side by side instead of one after the other) and the two
// partial accumulator sets are added through LDS before the epilogue.  For launches that cannot fill the chip any other
// way: a 16 x 16 x 4 MFMA tile is 16 Winograd tiles x 16 channels, so `conv` (128 outputs) of ONE 448 x 512 pair is 448
// wave-tasks for 1024 SIMDs whatever the tiling, and each task's K loop is the kernel's duration.
template <int TNW, int EPI, int PRE = 0, int STATS = 0, int SB = 0, int CK = 1, int KS = 1>
__global__ void __launch_bounds__(256 * KS, KS == 2 ? 1 : ((TNW == 1 && !SB && !PRE && CK == 1) ? 3 : 2)) conv_wino_kernel(ConvArgs p) {
    static_assert(KS == 1 || (KS == 2 && TNW == 1 && (CK == 2 || CK == 4) && !PRE && !STATS), "split-K variant: TNW = 1, CK = 2 / 4, plain epilogues");
    constexpr int NTHR = 256 * KS;
    constexpr int RB = 2, TW = 32, TH = 2 * RB;
    constexpr int HH = TH + 2, HWP = TW + 2, HP = HH * HWP;   // 6 x 34 halo pixels
    constexpr int LDA = 16 * CK + 4;                           // floats per halo pixel in LDS (20 / 36: conflict-free)
    constexpr int QS = 4 * CK;                                 // 16-byte channel quads per halo pixel per stage
    constexpr int NA = (HP * QS + NTHR - 1) / NTHR;            // float4 items per thread per stage
    constexpr int A_BUF = HP * LDA + 4;                        // + one dummy 16-byte slot for padding items
    constexpr int BN = 32 * TNW;
    // weight fragments are fetched PF taps ahead into a ring of NR: two taps at TNW = 2 (16 MFMAs) and in the three-
    // workgroups-per-CU variant (large grids: other waves cover the round trip, and a third tap costs it 4 spilled
    // registers); three taps in the other TNW = 1 variants, where a tap is only four MFMAs per wave and two taps do not
    // cover an L2 round trip when few waves share the SIMD (small batches)
    constexpr bool OCC3 = TNW == 1 && !SB && !PRE && CK == 1;  // the three-workgroups-per-CU launch bound above
    constexpr int PF = (TNW == 2 || OCC3) ? RAFT_WINO_PF2 : RAFT_WINO_PF1;
    constexpr int NR = PF < 4 ? 4 : 8;
    static_assert(EPI == EPI_LINEAR || EPI == EPI_RELU || EPI == EPI_RES || EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q,
                  "winograd kernel: linear / relu / residual / GRU gate epilogues");
    __shared__ __attribute__((aligned(16))) float smem[2 * A_BUF];

    const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3;
    const int ks = KS == 2 ? tid >> 8 : 0;                     // which 16-channel chunk of a stage this wave multiplies
    const int G = lane >> 4, LR = lane & 15;
    const int rb = w & 1, cg = w >> 1;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int ntn = p.npad / BN;
    const int M = p.B * p.H * p.W;

    int bid = blockIdx.x;   // XCD-aware remap (see conv_halo.h)
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt = bid / ntn, nt = bid - mt * ntn;
    const int tx0 = mt % tiles_x, ty0 = (mt / tiles_x) % tiles_y, b = mt / (tiles_x * tiles_y);
    const int y0 = ty0 * TH, x0 = tx0 * TW;
    const int n0 = nt * BN;
    const int cin = p.c0 + p.c1;
    const int nst = cin / (16 * CK);                   // stages (barriers)

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.a0, 0, (int)((((long)M - 1) * p.lda0 + p.c0) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.c1 ? p.a1 : p.a0), 0, p.c1 ? (int)((((long)M - 1) * p.lda1 + p.c1) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.wp, 0, (int)((long)16 * cin * p.npad * 4), 0x00020000);

    // ---- halo staging: item = (halo pixel, 16-byte channel quad of the chunk)
    int pix[NA], lds_off[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int item = tid + NTHR * i;
        const int hp = item / QS, c4 = item % QS;
        const int hy = hp / HWP, hx = hp - hy * HWP;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool ok = (hp < HP) & ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W);
        pix[i] = ok ? (b * p.H + yy) * p.W + xx : -1;
        lds_off[i] = hp < HP ? hp * (LDA / 4) + c4 : HP * (LDA / 4);   // 16-byte units
    }
    f32x4 ra[NA], pre_sc, pre_sh;
    auto gload = [&](int c) {
        const int ch = c * 16 * CK;
        if (PRE) {
            pre_sc = *(const f32x4 *)(p.pre_scale + (long)b * cin + ch + (tid % QS) * 4);
            pre_sh = *(const f32x4 *)(p.pre_shift + (long)b * cin + ch + (tid % QS) * 4);
        }
        const bool first = ch < p.c0;
        const int ld = first ? p.lda0 : p.lda1;
        const int chl = (first ? ch : ch - p.c0) + (tid % QS) * 4;
        if (first) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                ra[i] = raft_buffer_load_f4(rs0, pix[i] >= 0 ? (unsigned)((pix[i] * ld + chl) * 4) : RAFT_OOB);
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                ra[i] = raft_buffer_load_f4(rs1, pix[i] >= 0 ? (unsigned)((pix[i] * ld + chl) * 4) : RAFT_OOB);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            f32x4 v = ra[i];
            if (PRE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = pix[i] >= 0 ? fmaxf(fmaf(v[e], pre_sc[e], pre_sh[e]), 0.f) : 0.f;
            }
            *(f32x4 *)(smem + buf * A_BUF + lds_off[i] * 4) = v;
        }
    };

    // ---- fragments
    const int a_lane = ((2 * rb) * HWP + 2 * LR) * LDA + G * 4;          // patch origin of tile LR in row block rb
    const unsigned b_lane = (unsigned)(((G * p.npad) + n0 + cg * 16 * TNW + LR) * 16);   // bytes
    auto patch_row = [&](int buf, int sub, int r, f32x4 *d) {              // d[j] = patch(r, j), j = 0..3, 16-ch chunk `sub`
        if ((RAFT_WINO_ABL & 2) && !(buf == 0 && sub == 0 && r != 1 && r != 3)) return;
        const float *base = smem + buf * A_BUF + a_lane + r * HWP * LDA + sub * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = *(const f32x4 *)(base + j * LDA);
    };
    f32x4 fb[NR][TNW];
    auto frag_b = [&](int c, int t, f32x4 *bf) {
        const unsigned row = (unsigned)((t * (cin >> 2) + c * 4) * p.npad) * 16u;   // wave-uniform bytes
#pragma unroll
        for (int j = 0; j < TNW; ++j)
            bf[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)(b_lane + j * 256), (int)row, 0));
    };

    f32x4 acc[16][TNW];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload(0);
#pragma unroll
    for (int t = 0; t < PF; ++t) frag_b(ks, t, fb[t]);
    lstore(0);
    raft_barrier_lds();
    if (nst > 1) gload(1);
    // patch rows: dA holds row 0, later row 3; dB row 2; dC row 1.  Every LDS read is issued one tap row ahead of its
    // use (row 1 under the MFMAs of tap row 0, row 3 under tap row 1, the NEXT chunk's rows 0 and 2 under tap row 3).
    f32x4 dA[4], dB[4], dC[4];
    patch_row(0, ks, 0, dA);
    patch_row(0, ks, 2, dB);
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < nst;
#pragma unroll
        for (int sub0 = 0; sub0 < CK / KS; ++sub0) {
            const int sub = sub0 * KS + ks;                       // KS = 2: wave set ks takes the chunks ks, ks + 2, ..
            const int c = st * CK + sub;                          // 16-channel chunk index (weights)
            const bool last_sub = sub0 == CK / KS - 1;
            const bool more_c = more || !last_sub;                // another 16-channel chunk follows
#pragma unroll
            for (int ty = 0; ty < 4; ++ty) {
                // B^T over the patch rows: ty 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
                f32x4 R[4];
                if (ty == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) R[j] = dA[j] - dB[j];
                    patch_row(buf, sub, 1, dC);
                } else if (ty == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) R[j] = dC[j] + dB[j];
                    patch_row(buf, sub, 3, dA);
                } else if (ty == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) R[j] = dB[j] - dC[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) R[j] = dC[j] - dA[j];
                    if (last_sub) {
                        // every LDS read of this stage has been issued and consumed: hand the other buffer over
                        if (more && !(RAFT_WINO_ABL & 4)) {
                            lstore(buf ^ 1);
                            if (st + 2 < nst) gload(st + 2);
                        }
                        raft_barrier_lds();
                        if (more) {
                            patch_row(buf ^ 1, ks, 0, dA);
                            patch_row(buf ^ 1, ks, 2, dB);
                        }
                    } else {
                        patch_row(buf, sub + KS, 0, dA);
                        patch_row(buf, sub + KS, 2, dB);
                    }
                }
#pragma unroll
                for (int tx = 0; tx < 4; ++tx) {
                    const int t = ty * 4 + tx;
                    f32x4 V;
                    if (RAFT_WINO_ABL & 16) V = R[tx];
                    else if (tx == 0) V = R[0] - R[2];
                    else if (tx == 1) V = R[1] + R[2];
                    else if (tx == 2) V = R[2] - R[1];
                    else V = R[1] - R[3];
                    if (!(RAFT_WINO_ABL & 1)) {
                        if (t + PF < 16)
                            frag_b(c, t + PF, fb[(t + PF) & (NR - 1)]);
                        else if (more_c)
                            frag_b(c + KS, t + PF - 16, fb[(t + PF) & (NR - 1)]);
                    }
                    // keep the weight fetch of tap t + PF HERE: left alone, the scheduler sinks it next to its use to
                    // save registers and every tap then waits out an L2 round trip (seen in the ISA: load, s_waitcnt, mfma)
                    if (SB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < TNW; ++j)
                            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[e], fb[t & (NR - 1)][j][e], acc[t][j], 0, 0, 0);
                }
            }
        }
    }

    if (KS == 2) {
        // add the odd-chunk partial sums (waves 4-7) to the even-chunk ones (waves 0-3): 8 taps at a time through the
        // staging buffers, [tap][wave][lane] float4 (conflict-free); waves 4-7 are done afterwards
        static_assert(KS == 1 || 8 * 4 * 64 * 4 <= 2 * A_BUF, "split-K reduction buffer must fit the staging buffers");
        f32x4 *red = (f32x4 *)smem;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            raft_barrier_lds();                                   // staging buffers / previous half no longer read
            if (ks == 1) {
#pragma unroll
                for (int t = 0; t < 8; ++t) red[(t * 4 + w) * 64 + lane] = acc[8 * half + t][0];
            }
            raft_barrier_lds();
            if (ks == 0) {
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[8 * half + t][0] += red[(t * 4 + w) * 64 + lane];
            }
        }
        if (ks == 1) return;
    }

    // ---- epilogue: lane owns channel n; register r of an accumulator is tile m = 4G + r of the row block
    // GRU gates (the 3x3 ConvGRU of SmallRAFT, reference update.py:17-35) as in conv_halo.h: GRU_ZR writes z to o0 and
    // r * h to o1 (h = e0), GRU_Q writes (1 - z) h + z tanh(.) (h = e0, z = e1)
    constexpr bool GRU = EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q, HAS_E0 = EPI == EPI_RES || GRU;
    const int w0 = (EPI == EPI_GRU_ZR) ? p.hid : p.nvalid;          // valid columns of o0
    const int w1 = (EPI == EPI_GRU_ZR) ? p.nvalid - p.hid : 0;      // valid columns of o1
    const int we = (EPI == EPI_GRU_ZR) ? p.hid : p.nvalid;
    const __amdgpu_buffer_rsrc_t ro0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.o0, 0, (int)((((long)M - 1) * p.ldo0 + w0) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro1 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(w1 > 0 ? p.o1 : p.o0), 0, w1 > 0 ? (int)((((long)M - 1) * p.ldo1 + w1) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t re0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(HAS_E0 ? (const void *)p.e0 : (const void *)p.o0), 0,
        HAS_E0 ? (int)((((long)M - 1) * p.lde0 + we) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t re1 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(EPI == EPI_GRU_Q ? (const void *)p.e1 : (const void *)p.o0), 0,
        EPI == EPI_GRU_Q ? (int)((((long)M - 1) * p.lde1 + we) * 4) : 0, 0x00020000);
    // Addresses: one lane base per tensor (pixel (yy, x0 + 8G), channel n; RAFT_OOB when the lane's channel takes no
    // part) + a wave-uniform element offset in the instruction's scalar operand; elements outside the image (only in
    // tiles cut by the border) get the out-of-range bit.
    auto bstore = [](float v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, soff, 0);
    };
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
    };
    const bool interior = (y0 + TH <= p.H) & (x0 + TW <= p.W);   // wave-uniform
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int n = n0 + (cg * TNW + j) * 16 + LR;
        const bool nok = n < p.nvalid;
        const float bias = p.bias[n];                         // bias has npad entries
        const bool isz = n < p.hid;
        const unsigned nh = (unsigned)((EPI == EPI_GRU_ZR && !isz) ? n - p.hid : n);
        float s1 = 0.f, s2 = 0.f;
        // A^T over the tap rows: T[i][tx], i = 0: M0 + M1 + M2, i = 1: M1 - M2 - M3
        f32x4 T[2][4];
#pragma unroll
        for (int tx = 0; tx < 4; ++tx) {
            T[0][tx] = (acc[tx][j] + acc[4 + tx][j]) + acc[8 + tx][j];
            T[1][tx] = (acc[4 + tx][j] - acc[8 + tx][j]) - acc[12 + tx][j];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 ya = (T[i][0] + T[i][1]) + T[i][2], yb = (T[i][1] - T[i][2]) - T[i][3];
            const int yy = y0 + 2 * rb + i, xb = x0 + 8 * G;    // the lane's tiles 4G + r: pixels xb + 2 r + jx of row yy
            const unsigned pix0 = (unsigned)((b * p.H + yy) * p.W + xb);
            unsigned dead[4][2];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jx = 0; jx < 2; ++jx)
                    dead[r][jx] = (interior | ((yy < p.H) & (xb + 2 * r + jx < p.W))) ? 0u : RAFT_OOB;
            const unsigned bo0 = (EPI == EPI_GRU_ZR ? (nok & isz) : nok) ? (pix0 * p.ldo0 + nh) * 4u : RAFT_OOB;
            const unsigned bo1 = (EPI == EPI_GRU_ZR && nok && !isz) ? (pix0 * p.ldo1 + nh) * 4u : RAFT_OOB;
            const unsigned be0 = (HAS_E0 && (EPI == EPI_GRU_ZR ? (nok & !isz) : nok)) ? (pix0 * p.lde0 + nh) * 4u : RAFT_OOB;
            const unsigned be1 = (EPI == EPI_GRU_Q && nok) ? (pix0 * p.lde1 + n) * 4u : RAFT_OOB;
            float xv[4][2], zv[4][2];
            if (HAS_E0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int jx = 0; jx < 2; ++jx) {
                        xv[r][jx] = bload(re0, be0 | dead[r][jx], (2 * r + jx) * p.lde0 * 4);
                        if (EPI == EPI_GRU_Q) zv[r][jx] = bload(re1, be1 | dead[r][jx], (2 * r + jx) * p.lde1 * 4);
                    }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int jx = 0; jx < 2; ++jx) {
                    float v = (jx ? yb[r] : ya[r]) + bias;
                    const int so0 = (2 * r + jx) * p.ldo0 * 4;
                    if (EPI == EPI_GRU_ZR) {
                        const float g = raft_sigmoid(v);
                        bstore(g, ro0, bo0 | dead[r][jx], so0);
                        bstore(g * xv[r][jx], ro1, bo1 | dead[r][jx], (2 * r + jx) * p.ldo1 * 4);
                        continue;
                    }
                    if (EPI == EPI_GRU_Q) {
                        v = (1.0f - zv[r][jx]) * xv[r][jx] + zv[r][jx] * raft_tanh(v);
                    } else if (EPI == EPI_RES) {
                        v = fmaxf(xv[r][jx] + fmaxf(v, 0.f), 0.f);
                    } else {
                        if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                        v *= p.scale;
                    }
                    if (STATS && dead[r][jx] == 0u) {
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                    if ((RAFT_WINO_ABL & 8) && v != 12345.678f) continue;
                    bstore(v, ro0, bo0 | dead[r][jx], so0);
                }
            }
        }
        if (STATS) {   // lanes LR, LR+16, LR+32, LR+48 hold the same channel
            s1 += __shfl_xor(s1, 16, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (G == 0) *(float2 *)(p.stats + ((long)(2 * mt + rb) * p.npad + n) * 2) = make_float2(s1, s2);
        }
    }
}

// launcher (conv_wino.hip); `a.wp` holds the winograd-transformed weights packed as a 4x4-tap kernel
// epi: EPI_LINEAR / EPI_RELU / EPI_RES; a.pre_scale != NULL selects PRE, a.stats != NULL selects STATS (the encoder
// combinations: LINEAR + STATS (+ PRE), RELU, RES).  Stats entries per image: 2 * ceil(H/4) * ceil(W/32).
// decide_npad > 0: choose the variant (channel width, K split, stage depth) as a layer of that many output channels would --
// flow_head.conv1 alone must round exactly like its half of the fused flow / mask head (RAFT.predict_step == flow_predictions[-1])
int raft_launch_conv_wino(const ConvArgs &a, int epi, hipStream_t s, int decide_npad = 0);
