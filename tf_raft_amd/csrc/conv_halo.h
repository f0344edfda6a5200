// Halo-tiled implicit-GEMM stride-1 'same' convolution on exact-fp32 MFMA (v_mfma_f32_16x16x4_f32)
// for gfx950 -- the update-block convolution of the RAFT loop.
//
// A workgroup owns a 2-D tile of TH x 16 output pixels of one image and BN = 64*TN output channels.
// K is walked CHUNK-major: for each 32-channel chunk of the input the (TH+KH-1) x (16+KW-1) HALO
// tile is fetched ONCE into LDS (out-of-image pixels arrive as zeros through the buffer bounds
// check) and serves all KH*KW taps -- every tap is just a shifted window of the same LDS tile.
// Compared with one LDS tile per (tap, chunk) this divides the global->LDS traffic, the LDS writes
// and the barriers by ~KH*KW (3x3: 9 x 112 = 1008 pixel rows per chunk become 9 x 18 = 162).
//
//   * MFMA rows = the 16 pixels of one tile row (lane l supplies A[pixel l&15][k = l>>4]); a wave
//     holds all TH tile rows x TN 16-channel column blocks: TH x TN accumulators of 4 registers.
//     4 waves split the BN channels (wave w owns channels [w*16*TN, (w+1)*16*TN) of the tile).
//   * TH = 7 exists because 448x512 inputs give 56 x 64 feature maps: 7 x 16 tiles cut B*56*64
//     pixels into B*32 tiles, a power of two that fills the 256 CUs evenly (a 128-pixel tile cannot).
//   * A fragments: ds_read_b128 from LDS pixel rows of 40 floats (conflict-free for the hardware's
//     b128 lane groups, tools/bank_check.py); lane group G = l>>4 takes k-quad 4*kk+G of the chunk.
//   * B fragments (weights) do not go through LDS at all: the packed layout [tap][k/4][npad][4]
//     (include/raft_hip.h) is already fragment-shaped, so each lane loads its 16-byte k-quad straight
//     from L2 one round ahead of use; the four waves read disjoint channel ranges.
//   * software pipeline: fragments of round q+1 are fetched while the MFMAs of round q issue; the
//     next chunk's halo tile is written to the other LDS buffer and the (single) barrier per chunk is
//     taken before the chunk's last round, so the first fragments of the next chunk are prefetched
//     under that last round.  Global loads of chunk c+2 are issued as soon as chunk c+1 has left the
//     staging registers.
//   * epilogues are branch-free (buffer stores with out-of-range offsets for masked pixels /
//     channels): on gfx9 vmcnt counts stores, so a compiler-placed vmcnt(0) in a divergent store
//     block would serialise them.
// Accumulation order: chunk-major, tap-minor, k ascending within the lane-group permutation.
#pragma once
#include <stdlib.h>
#include <type_traits>

#include "conv_mfma.h"

// Template parameters beyond the tile shape:
//   STRIDE 1 | 2   output stride (encoder down-sampling convolutions, TF 'SAME'/'valid' padding via p.pt/p.pl);
//                  for 1x1 kernels the stride is applied while staging (only the needed pixels are fetched)
//   PRE            the input is relu(x * scale[b][c] + shift[b][c]) applied while staging the halo tile
//                  (= instance norm + relu of the producer fused into the consumer); padding stays zero
//   STATS          also write per-tile (sum, sum of squares) of the raw output per channel (instance-norm moments)
//   STEM           7x7 stride-2 stem on a 4-channel-padded image: K chunk c = kernel row c, k = (kx, ch) of the
//                  7 x 4 input window (+ 4 zero columns); KH = KW = 1 and cin = 7 * 32 in this mode
//   DEEP           weight (B) fragments are fetched TWO rounds ahead through a 4-slot register ring instead of one
//                  round ahead: a layer that leaves one workgroup per CU (N = 128 at B = 4: one wave per SIMD, nothing
//                  to switch to) otherwise waits out part of every L2 round trip -- a round is only TH x 32 MFMA cycles
template <int KH, int KW, int TH, int TN, int EPI, int STRIDE = 1, int PRE = 0, int STATS = 0, int STEM = 0, int DEEP = 0>
__global__ void __launch_bounds__(256) conv_halo_kernel(ConvArgs p) {
    constexpr int TW = 16, NKK = 2;
    constexpr int TAPS = KH * KW, R = TAPS * NKK;
    constexpr int LSTEP = (TAPS == 1 && !STEM) ? STRIDE : 1;   // input pixels between neighbouring halo pixels
    constexpr int FSTEP = (TAPS == 1) ? 1 : STRIDE;            // halo pixels between neighbouring output pixels
    constexpr int LDA = (FSTEP == 2) ? 36 : 40;                // conflict-free pixel-row stride (tools/bank_check.py)
    constexpr int HH = (TH - 1) * FSTEP + KH, HWP = (TW - 1) * FSTEP + KW, HP = HH * HWP;   // halo tile
    constexpr int NA = (HP * 8 + 255) / 256;                   // float4 chunks per thread per K chunk
    constexpr int A_BUF = HP * LDA + 8;                        // + one dummy 16-byte slot for padding items
    constexpr int BN = 64 * TN;
    static_assert(!STEM || (TAPS == 1 && STRIDE == 2 && !PRE), "stem mode: 1x1 addressing, stride 2");
    __shared__ __attribute__((aligned(16))) float smem[2 * A_BUF];

    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int G = lane >> 4, LR = lane & 15;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int ntn = p.npad / BN;
    const int M = p.B * p.H * p.W;                     // output pixels
    const int Hi = p.Hi, Wi = p.Wi;
    const int Min = p.B * Hi * Wi;                     // input pixels

    // XCD-aware remap (bijective for any grid size): logical tiles of one XCD are contiguous, and the
    // N tiles of one pixel tile are neighbours, so they share the halo tile in that XCD's L2
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt = bid / ntn, nt = bid - mt * ntn;
    const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y, b = mt / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = nt * BN;
    const int cin = p.c0 + p.c1;
    const int nch = cin >> 5;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.a0, 0, (int)((((long)Min - 1) * p.lda0 + (STEM ? 4 : p.c0)) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.c1 ? p.a1 : p.a0), 0, p.c1 ? (int)((((long)Min - 1) * p.lda1 + p.c1) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.wp, 0, (int)((long)TAPS * cin * p.npad * 4), 0x00020000);

    // ---- halo staging assignment: item = (halo pixel, 16-byte channel quad)
    int pix[NA];          // input pixel index of the item's halo pixel, or -1 (outside the image / padding item)
    int lds_off[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int item = tid + 256 * i;
        const int hp = item >> 3, c4 = item & 7;
        const int hy = hp / HWP, hx = hp - hy * HWP;
        lds_off[i] = hp < HP ? hp * LDA + c4 * 4 : HP * LDA;
        if (STEM) {
            // pix = input pixel of kernel column c4 in kernel row 0; the row is advanced per chunk in gload()
            const int yy = (y0 + hy) * 2 - p.pt, xx = (x0 + hx) * 2 - p.pl + c4;
            const bool ok = (hp < HP) & (c4 < 7) & ((unsigned)xx < (unsigned)Wi);
            pix[i] = ok ? (b * Hi + yy) * Wi + xx : -(1 << 30);    // yy may still be out of range: checked per chunk
        } else {
            const int yy = y0 * STRIDE - p.pt + hy * LSTEP, xx = x0 * STRIDE - p.pl + hx * LSTEP;
            const bool ok = (hp < HP) & ((unsigned)yy < (unsigned)Hi) & ((unsigned)xx < (unsigned)Wi);
            pix[i] = ok ? (b * Hi + yy) * Wi + xx : -1;
        }
    }
    f32x4 ra[NA], pre_sc, pre_sh;
    auto gload = [&](int c) {
        if (STEM) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int hp = (tid + 256 * i) >> 3;
                const int yy = (y0 + hp / HWP) * 2 - p.pt + c;                   // kernel row c
                const bool ok = (pix[i] > -(1 << 29)) & ((unsigned)yy < (unsigned)Hi);
                ra[i] = raft_buffer_load_f4(rs0, ok ? (unsigned)((pix[i] + c * Wi) * 16) : RAFT_OOB);
            }
            return;
        }
        const int ch = c * 32;
        const bool first = ch < p.c0;
        const int ld = first ? p.lda0 : p.lda1;
        const int chl = (first ? ch : ch - p.c0) + (tid & 7) * 4;
        if (first) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                ra[i] = raft_buffer_load_f4(rs0, pix[i] >= 0 ? (unsigned)((pix[i] * ld + chl) * 4) : RAFT_OOB);
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                ra[i] = raft_buffer_load_f4(rs1, pix[i] >= 0 ? (unsigned)((pix[i] * ld + chl) * 4) : RAFT_OOB);
        }
        if (PRE) {
            pre_sc = *(const f32x4 *)(p.pre_scale + (long)b * cin + ch + (tid & 7) * 4);
            pre_sh = *(const f32x4 *)(p.pre_shift + (long)b * cin + ch + (tid & 7) * 4);
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
            *(f32x4 *)(smem + buf * A_BUF + lds_off[i]) = v;
        }
    };

    // ---- fragment fetch
    f32x4 fa[2][TH], fb[DEEP ? 4 : 2][TN];
    const int a_lane = LR * FSTEP * LDA + G * 4;                          // + window shift + kk*16
    const unsigned b_lane = (unsigned)(((G * p.npad) + n0 + wn * 16 * TN + LR) * 16);   // bytes
    auto frag_a = [&](int buf, int q, f32x4 *a) {
        const int t = q / NKK, kk = q - t * NKK;
        const float *base = smem + buf * A_BUF + a_lane + ((t / KW) * HWP + (t % KW)) * LDA + kk * 16;
#pragma unroll
        for (int i = 0; i < TH; ++i) a[i] = *(const f32x4 *)(base + i * FSTEP * HWP * LDA);
    };
    auto frag_b = [&](int c, int q, f32x4 *bf) {
        const int t = q / NKK, kk = q - t * NKK;
        const unsigned row = (unsigned)((t * (cin >> 2) + c * 8 + kk * 4) * p.npad) * 16u;   // wave-uniform bytes
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bf[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)(b_lane + j * 256), (int)row, 0));
    };

    f32x4 acc[TH][TN];
#pragma unroll
    for (int i = 0; i < TH; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue resources (set up here so that the epilogue inputs can be prefetched)
    const int w0 = (EPI == EPI_GRU_ZR) ? p.hid : p.nvalid;          // valid columns of o0
    const int w1 = (EPI == EPI_GRU_ZR) ? p.nvalid - p.hid : 0;      // valid columns of o1
    const __amdgpu_buffer_rsrc_t ro0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.o0, 0, (int)((((long)M - 1) * p.ldo0 + w0) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro1 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(w1 > 0 ? p.o1 : p.o0), 0, w1 > 0 ? (int)((((long)M - 1) * p.ldo1 + w1) * 4) : 0, 0x00020000);
    const bool has_e0 = EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q || EPI == EPI_RES, has_e1 = EPI == EPI_GRU_Q;
    const int we = (EPI == EPI_GRU_ZR) ? p.hid : p.nvalid;
    const __amdgpu_buffer_rsrc_t re0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_e0 ? (const void *)p.e0 : (const void *)p.o0), 0,
        has_e0 ? (int)((((long)M - 1) * p.lde0 + we) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t re1 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_e1 ? (const void *)p.e1 : (const void *)p.o0), 0,
        has_e1 ? (int)((((long)M - 1) * p.lde1 + we) * 4) : 0, 0x00020000);
    auto bstore = [](float v, __amdgpu_buffer_rsrc_t r, unsigned off) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0);
    };
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
    };
    // gate / residual inputs of the epilogue are fetched BEFORE the K loop (a layer that leaves one workgroup per CU
    // has nothing to hide their latency behind afterwards); masked pixels / channels read through RAFT_OOB -> 0
    float pe0[TH][TN][4], pe1[TH][TN][4];
    if (EPI == EPI_RES || EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q) {
#pragma unroll
        for (int i = 0; i < TH; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn * TN + j) * 16 + LR;
                const bool nok = n < p.nvalid;
                const bool isz = n < p.hid;
                const unsigned ne = (EPI == EPI_GRU_ZR) ? (unsigned)(isz ? n : n - p.hid) : (unsigned)n;
                const bool want0 = (EPI == EPI_GRU_ZR) ? (nok & !isz) : nok;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int yy = y0 + i, xx = x0 + 4 * G + r;
                    const bool mokr = (yy < p.H) & (xx < p.W);
                    const unsigned m = (unsigned)((b * p.H + yy) * p.W + xx);
                    pe0[i][j][r] = bload(re0, (want0 & mokr) ? (m * p.lde0 + ne) * 4u : RAFT_OOB);
                    if (EPI == EPI_GRU_Q) pe1[i][j][r] = bload(re1, (nok & mokr) ? (m * p.lde1 + n) * 4u : RAFT_OOB);
                }
            }
    }
    gload(0);
    frag_b(0, 0, fb[0]);
    if (p.init) {
        // accumulators start from a precomputed partial sum (lane owns channel n; register r of
        // accumulator (i, j) is pixel (y0 + i, x0 + 4G + r)); issued behind the first tile's loads
        const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
            (void *)p.init, 0, (int)((((long)M - 1) * p.ldi + p.nvalid) * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < TH; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn * TN + j) * 16 + LR;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int yy = y0 + i, xx = x0 + 4 * G + r;
                    const bool ok = (yy < p.H) & (xx < p.W) & (n < p.nvalid);
                    const unsigned m = (unsigned)((b * p.H + yy) * p.W + xx);
                    acc[i][j][r] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(ri, ok ? (int)((m * p.ldi + n) * 4u) : (int)RAFT_OOB, 0, 0));
                }
            }
    }
    lstore(0);
    raft_barrier_lds();
    if (nch > 1) gload(1);
    frag_a(0, 0, fa[0]);
    if constexpr (DEEP) {
        // global round g = c * R + q uses ring slot g & 3 (R is even, so two chunks advance the ring by a multiple of
        // 4: the slot of every round is a compile-time constant of (chunk parity, q))
        static_assert(R % 2 == 0 && R >= 2, "ring indexing needs an even number of rounds per chunk");
        frag_b(0, 1, fb[1]);
        auto chunk = [&](auto parity, int c) {
            constexpr int P = decltype(parity)::value;
            const bool more = c + 1 < nch;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int cur = q & 1;
                constexpr int RB = P * R;
                if (q + 1 < R)
                    frag_a(P, q + 1, fa[cur ^ 1]);
                else if (more)
                    frag_a(P ^ 1, 0, fa[cur ^ 1]);
                if (q + 2 < R)
                    frag_b(c, q + 2, fb[(RB + q + 2) & 3]);
                else if (more)
                    frag_b(c + 1, q + 2 - R, fb[(RB + q + 2) & 3]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < TH; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][i][r], fb[(RB + q) & 3][j][r], acc[i][j], 0, 0, 0);
                if (q == R - 2) {
                    if (more) {
                        lstore(P ^ 1);
                        if (c + 2 < nch) gload(c + 2);
                    }
                    raft_barrier_lds();
                }
            }
        };
        int c = 0;
        for (; c + 1 < nch; c += 2) {
            chunk(std::integral_constant<int, 0>{}, c);
            chunk(std::integral_constant<int, 1>{}, c + 1);
        }
        if (c < nch) chunk(std::integral_constant<int, 0>{}, c);
    } else
    for (int c = 0; c < nch; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < nch;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int cur = q & 1;
            if (q + 1 < R) {
                frag_a(buf, q + 1, fa[cur ^ 1]);
                frag_b(c, q + 1, fb[cur ^ 1]);
            } else if (more) {
                frag_a(buf ^ 1, 0, fa[cur ^ 1]);
                frag_b(c + 1, 0, fb[cur ^ 1]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < TH; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][i][r], fb[cur][j][r], acc[i][j], 0, 0, 0);
            if (q == R - 2) {
                if (more) {
                    lstore(buf ^ 1);
                    if (c + 2 < nch) gload(c + 2);
                }
                raft_barrier_lds();
            }
        }
    }

    // ---- epilogue: lane owns channel n; accumulator (i, j)[r] is pixel (y0 + i, x0 + 4G + r)
    float biasv[TN], s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        biasv[j] = p.bias[n0 + (wn * TN + j) * 16 + LR];   // bias has npad entries
        s1[j] = 0.f;
        s2[j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < TH; ++i) {
        unsigned mrow[4];
        bool mok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int yy = y0 + i, xx = x0 + 4 * G + r;
            mok[r] = (yy < p.H) & (xx < p.W);
            mrow[r] = (unsigned)((b * p.H + yy) * p.W + xx);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 16 + LR;
            const bool nok = n < p.nvalid;
            const float bias = biasv[j];
            if (EPI == EPI_LINEAR || EPI == EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r] + bias;
                    if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                    v *= p.scale;
                    if (STATS && mok[r]) {
                        s1[j] += v;
                        s2[j] = fmaf(v, v, s2[j]);
                    }
                    bstore(v, ro0, (nok & mok[r]) ? (mrow[r] * p.ldo0 + n) * 4u : RAFT_OOB);
                }
            } else if (EPI == EPI_RES) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    bstore(fmaxf(pe0[i][j][r] + fmaxf(acc[i][j][r] + bias, 0.f), 0.f), ro0,
                           (nok & mok[r]) ? (mrow[r] * p.ldo0 + n) * 4u : RAFT_OOB);
            } else if (EPI == EPI_GRU_ZR) {
                const bool isz = n < p.hid;
                const unsigned nh = (unsigned)(isz ? n : n - p.hid);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float g = raft_sigmoid(acc[i][j][r] + bias);
                    const bool ok = nok & mok[r];
                    bstore(g, ro0, (ok & isz) ? (mrow[r] * p.ldo0 + nh) * 4u : RAFT_OOB);
                    bstore(g * pe0[i][j][r], ro1, (ok & !isz) ? (mrow[r] * p.ldo1 + nh) * 4u : RAFT_OOB);
                }
            } else {   // EPI_GRU_Q
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float q = raft_tanh(acc[i][j][r] + bias);
                    bstore((1.0f - pe1[i][j][r]) * pe0[i][j][r] + pe1[i][j][r] * q, ro0,
                           (nok & mok[r]) ? (mrow[r] * p.ldo0 + n) * 4u : RAFT_OOB);
                }
            }
        }
    }
    if (STATS) {
        // per-tile moments of the raw output: lanes LR, LR+16, LR+32, LR+48 hold the same channel
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float a = s1[j], q = s2[j];
            a += __shfl_xor(a, 16, 64);
            q += __shfl_xor(q, 16, 64);
            a += __shfl_xor(a, 32, 64);
            q += __shfl_xor(q, 32, 64);
            const int n = n0 + (wn * TN + j) * 16 + LR;
            if (G == 0) *(float2 *)(p.stats + ((long)mt * p.npad + n) * 2) = make_float2(a, q);
        }
    }
}

// ---- per-kernel-size launchers (one translation unit each: conv_halo_<KH><KW>.hip) -------------
int raft_launch_conv_halo_1x1(const ConvArgs &a, int th, int tn, int epi, hipStream_t s);
int raft_launch_conv_halo_3x3(const ConvArgs &a, int th, int tn, int epi, hipStream_t s);
int raft_launch_conv_halo_1x5(const ConvArgs &a, int th, int tn, int epi, hipStream_t s);
int raft_launch_conv_halo_5x1(const ConvArgs &a, int th, int tn, int epi, hipStream_t s);

// Deep weight prefetch (DEEP = 1) for the single-column-block tiles: measured +2..4 % on every update-block layer at
// B = 4 (profiles/r03j_conv_bench_deep*.txt), same register occupancy.
static inline bool raft_conv_deep(const ConvArgs &, int, int tn, int) {
    return tn == 1;
}

template <int KH, int KW, int EPI>
static int raft_launch_conv_halo_tile(const ConvArgs &a, int th, int tn, hipStream_t s) {
    const int tiles = a.B * ((a.H + th - 1) / th) * ((a.W + 15) / 16);
    const int grid = tiles * (a.npad / (64 * tn));
    const int key = th * 10 + tn + (raft_conv_deep(a, th, tn, grid) ? 100 : 0);
    switch (key) {
        case 141: conv_halo_kernel<KH, KW, 4, 1, EPI, 1, 0, 0, 0, 1><<<grid, 256, 0, s>>>(a); break;
        case 171: conv_halo_kernel<KH, KW, 7, 1, EPI, 1, 0, 0, 0, 1><<<grid, 256, 0, s>>>(a); break;
        case 181: conv_halo_kernel<KH, KW, 8, 1, EPI, 1, 0, 0, 0, 1><<<grid, 256, 0, s>>>(a); break;
        case 41: conv_halo_kernel<KH, KW, 4, 1, EPI><<<grid, 256, 0, s>>>(a); break;
        case 42: conv_halo_kernel<KH, KW, 4, 2, EPI><<<grid, 256, 0, s>>>(a); break;
        case 71: conv_halo_kernel<KH, KW, 7, 1, EPI><<<grid, 256, 0, s>>>(a); break;
        case 72: conv_halo_kernel<KH, KW, 7, 2, EPI><<<grid, 256, 0, s>>>(a); break;
        case 81: conv_halo_kernel<KH, KW, 8, 1, EPI><<<grid, 256, 0, s>>>(a); break;
        case 82: conv_halo_kernel<KH, KW, 8, 2, EPI><<<grid, 256, 0, s>>>(a); break;
        default: return RAFT_E_UNSUPPORTED;
    }
    return raft_launch_status();
}

template <int KH, int KW>
static int raft_launch_conv_halo_epi(const ConvArgs &a, int th, int tn, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_LINEAR: return raft_launch_conv_halo_tile<KH, KW, EPI_LINEAR>(a, th, tn, s);
        case EPI_RELU: return raft_launch_conv_halo_tile<KH, KW, EPI_RELU>(a, th, tn, s);
        case EPI_GRU_ZR: return raft_launch_conv_halo_tile<KH, KW, EPI_GRU_ZR>(a, th, tn, s);
        case EPI_GRU_Q: return raft_launch_conv_halo_tile<KH, KW, EPI_GRU_Q>(a, th, tn, s);
    }
    return RAFT_E_UNSUPPORTED;
}
