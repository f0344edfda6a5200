// 1-D Winograd F(2, 5) for the 1x5 / 5x1 convolutions of the SepConvGRU on fp32 MFMA (v_mfma_f32_16x16x4_f32), gfx950.
//
// Two neighbouring outputs along the convolution axis need 6 multiplies per (input channel, output channel) instead of
// 10 -- 1.67x fewer MACs than the direct kernel of conv_halo.h.  Interpolation points {0, +-1, +-1/2, inf}; with the
// rows of B^T / G rescaled by powers of two (exact) the transforms are
//
//   B'^T d :  v0 = d0 - 5 d2 + 4 d4           G' g : u0 = g0
//             v1 = -(d1 + d2) + 4 (d3 + d4)          u1 = (g0 + g1 + g2 + g3 + g4) / 6
//             v2 =  (d1 - d2) - 4 (d3 - d4)          u2 = (g0 - g1 + g2 - g3 + g4) / 6
//             v3 =  (d3 - d1) + 2 (d4 - d2)          u3 = -(4/3) g0 - (2/3) g1 - (1/3) g2 - (1/6) g3 - (1/12) g4
//             v4 = -(d3 - d1) + 2 (d4 - d2)          u4 = -(4/3) g0 + (2/3) g1 - (1/3) g2 + (1/6) g3 - (1/12) g4
//             v5 = d1 - 5 d3 + 4 d5                  u5 = g4 / 4
//   A^T m  :  y0 = m0 + m1 + m2 + m3 + m4,   y1 = (m1 - m2) + (m3 - m4) / 2 + m5
//
// (G' g is evaluated once on the host in float64: tf_raft_amd/packing.py winograd1d_kernel).  In fp32 the result
// deviates from the float64 convolution 1.3-1.6x as much as the direct fp32 kernel does (K = 256; tests).
//
//   * an MFMA row is one output PAIR; a row block is 16 pairs.  AXIS 0 (1x5): a row block = 32 consecutive pixels of
//     one image row (pairs along x), a workgroup owns 4 rows x 32 columns; AXIS 1 (5x1): a row block = 16 consecutive
//     columns of one row pair (pairs along y), a workgroup owns 8 rows x 16 columns.  Either way 2 TM row blocks x
//     BN = 32*TNW channels (TM = 2 above); wave w -> row blocks TM (w & 1) + {0 .. TM-1}, channel group w >> 1 with TNW
//     column blocks: TM x 6 x TNW accumulators of 4 registers.  A weight fragment feeds both row blocks, a transformed input both
//     column blocks.
//   * K in 16-channel chunks: halo tile staged once in LDS (AXIS 0: 4 x 36 pixels at 20 floats, AXIS 1: 12 x 16 pixels
//     at 24 floats: conflict-free ds_read_b128 for lanes stepping 2 / 1 pixels, tools/bank_check.py); a lane reads the
//     6 inputs of its pair as b128 (4 channels), transforms them with 6 adds + 8 fmas per channel, and the four
//     channels are the four k-steps of the tap's MFMAs.
//   * weights: packed layout [tap][k/4][npad][4] with 6 taps, fetched from L2 two taps ahead (3-slot ring).
//   * epilogues: the GRU gate epilogues of conv_halo.h (accumulator preload `init` becomes an addend after A^T).
//
// MO = 4 selects F(4, 5): four neighbouring outputs from 8 multiplies (2.5x fewer MACs than direct, 1.5x fewer than
// F(2, 5)), points {0, +-1, +-1/2, +-2, inf}; fp32 deviation from the float64 convolution about 2x the direct kernel's
// (K = 256).  Rows of B^T rescaled by 4, 4, 4, 2, 2, 4, 4, 4 (G' rows by the inverse, exact):
//
//   v0 = 4 (d6 - d0) + 21 (d2 - d4)                        v7 = 4 (d7 - d1) + 21 (d3 - d5)
//   v1 =  4 ((d1 + d2) + (d5 + d6)) - 17 (d3 + d4)         v2 = 4 ((d2 - d1) + (d6 - d5)) + 17 (d3 - d4)
//   v3 =  pA + 2 qA,  v4 = 2 qA - pA     pA = 4 d1 - 5 d3 + d5,  qA = 4 d2 - 5 d4 + d6
//   v5 = 2 pB + qB,   v6 = qB - 2 pB     pB = d1 - 5 d3 + 4 d5,  qB = d2 - 5 d4 + 4 d6
//   y0 = m0 + s12 + s34 + s56        y1 = d12 + d34 / 2 + 2 d56       (s.. = m_a + m_b, d.. = m_a - m_b)
//   y2 = s12 + s34 / 4 + 4 s56       y3 = d12 + d34 / 8 + 8 d56 + m7
//
// An MFMA row is then one output QUAD: AXIS 0 row block = 64 consecutive pixels of an image row (lanes step 4 pixels:
// halo pixels at 32 floats + 8 floats of padding every 4 pixels keeps the b128 reads conflict-free), AXIS 1 row block
// = 16 columns x 4 rows.  32-channel stages only (CK = 2), 4-slot weight ring.
#pragma once
#include <stdlib.h>

#include "conv_mfma.h"

// Diagnostics only (tools/ablate/wino1d_abl.hip): bit 1 = weight fragments fetched once, 2 = input transform (LDS reads)
// once, 4 = no halo staging inside the loop, 8 = no epilogue loads and a single store, 16 = MFMAs of tap 0 only.
// 0 in the library build.
#ifndef RAFT_WINO1D_ABL
#define RAFT_WINO1D_ABL 0
#endif

// Epilogue shared by the kernels of this file: A^T m, bias / context addend, activation or GRU gate, stores.  Lane owns
// channel n; register r of an accumulator is output group m = 4G + r of row block TM * rbp + i.
template <int AXIS, int TNW, int EPI, int TM, int MO>
__device__ __forceinline__ void wino1d_epilogue(const ConvArgs &p, f32x4 (&acc)[TM][MO + 4][TNW], int rbp, int cg, int G, int LR,
                                                int b, int y0, int x0, int n0, int M) {
    constexpr int TILE_H = AXIS == 0 ? 2 * TM : 2 * MO * TM, TILE_W = AXIS == 0 ? 16 * MO : 16;
    const int w0 = (EPI == EPI_GRU_ZR) ? p.hid : p.nvalid;          // valid columns of o0
    const int w1 = (EPI == EPI_GRU_ZR) ? p.nvalid - p.hid : 0;      // valid columns of o1
    const __amdgpu_buffer_rsrc_t ro0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.o0, 0, (int)((((long)M - 1) * p.ldo0 + w0) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro1 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(w1 > 0 ? p.o1 : p.o0), 0, w1 > 0 ? (int)((((long)M - 1) * p.ldo1 + w1) * 4) : 0, 0x00020000);
    const bool has_e0 = EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q, has_e1 = EPI == EPI_GRU_Q;
    const int we = (EPI == EPI_GRU_ZR) ? p.hid : p.nvalid;
    const __amdgpu_buffer_rsrc_t re0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_e0 ? (const void *)p.e0 : (const void *)p.o0), 0,
        has_e0 ? (int)((((long)M - 1) * p.lde0 + we) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t re1 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(has_e1 ? (const void *)p.e1 : (const void *)p.o0), 0,
        has_e1 ? (int)((((long)M - 1) * p.lde1 + we) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.init ? (const void *)p.init : (const void *)p.o0), 0,
        p.init ? (int)((((long)M - 1) * p.ldi + p.nvalid) * 4) : 0, 0x00020000);
    // Addresses: one lane base per tensor (first pixel of the lane's groups, channel n; RAFT_OOB when the lane's channel
    // takes no part) + a wave-uniform element offset in the instruction's scalar operand; elements outside the image
    // (only in tiles cut by the border) get the out-of-range bit.  A null `init` has a zero-sized descriptor: loads give 0.
    auto bstore = [](float v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, soff, 0);
    };
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
    };
    const bool interior = (y0 + TILE_H <= p.H) & (x0 + TILE_W <= p.W);   // wave-uniform
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rb = TM * rbp + i;
        // the lane's 4 groups m = 4G + r start at pixel (yb, xb); element (r, jx) lies es(r, jx) pixels further
        const int yb = AXIS == 0 ? y0 + rb : y0 + MO * rb;
        const int xb = AXIS == 0 ? x0 + MO * 4 * G : x0 + 4 * G;
        const unsigned pix0 = (unsigned)((b * p.H + yb) * p.W + xb);
        unsigned dead[4][MO];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int jx = 0; jx < MO; ++jx) {
                const int yy = AXIS == 0 ? yb : yb + jx;
                const int xx = AXIS == 0 ? xb + MO * r + jx : xb + r;
                dead[r][jx] = (interior | ((yy < p.H) & (xx < p.W))) ? 0u : RAFT_OOB;
            }
        auto es = [&](int r, int jx) { return AXIS == 0 ? MO * r + jx : jx * p.W + r; };
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int n = n0 + (cg * TNW + j) * 16 + LR;
            const bool nok = n < p.nvalid;
            const float bias = p.bias[n];                         // bias has npad entries
            const bool isz = n < p.hid;
            const unsigned nh = (unsigned)((EPI == EPI_GRU_ZR && !isz) ? n - p.hid : n);
            f32x4 y[MO];                                          // A^T m
            if constexpr (MO == 2) {
                y[0] = ((acc[i][0][j] + acc[i][1][j]) + (acc[i][2][j] + acc[i][3][j])) + acc[i][4][j];
                y[1] = ((acc[i][1][j] - acc[i][2][j]) + 0.5f * (acc[i][3][j] - acc[i][4][j])) + acc[i][5][j];
            } else {
                const f32x4 s12 = acc[i][1][j] + acc[i][2][j], d12 = acc[i][1][j] - acc[i][2][j];
                const f32x4 s34 = acc[i][3][j] + acc[i][4][j], d34 = acc[i][3][j] - acc[i][4][j];
                const f32x4 s56 = acc[i][5][j] + acc[i][6][j], d56 = acc[i][5][j] - acc[i][6][j];
                y[0] = (acc[i][0][j] + s12) + (s34 + s56);
                y[1] = (d12 + 0.5f * d34) + 2.0f * d56;
                y[2] = (s12 + 0.25f * s34) + 4.0f * s56;
                y[3] = ((d12 + 0.125f * d34) + 8.0f * d56) + acc[i][7][j];
            }
            const unsigned bi = nok ? (pix0 * p.ldi + n) * 4u : RAFT_OOB;                        // init
            const unsigned bo0 = (EPI == EPI_GRU_ZR ? (nok & isz) : nok) ? (pix0 * p.ldo0 + nh) * 4u : RAFT_OOB;
            const unsigned bo1 = (EPI == EPI_GRU_ZR && nok && !isz) ? (pix0 * p.ldo1 + nh) * 4u : RAFT_OOB;
            const unsigned be0 = (EPI == EPI_GRU_ZR ? (nok & !isz) : (EPI == EPI_GRU_Q && nok)) ? (pix0 * p.lde0 + nh) * 4u : RAFT_OOB;
            const unsigned be1 = (EPI == EPI_GRU_Q && nok) ? (pix0 * p.lde1 + n) * 4u : RAFT_OOB;
            float iv[4][MO], hv[4][MO], zv[4][MO];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jx = 0; jx < MO; ++jx) {
                    if (RAFT_WINO1D_ABL & 8) {
                        iv[r][jx] = hv[r][jx] = zv[r][jx] = 0.5f;
                        continue;
                    }
                    iv[r][jx] = bload(ri, bi | dead[r][jx], es(r, jx) * p.ldi * 4);
                    if (EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q) hv[r][jx] = bload(re0, be0 | dead[r][jx], es(r, jx) * p.lde0 * 4);
                    if (EPI == EPI_GRU_Q) zv[r][jx] = bload(re1, be1 | dead[r][jx], es(r, jx) * p.lde1 * 4);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jx = 0; jx < MO; ++jx) {
                    const float v = (y[jx][r] + iv[r][jx]) + bias;
                    const int so0 = es(r, jx) * p.ldo0 * 4;
                    if ((RAFT_WINO1D_ABL & 8) && v != 12345.678f) continue;
                    if (EPI == EPI_LINEAR || EPI == EPI_RELU) {
                        bstore((EPI == EPI_RELU ? fmaxf(v, 0.f) : v) * p.scale, ro0, bo0 | dead[r][jx], so0);
                    } else if (EPI == EPI_GRU_ZR) {
                        const float g = raft_sigmoid(v);
                        bstore(g, ro0, bo0 | dead[r][jx], so0);
                        bstore(g * hv[r][jx], ro1, bo1 | dead[r][jx], es(r, jx) * p.ldo1 * 4);
                    } else {
                        const float q = raft_tanh(v);
                        bstore((1.0f - zv[r][jx]) * hv[r][jx] + zv[r][jx] * q, ro0, bo0 | dead[r][jx], so0);
                    }
                }
        }
    }
}

// CK = 16-channel chunks staged per barrier (1 or 2).  TM = row blocks per wave (2, or 1: half-height workgroup tiles
// -- twice the workgroups for layers such as gru_q (N = 128) that otherwise leave under one workgroup per CU).
template <int AXIS, int TNW, int EPI, int CK = 1, int TM = 2, int MO = 2>
__global__ void __launch_bounds__(256, 2) conv_wino1d_kernel(ConvArgs p) {
    static_assert(MO == 2 || (MO == 4 && CK == 2), "F(4,5) stages 32 channels per barrier");
    constexpr int NT = MO + 4;                                  // taps = inputs of one output group
    // weight fragments are fetched PF taps ahead into a ring (NT % RING == 0): two taps where a tap is at least 8 MFMAs per
    // wave, four where it is only 4 (TM = TNW = 1: the under-filled launches of small batches, where nothing else on the
    // SIMD covers the L2 round trip -- see conv_wino.h)
    constexpr int PF = (TM * TNW == 1) ? 4 : 2;
    constexpr int RING = PF == 2 ? (NT == 6 ? 3 : 4) : NT;
    constexpr int TILE_H = AXIS == 0 ? 2 * TM : 2 * MO * TM, TILE_W = AXIS == 0 ? 16 * MO : 16;
    constexpr int HH = AXIS == 0 ? 2 * TM : TILE_H + 4, HWP = AXIS == 0 ? TILE_W + 4 : 16, HP = HH * HWP;   // halo tile
    constexpr bool SWZ = AXIS == 0 && MO == 4;                  // lanes step 4 pixels: pad 8 floats every 4 pixels
    constexpr int LDA = SWZ ? 32 : AXIS == 0 ? (CK == 1 ? 20 : 36) : (CK == 1 ? 24 : 40);   // floats per halo pixel in LDS
    constexpr int ROW = SWZ ? HWP * LDA + (HWP / 4) * 8 : HWP * LDA;   // floats per halo row
    constexpr int QS = 4 * CK;                                  // 16-byte channel quads per halo pixel per stage
    constexpr int NA = (HP * QS + 255) / 256;
    constexpr int A_BUF = HH * ROW + 4;
    constexpr int BN = 32 * TNW;
    static_assert(EPI == EPI_LINEAR || EPI == EPI_RELU || EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q, "epilogue");
    __shared__ __attribute__((aligned(16))) float smem[2 * A_BUF];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int G = lane >> 4, LR = lane & 15;
    const int rbp = w & 1, cg = w >> 1;
    const int tiles_x = (p.W + TILE_W - 1) / TILE_W, tiles_y = (p.H + TILE_H - 1) / TILE_H;
    const int ntn = p.npad / BN;
    const int M = p.B * p.H * p.W;

    int bid = blockIdx.x;   // XCD-aware remap (see conv_halo.h)
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt = bid / ntn, nt = bid - mt * ntn;
    const int tx0 = mt % tiles_x, ty0 = (mt / tiles_x) % tiles_y, b = mt / (tiles_x * tiles_y);
    const int y0 = ty0 * TILE_H, x0 = tx0 * TILE_W;
    const int n0 = nt * BN;
    const int cin = p.c0 + p.c1;
    const int nst = cin / (16 * CK);                            // stages (barriers)

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.a0, 0, (int)((((long)M - 1) * p.lda0 + p.c0) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.c1 ? p.a1 : p.a0), 0, p.c1 ? (int)((((long)M - 1) * p.lda1 + p.c1) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.wp, 0, (int)((long)NT * cin * p.npad * 4), 0x00020000);

    // ---- halo staging: item = (halo pixel, 16-byte channel quad of the chunk)
    int pix[NA], lds_off[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int item = tid + 256 * i;
        const int hp = item / QS, c4 = item % QS;
        const int hy = hp / HWP, hx = hp - hy * HWP;
        const int yy = y0 + hy - (AXIS == 1 ? 2 : 0), xx = x0 + hx - (AXIS == 0 ? 2 : 0);
        const bool ok = (hp < HP) & ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W);
        pix[i] = ok ? (b * p.H + yy) * p.W + xx : -1;
        lds_off[i] = hp < HP ? (hy * ROW + hx * LDA) / 4 + (SWZ ? (hx >> 2) * 2 : 0) + c4 : HH * ROW / 4;   // 16-byte units
    }
    f32x4 ra[NA];
    auto gload = [&](int c) {
        const int ch = c * 16 * CK;
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
        for (int i = 0; i < NA; ++i) *(f32x4 *)(smem + buf * A_BUF + lds_off[i] * 4) = ra[i];
    };

    // ---- fragments.  Row block i of this wave: rb = TM rbp + i.
    //   AXIS 0: group LR of image row y0 + rb: inputs at halo (rb, MO LR + k)     AXIS 1: group = rows MO rb .. MO rb +
    //   MO - 1 of column x0 + LR: inputs at halo (MO rb + k, LR),  k = 0 .. NT-1
    auto a_lane = [&](int i) {
        const int rb = TM * rbp + i;
        return (AXIS == 0 ? rb * ROW + MO * LR * LDA + (SWZ ? LR * 8 : 0) : (MO * rb * HWP + LR) * LDA) + G * 4;
    };
    auto koff = [](int k) { return AXIS == 0 ? k * LDA + (SWZ ? (k >> 2) * 8 : 0) : k * HWP * LDA; };
    const unsigned b_lane = (unsigned)(((G * p.npad) + n0 + cg * 16 * TNW + LR) * 16);   // bytes
    f32x4 fb[RING][TNW];
    auto frag_b = [&](int c, int t, f32x4 *bf) {
        const unsigned row = (unsigned)((t * (cin >> 2) + c * 4) * p.npad) * 16u;   // wave-uniform bytes
#pragma unroll
        for (int j = 0; j < TNW; ++j)
            bf[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)(b_lane + j * 256), (int)row, 0));
    };
    // B'^T d for one output group, four channels at a time
    auto transform = [&](int buf, int sub, int i, f32x4 *V) {
        const float *base = smem + buf * A_BUF + a_lane(i) + sub * 16;
        f32x4 d[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) d[k] = *(const f32x4 *)(base + koff(k));
        if constexpr (MO == 2) {
            const f32x4 s12 = d[1] + d[2], m12 = d[1] - d[2], s34 = d[3] + d[4], m34 = d[3] - d[4];
            const f32x4 a = d[3] - d[1], bb = d[4] - d[2];
            V[0] = (d[0] + 4.0f * d[4]) - 5.0f * d[2];
            V[1] = 4.0f * s34 - s12;
            V[2] = m12 - 4.0f * m34;
            V[3] = a + 2.0f * bb;
            V[4] = 2.0f * bb - a;
            V[5] = (d[1] + 4.0f * d[5]) - 5.0f * d[3];
        } else {
            const f32x4 s34 = d[3] + d[4], m34 = d[3] - d[4];
            const f32x4 pa = (4.0f * d[1] + d[5]) - 5.0f * d[3], qa = (4.0f * d[2] + d[6]) - 5.0f * d[4];
            const f32x4 pb = (d[1] + 4.0f * d[5]) - 5.0f * d[3], qb = (d[2] + 4.0f * d[6]) - 5.0f * d[4];
            V[0] = 4.0f * (d[6] - d[0]) + 21.0f * (d[2] - d[4]);
            V[1] = 4.0f * ((d[1] + d[2]) + (d[5] + d[6])) - 17.0f * s34;
            V[2] = 4.0f * ((d[2] - d[1]) + (d[6] - d[5])) + 17.0f * m34;
            V[3] = pa + 2.0f * qa;
            V[4] = 2.0f * qa - pa;
            V[5] = 2.0f * pb + qb;
            V[6] = qb - 2.0f * pb;
            V[7] = 4.0f * (d[7] - d[1]) + 21.0f * (d[3] - d[5]);
        }
    };

    f32x4 acc[TM][NT][TNW];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[i][t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload(0);
#pragma unroll
    for (int t = 0; t < PF; ++t) frag_b(0, t, fb[t]);
    lstore(0);
    raft_barrier_lds();
    if (nst > 1) gload(1);
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < nst;
#pragma unroll
        for (int sub = 0; sub < CK; ++sub) {
            const int c = st * CK + sub;                          // 16-channel chunk index (weights)
            const bool more_c = more || sub + 1 < CK;
            f32x4 V[TM][NT];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                if (!(RAFT_WINO1D_ABL & 2) || (st == 0 && sub == 0)) transform(buf, sub, i, V[i]);
            if (sub == CK - 1) {
                // every LDS read of this stage has been issued: stage the next one into the other buffer
                if (more && !(RAFT_WINO1D_ABL & 4)) {
                    lstore(buf ^ 1);
                    if (st + 2 < nst) gload(st + 2);
                }
                raft_barrier_lds();
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (RAFT_WINO1D_ABL & 1) {
                } else if (t + PF < NT)
                    frag_b(c, t + PF, fb[(t + PF) % RING]);
                else if (more_c)
                    frag_b(c + 1, t + PF - NT, fb[(t + PF) % RING]);
                __builtin_amdgcn_sched_barrier(0);   // keep the weight fetch PF taps ahead (see conv_wino.h)
                if ((RAFT_WINO1D_ABL & 16) && t != 0) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TNW; ++j)
                            acc[i][t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[i][t][e], fb[t % RING][j][e], acc[i][t][j], 0, 0, 0);
            }
        }
    }

    wino1d_epilogue<AXIS, TNW, EPI, TM, MO>(p, acc, rbp, cg, G, LR, b, y0, x0, n0, M);
}

// launcher (conv_wino1d.hip): kh x kw = 1x5 or 5x1; mo = 2: F(2, 5), `a.wp` holds G' g packed as a 6-tap kernel; mo = 4:
// F(4, 5), 8 taps (needs c0, c1 multiples of 32).  a.init / GRU epilogues as in raft_launch_conv
// tnw_hint = 1 / 2: the caller's choice of 32- / 64-channel workgroups where the launcher would decide by grid size
int raft_launch_conv_wino1d(const ConvArgs &a, int kh, int kw, int epi, hipStream_t s, int mo = 2, int tnw_hint = 0);
