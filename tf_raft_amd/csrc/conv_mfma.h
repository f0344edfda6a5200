// Implicit-GEMM stride-1 'same' convolution on exact-fp32 MFMA for gfx950.
//
// GEMM view: M = B*H*W output pixels, N = output channels, K = taps x input channels.
// One K-step = (tap t, 32-channel chunk): the A tile is the [BM pixels][32 ch] slice of the NHWC
// input shifted by the tap (zero outside the image), the B tile is [8 k-quads][BN][4] of the
// packed weights (include/raft_hip.h).  Both go global -> registers -> LDS (double-buffered, one
// barrier per K-step) and are consumed as ds_read_b128 fragments.
//
// Two MFMA shapes (template parameter MF):
//   MF = 32: v_mfma_f32_32x32x2_f32, lane l supplies A[row l&31][k = l>>5]; C/D column l&31,
//            rows (r&3) + 8*(r>>2) + 4*(l>>5).   A row stride 36 floats.
//   MF = 16: v_mfma_f32_16x16x4_f32, lane l supplies A[row l&15][k = l>>4]; C/D column l&15,
//            rows 4*(l>>4) + r.                    A row stride 40 floats.
// (row strides chosen so that every hardware ds_read_b128 lane group touches 16 distinct 16-B
// slots, tools/bank_check.py).  MF = 16 exists for its 16-row granularity: 448x512 inputs give
// M = B*3584 = B * 2^9 * 7 pixels, so 112-row tiles (7 x 16) cut M into a power-of-two number
// of workgroups that fills the 256 CUs exactly, which no 32/64/128-row tile can do.
// Per b128 fragment read the lane group G = l / MF takes k-quad (NG*kk + G) of the 32-wide
// K-step (NG = 64 / MF groups), for A and B alike, i.e. a fixed permutation of the K sum.
//
// A-tile loads are UNCONDITIONAL buffer loads: out-of-image taps get an out-of-range offset and
// the buffer bounds check returns 0 (a per-chunk `ok ? load : 0` makes hipcc branch around every
// load with an s_waitcnt in between, which serialises the loads and exposes their latency).
// The same rule shapes the epilogues: the GRU gates read h / z with unconditional (clamped)
// loads issued as one batch per accumulator tile before any arithmetic.
// 4 waves as WGM x WGN; wave tile (BM/WGM) x (BN/WGN) = TM x TN MFMA tiles.
// Workgroup ids are remapped so that consecutive tiles (same pixel tile, neighbouring N tiles)
// run on the same XCD and share its L2 (hardware places workgroup b on XCD b % 8).
#pragma once
#include "common.h"

// Diagnostic ablation (tools/ablate only; always 0 in the library): removes pieces of the main loop
// to price them.  1: no in-loop global loads; 2: + no LDS writes; 3: + no barrier; 4: + no fragment
// reads (MFMA only); 5: full loop, no epilogue stores; 6: loads issued but never written to LDS;
// 7: A loads all out of range (no A traffic).
#ifndef RAFT_ABL
#define RAFT_ABL 0
#endif

enum ConvEpilogue {
    EPI_LINEAR = 0,   // out = (acc + bias) * scale
    EPI_RELU = 1,     // out = relu(acc + bias) * scale
    EPI_GRU_ZR = 2,   // n <  hid: o0 = sigmoid(v)            (z)
                      // n >= hid: o1 = sigmoid(v) * e0[n-hid] (r * h)
    EPI_GRU_Q = 3,    // o0 = (1 - e1) * e0 + e1 * tanh(v)     (h <- (1-z) h + z q), e0 = h, e1 = z
    EPI_RES = 4       // o0 = relu(e0 + relu(acc + bias))     (ResBlock tail, reference extractor.py:41-49)
};

struct ConvArgs {
    const float *a0, *a1;   // input sources (NHWC); channels [0,c0) of a0 then [0,c1) of a1
    int lda0, lda1, c0, c1;
    const float *wp, *bias;
    int B, H, W;
    int npad, nvalid, hid;
    float scale;
    float *o0, *o1;
    int ldo0, ldo1;
    const float *e0, *e1;
    int lde0, lde1;
    // ---- halo-tiled kernel only (conv_halo.h); zero-initialised = plain stride-1 'same' convolution
    int Hi, Wi;                            // input height / width (0: same as the output H, W)
    int pt, pl;                            // zero padding before the first row / column (TF 'SAME')
    const float *pre_scale, *pre_shift;    // PRE: input is relu(x * scale[b][c] + shift[b][c]) (fused instance norm)
    float *stats;                          // STATS: per-tile (sum, sum of squares) of the raw output, [tile][npad][2]
    const float *init;                     // accumulators start from init[pixel * ldi + n] instead of 0 (NULL: 0):
    int ldi;                               //   a precomputed partial convolution (loop-invariant GRU context term)
};

// GRU gate functions on the transcendental unit: exp2 and rcp are each within 1 ulp, the composites within a few
// 1e-7 absolute of libm's -- three orders below the 1e-4 `net` tolerance of the parity tests (the reference's own
// gates are whatever Eigen / cuDNN approximations TensorFlow dispatches to).  libm expf + an IEEE divide cost ~32
// VALU instructions per element, 28 elements per lane per gate kernel.
__device__ __forceinline__ float raft_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float raft_tanh(float x) {
    // 1 - 2 / (exp(2x) + 1): exp2 overflows to +inf -> 1, underflows to 0 -> -1
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.88539008177792681f * x) + 1.0f);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 raft_buffer_load_f4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0));
}

constexpr unsigned RAFT_OOB = 0x80000000u;   // >= any buffer extent we accept (< 2 GiB): load returns 0

template <int MF>
struct MfmaShape;
template <>
struct MfmaShape<32> {
    typedef f32x16 acc_t;
    static constexpr int REGS = 16, LDA = 36;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }
};
template <>
struct MfmaShape<16> {
    typedef f32x4 acc_t;
    static constexpr int REGS = 4, LDA = 40;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int r, int g) { return 4 * g + r; }
};

template <int KH, int KW, int MF, int BM, int BN, int WGM, int WGN, int EPI>
__global__ void __launch_bounds__(256) conv_mfma_kernel(ConvArgs p) {
    typedef MfmaShape<MF> S_;
    typedef typename S_::acc_t acc_t;
    constexpr int BK = 32, LDA = S_::LDA, REGS = S_::REGS;
    constexpr int NG = 64 / MF;                   // lane groups = k-quads per fragment read
    constexpr int NKK = 8 / NG;                   // fragment-read rounds per K-step
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / MF, TN = WTN / MF;
    constexpr int NA = (BM + 31) / 32;            // float4 chunks per thread per K-step (A)
    constexpr int AROWS = NA * 32;
    constexpr int NB = BN / 32;                   // float4 chunks per thread per K-step (B)
    constexpr int A_BUF = AROWS * LDA, B_BUF = BK * BN;
    static_assert(WGM * WGN == 4 && WTM % MF == 0 && WTN % MF == 0 && BN % 32 == 0, "bad conv tile");
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_BUF + B_BUF)];
    float *sA = smem;
    float *sB = smem + 2 * A_BUF;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN, G = lane / MF, LR = lane & (MF - 1);
    const int M = p.B * p.H * p.W;
    const int ntn = p.npad / BN;

    // XCD-aware remap (bijective for any grid size): logical tiles of one XCD are contiguous
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt = bid / ntn, nt = bid - mt * ntn;
    const int m0 = mt * BM;
    const int n0 = nt * BN;
    const int cin = p.c0 + p.c1;
    const int nch = cin / BK;
    const int S = KH * KW * nch;

    // buffer descriptors of the two input sources (extent = last pixel's last used channel)
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.a0, 0, (int)((((long)M - 1) * p.lda0 + p.c0) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.c1 ? p.a1 : p.a0), 0, p.c1 ? (int)((((long)M - 1) * p.lda1 + p.c1) * 4) : 0, 0x00020000);

    // A staging rows of this thread
    const int srow = tid >> 3, sc4 = tid & 7;
    int py[NA], px[NA], pm[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = srow + 32 * i;
        const int m = m0 + r;
        pm[i] = m;
        px[i] = m % p.W;
        py[i] = (m / p.W) % p.H;
        if (m >= M || r >= BM) py[i] = -(1 << 20);   // row beyond the tile / M: every tap is out of range
    }

    f32x4 ra[NA], rb[NB];
    auto gload = [&](int s) {
        const int t = s / nch, cc = s - t * nch;
        const int dy = t / KW - (KH - 1) / 2, dx = t % KW - (KW - 1) / 2;
        const int c = cc * BK;
        const bool first = c < p.c0;
        const int ld = first ? p.lda0 : p.lda1;
        const int ch = (first ? c : c - p.c0) + sc4 * 4;
        unsigned off[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned yy = (unsigned)(py[i] + dy), xx = (unsigned)(px[i] + dx);
            const bool ok = (yy < (unsigned)p.H) & (xx < (unsigned)p.W);   // branch-free
            off[i] = (ok && RAFT_ABL != 7) ? (unsigned)(((pm[i] + dy * p.W + dx) * ld + ch) * 4) : RAFT_OOB;
        }
        if (first) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = raft_buffer_load_f4(rs0, off[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = raft_buffer_load_f4(rs1, off[i]);
        }
        const float *wsrc = p.wp + (((long)t * (cin / 4) + c / 4) * p.npad + n0) * 4;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = tid + 256 * i;
            const int kq = q / BN, j = q - kq * BN;
            rb[i] = *(const f32x4 *)(wsrc + ((long)kq * p.npad + j) * 4);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            *(f32x4 *)(sA + buf * A_BUF + (srow + 32 * i) * LDA + sc4 * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *(f32x4 *)(sB + buf * B_BUF + (tid + 256 * i) * 4) = rb[i];
    };

    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < REGS; ++r) acc[i][j][r] = 0.f;

    // Software pipeline (one barrier per K-step, two LDS buffers, two fragment register sets):
    //   * the fragments of round kk+1 are read from LDS while the MFMAs of round kk issue;
    //   * tile s+1 is written to the other LDS buffer and the barrier is taken BEFORE the last
    //     round of step s, so that round 0 of step s+1 is prefetched under the last round's MFMAs
    //     (every fragment of tile s has been read by then, so buffer s&1 may be overwritten after
    //     the next barrier);
    //   * the global loads of tile s+2 are issued right after tile s+1 left the staging registers.
    f32x4 fa[2][TM], fb[2][TN];
    auto fread = [&](int buf, int kk, f32x4 *a, f32x4 *b) {
        const float *cA = sA + buf * A_BUF + (wm * WTM + LR) * LDA;
        const float *cB = sB + buf * B_BUF + (wn * WTN + LR) * 4;
        const int kq = NG * kk + G;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *(const f32x4 *)(cA + i * MF * LDA + kq * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *(const f32x4 *)(cB + (kq * BN + j * MF) * 4);
    };
    gload(0);
    lstore(0);
    __syncthreads();
    if (S > 1) gload(1);
    fread(0, 0, fa[0], fb[0]);
    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < S;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int cur = kk & 1;
            if (RAFT_ABL != 4) {
                if (kk + 1 < NKK) {
                    fread(buf, kk + 1, fa[cur ^ 1], fb[cur ^ 1]);
                } else if (more) {
                    fread(buf ^ 1, 0, fa[cur ^ 1], fb[cur ^ 1]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = S_::mfma(fa[cur][i][r], fb[cur][j][r], acc[i][j]);
            if (kk == NKK - 2) {
                if (more) {
                    if (RAFT_ABL < 2 || RAFT_ABL == 5 || RAFT_ABL == 7) lstore(buf ^ 1);
                    if (s + 2 < S && (RAFT_ABL == 0 || RAFT_ABL >= 5)) gload(s + 2);
                }
                if (RAFT_ABL < 3 || RAFT_ABL >= 5) __syncthreads();
            }
        }
    }

    // ---- epilogue: lane owns output channel n, REGS rows per accumulator tile.
    // Branch-free: rows >= M / channels >= nvalid get an out-of-range buffer offset (loads return 0,
    // stores are dropped by the bounds check).  On gfx9 vmcnt counts stores as well, so any
    // `s_waitcnt vmcnt(0)` the compiler places in a divergent store block (e.g. for a late bias
    // load) would serialise the stores: there are no such blocks here.
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
    auto bstore = [](float v, __amdgpu_buffer_rsrc_t r, unsigned off) {
        if (RAFT_ABL == 5 && v != 12345.678f) return;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0);
    };
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
    };
    float biasv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) biasv[j] = p.bias[n0 + wn * WTN + j * MF + LR];   // bias has npad entries
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        unsigned mrow[REGS];
        bool mok[REGS];
#pragma unroll
        for (int r = 0; r < REGS; ++r) {
            const int m = m0 + wm * WTM + i * MF + S_::row(r, G);
            mok[r] = m < M;
            mrow[r] = (unsigned)m;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * MF + LR;
            const bool nok = n < p.nvalid;
            const float bias = biasv[j];
            if (EPI == EPI_LINEAR || EPI == EPI_RELU) {
#pragma unroll
                for (int r = 0; r < REGS; ++r) {
                    float v = acc[i][j][r] + bias;
                    if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                    bstore(v * p.scale, ro0, (nok & mok[r]) ? (mrow[r] * p.ldo0 + n) * 4u : RAFT_OOB);
                }
            } else if (EPI == EPI_GRU_ZR) {
                const bool isz = n < p.hid;
                const unsigned nh = (unsigned)(isz ? n : n - p.hid);
                float hv[REGS];
#pragma unroll
                for (int r = 0; r < REGS; ++r)
                    hv[r] = bload(re0, (nok & mok[r] & !isz) ? (mrow[r] * p.lde0 + nh) * 4u : RAFT_OOB);
#pragma unroll
                for (int r = 0; r < REGS; ++r) {
                    const float g = raft_sigmoid(acc[i][j][r] + bias);
                    const bool ok = nok & mok[r];
                    bstore(g, ro0, (ok & isz) ? (mrow[r] * p.ldo0 + nh) * 4u : RAFT_OOB);
                    bstore(g * hv[r], ro1, (ok & !isz) ? (mrow[r] * p.ldo1 + nh) * 4u : RAFT_OOB);
                }
            } else {   // EPI_GRU_Q
                float hv[REGS], zv[REGS];
#pragma unroll
                for (int r = 0; r < REGS; ++r) {
                    const bool ok = nok & mok[r];
                    hv[r] = bload(re0, ok ? (mrow[r] * p.lde0 + n) * 4u : RAFT_OOB);
                    zv[r] = bload(re1, ok ? (mrow[r] * p.lde1 + n) * 4u : RAFT_OOB);
                }
#pragma unroll
                for (int r = 0; r < REGS; ++r) {
                    const float q = raft_tanh(acc[i][j][r] + bias);
                    bstore((1.0f - zv[r]) * hv[r] + zv[r] * q, ro0,
                           (nok & mok[r]) ? (mrow[r] * p.ldo0 + n) * 4u : RAFT_OOB);
                }
            }
        }
    }
}

// Tile configurations instantiated for every (kernel size, epilogue); see conv.hip::pick_tile.
enum ConvTile {
    TILE_32_128x128 = 0,
    TILE_32_64x128 = 1,
    TILE_32_128x64 = 2,
    TILE_32_64x64 = 3,
    TILE_16_112x128 = 4,
    TILE_16_112x64 = 5,
    TILE_COUNT = 6
};

// Host-side launch with tile selection.  Returns RAFT_E_UNSUPPORTED for an un-instantiated shape.
int raft_launch_conv(const ConvArgs &a, int kh, int kw, int epi, hipStream_t stream);
