// Implicit-GEMM stride-1 'same' convolution on v_mfma_f32_32x32x2_f32 (exact fp32) for gfx950.
//
// GEMM view: M = B*H*W output pixels, N = output channels, K = taps x input channels.
// One K-step = (tap t, 32-channel chunk): the A tile is the [BM pixels][32 ch] slice of the NHWC
// input shifted by the tap (zero outside the image), the B tile is [8 k-quads][BN][4] of the
// packed weights (include/raft_hip.h).  Both go global -> registers -> LDS (double-buffered, one
// barrier per K-step) and are consumed as ds_read_b128 fragments:
//   A row stride 36 floats (144 B): a 16-lane ds_read_b128 group hits 16 distinct 16-B slots;
//   B rows are lane-contiguous.
// A-tile loads are UNCONDITIONAL buffer loads: out-of-image taps get an out-of-range offset and
// the buffer bounds check returns 0 (a per-chunk `ok ? load : 0` makes hipcc branch around every
// load with an s_waitcnt in between, which serialises the loads and exposes their latency).
// Per 8-wide k sub-step lanes 0-31 carry k-quad 2*kk, lanes 32-63 k-quad 2*kk+1, for A and B alike
// (v_mfma_f32_32x32x2: lane l supplies k = l>>5), i.e. a fixed permutation of the K sum.
// 4 waves (2 x 2); wave tile (BM/2) x (BN/2) = TM x TN MFMA tiles of 32 x 32.
// Accumulator layout: lane owns output channel n = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5).
// Workgroup ids are remapped so that consecutive tiles (same pixel tile, neighbouring N tiles)
// run on the same XCD and share its L2 (hardware places workgroup b on XCD b % 8).
#pragma once
#include "common.h"

enum ConvEpilogue {
    EPI_LINEAR = 0,   // out = (acc + bias) * scale
    EPI_RELU = 1,     // out = relu(acc + bias) * scale
    EPI_GRU_ZR = 2,   // n <  hid: o0 = sigmoid(v)            (z)
                      // n >= hid: o1 = sigmoid(v) * e0[n-hid] (r * h)
    EPI_GRU_Q = 3     // o0 = (1 - e1) * e0 + e1 * tanh(v)     (h <- (1-z) h + z q), e0 = h, e1 = z
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
};

__device__ __forceinline__ float raft_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 raft_buffer_load_f4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0));
}

constexpr unsigned RAFT_OOB = 0x80000000u;   // >= any buffer extent we accept (< 2 GiB): load returns 0

template <int KH, int KW, int BM, int BN, int EPI>
__global__ void __launch_bounds__(256) conv_mfma_kernel(ConvArgs p) {
    constexpr int BK = 32, LDA = 36;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_BUF = BM * LDA, B_BUF = BK * BN;
    constexpr int NA = BM / 32, NB = BN / 32;   // float4 chunks per thread per K-step
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_BUF + B_BUF)];
    float *sA = smem;
    float *sB = smem + 2 * A_BUF;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, half = lane >> 5, l31 = lane & 31;
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
        const int m = m0 + srow + 32 * i;
        pm[i] = m;
        px[i] = m % p.W;
        py[i] = (m / p.W) % p.H;
        if (m >= M) py[i] = -(1 << 20);          // row beyond M: every tap is out of range
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
            const int yy = py[i] + dy, xx = px[i] + dx;
            const bool ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            off[i] = ok ? (unsigned)(((pm[i] + dy * p.W + dx) * ld + ch) * 4) : RAFT_OOB;
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

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        if (s + 1 < S) gload(s + 1);
        const float *cA = sA + buf * A_BUF + (wm * (BM / 2) + l31) * LDA;
        const float *cB = sB + buf * B_BUF + (wn * (BN / 2) + l31) * 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kq = 2 * kk + half;
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const f32x4 *)(cA + i * 32 * LDA + kq * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const f32x4 *)(cB + (kq * BN + j * 32) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][r], fb[j][r], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < S) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + l31;
        if (n >= p.nvalid) continue;
        const float bias = p.bias[n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= M) continue;
                const float v = acc[i][j][r] + bias;
                if (EPI == EPI_LINEAR) {
                    p.o0[m * p.ldo0 + n] = v * p.scale;
                } else if (EPI == EPI_RELU) {
                    p.o0[m * p.ldo0 + n] = fmaxf(v, 0.f) * p.scale;
                } else if (EPI == EPI_GRU_ZR) {
                    const float g = raft_sigmoid(v);
                    if (n < p.hid)
                        p.o0[m * p.ldo0 + n] = g;
                    else
                        p.o1[m * p.ldo1 + (n - p.hid)] = g * p.e0[m * p.lde0 + (n - p.hid)];
                } else {   // EPI_GRU_Q
                    const float q = tanhf(v);
                    const float hprev = p.e0[m * p.lde0 + n], z = p.e1[m * p.lde1 + n];
                    p.o0[m * p.ldo0 + n] = (1.0f - z) * hprev + z * q;
                }
            }
        }
    }
}

// Host-side launch with tile selection.  Returns RAFT_E_UNSUPPORTED for an un-instantiated shape.
int raft_launch_conv(const ConvArgs &a, int kh, int kw, int epi, hipStream_t stream);
