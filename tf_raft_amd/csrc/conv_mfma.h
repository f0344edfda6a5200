// Implicit-GEMM stride-1 'same' convolution on v_mfma_f32_32x32x2_f32 (exact fp32) for gfx950.
//
// GEMM view: M = B*H*W output pixels, N = output channels, K = taps x input channels.
// One K-step = (tap t, 32-channel chunk): the A tile is the [BM pixels][32 ch] slice of the NHWC
// input shifted by the tap (zero outside the image), the B tile is [8 k-quads][BN][4] of the
// packed weights (include/raft_hip.h).  Both go global -> registers -> LDS (double-buffered, one
// barrier per K-step) and are consumed as ds_read_b128 fragments:
//   A row stride 36 floats (144 B): a 16-lane ds_read_b128 group hits 16 distinct 16-B slots;
//   B rows are lane-contiguous.
// Per 8-wide k sub-step lanes 0-31 carry k-quad 2*kk, lanes 32-63 k-quad 2*kk+1, for A and B alike
// (v_mfma_f32_32x32x2: lane l supplies k = l>>5), i.e. a fixed permutation of the K sum.
// 4 waves (2 x 2); wave tile (BM/2) x (BN/2) = TM x TN MFMA tiles of 32 x 32.
// Accumulator layout: lane owns output channel n = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5).
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
    const int64_t M = (int64_t)p.B * p.H * p.W;
    const int ntn = p.npad / BN;
    const int mt = blockIdx.x / ntn, nt = blockIdx.x % ntn;
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;
    const int cin = p.c0 + p.c1;
    const int nch = cin / BK;
    const int S = KH * KW * nch;

    // A staging rows of this thread
    const int srow = tid >> 3, sc4 = tid & 7;
    int py[NA], px[NA];
    int64_t pm[NA];
    bool pv[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int64_t m = m0 + srow + 32 * i;
        pv[i] = m < M;
        pm[i] = m;
        px[i] = (int)(m % p.W);
        py[i] = (int)((m / p.W) % p.H);
    }

    f32x4 ra[NA], rb[NB];
    auto gload = [&](int s) {
        const int t = s / nch, cc = s - t * nch;
        const int dy = t / KW - (KH - 1) / 2, dx = t % KW - (KW - 1) / 2;
        const int c = cc * BK;
        const float *src;
        int ld, ch;
        if (c < p.c0) {
            src = p.a0; ld = p.lda0; ch = c;
        } else {
            src = p.a1; ld = p.lda1; ch = c - p.c0;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int yy = py[i] + dy, xx = px[i] + dx;
            const bool ok = pv[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ra[i] = ok ? *(const f32x4 *)(src + (pm[i] + (int64_t)dy * p.W + dx) * ld + ch + sc4 * 4) : z;
        }
        const float *wsrc = p.wp + (((int64_t)t * (cin / 4) + c / 4) * p.npad + n0) * 4;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = tid + 256 * i;
            const int kq = q / BN, j = q - kq * BN;
            rb[i] = *(const f32x4 *)(wsrc + ((int64_t)kq * p.npad + j) * 4);
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
                const int64_t m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
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
