// Shared definitions of the fp32-MFMA convolution kernels for gfx950 (conv_halo.h, conv_wino.h, conv_wino1d.h, conv_wino4.h):
// the argument block, the epilogue codes, gate activations, buffer-load helper.
//
// GEMM view of every kernel: M = B*H*W output pixels, N = output channels, K = taps x input channels; NHWC activations, weights
// packed by tf_raft_amd/packing.py into fragment-shaped k-quads (include/raft_hip.h).  Two rules shape all of them:
//   * A-tile loads are UNCONDITIONAL buffer loads: out-of-image taps get an out-of-range offset and the buffer bounds check
//     returns 0 (a per-chunk `ok ? load : 0` makes hipcc branch around every load with an s_waitcnt in between, which
//     serialises the loads and exposes their latency);
//   * the same for the epilogues: the GRU gates read h / z with unconditional (clamped) loads issued as one batch per
//     accumulator tile before any arithmetic.
#pragma once
#include "common.h"

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

// (The first version of the direct kernel -- a (tap, chunk)-stepped LDS pipeline with 32x32 / 16x16 MFMA tiles, tile codes 0..5 of
// RAFT_CONV_TILE -- was kept for A/B through round 3 and removed in round 4: the halo-tiled kernel of conv_halo.h replaced it
// everywhere in round 1; its measurements stay in profiles/r01*.)

// Host-side launch with tile selection.  Returns RAFT_E_UNSUPPORTED for an un-instantiated shape.
int raft_launch_conv(const ConvArgs &a, int kh, int kw, int epi, hipStream_t stream);
