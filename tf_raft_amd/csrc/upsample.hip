// Flow upsampling kernels for gfx950.
//   raft_upsample_convex_f32 : RAFT.upsample_flow  [reference model.py:39-66]
//   raft_upflow8_f32         : upflow8             [reference corr.py:93-96]
#include "common.h"

// One wavefront per PAIR of x-adjacent coarse pixels.  lane = il*16 + px*8 + j (il = 0..3, px = which pixel of
// the pair, j = sub-column); the lane produces sub-rows i = il and il + 4, so every store instruction writes four
// output rows of 16 consecutive float2 = whole 128-byte lines.  The pair's 2 x 576 mask logits are contiguous in
// memory: 16-byte coalesced loads into LDS (pixel stride 592 floats => the stride-9 tap reads of a half-wave hit 32
// distinct banks), the 3x4 zero-padded neighbourhood of 8*flow goes to LDS once per pair (broadcast reads), the
// softmax over the 9 taps runs in registers.  exp(x - max) is evaluated as exp2((x - max) * log2 e) on the
// transcendental unit and the weights are scaled by one reciprocal of their sum: both within a few ulp of the
// reference's softmax, far inside the 1e-5 relative tolerance of the parity test.
constexpr int UPS_PX_STRIDE = 592;

__global__ void __launch_bounds__(256) upsample_convex_kernel(const float *__restrict__ flow,
                                                              const float *__restrict__ mask, int B, int h, int w,
                                                              float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float sm[4][2 * UPS_PX_STRIDE];
    __shared__ float2 sf[4][12];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wp = (w + 1) >> 1;
    const int64_t npair = (int64_t)B * h * wp;
    const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
    const bool active = pair < npair;                         // wave-uniform
    int x0 = 0, y = 0;
    int64_t b = 0;
    bool two = false;
    if (active) {
        x0 = 2 * (int)(pair % wp);
        y = (int)((pair / wp) % h);
        b = pair / ((int64_t)wp * h);
        two = x0 + 1 < w;
        const f32x4 *m = (const f32x4 *)(mask + ((b * h + y) * (int64_t)w + x0) * 576);
        const int lim = two ? 288 : 144;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int f = lane + 64 * k;
            if (f < lim) {
                const int px = f >= 144, r = f - 144 * px;
                *(f32x4 *)(&sm[wave][px * UPS_PX_STRIDE + 4 * r]) = m[f];
            }
        }
        if (lane < 12) {                                      // neighbourhood rows y-1..y+1, columns x0-1..x0+2
            const int yy = y + lane / 4 - 1, xx = x0 + (lane & 3) - 1;
            float2 f = make_float2(0.f, 0.f);
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                f = *(const float2 *)(flow + ((b * h + yy) * (int64_t)w + xx) * 2);
                f.x *= 8.f;
                f.y *= 8.f;
            }
            sf[wave][lane] = f;
        }
    }
    __syncthreads();
    const int il = lane >> 4, px = (lane >> 3) & 1, j = lane & 7;
    if (!active || (px && !two)) return;
    float2 nb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) nb[k] = sf[wave][(k / 3) * 4 + (k % 3) + px];   // patch depth order (ky, kx, ch)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int i = il + 4 * pass;
        const float *t = &sm[wave][px * UPS_PX_STRIDE + (i * 8 + j) * 9];
        float v[9], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            v[k] = t[k];
            mx = fmaxf(mx, v[k]);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            v[k] = __builtin_amdgcn_exp2f((v[k] - mx) * 1.44269504088896341f);
            s += v[k];
        }
        const float inv = 1.0f / s;
        float ox = 0.f, oy = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float wk = v[k] * inv;
            ox += wk * nb[k].x;
            oy += wk * nb[k].y;
        }
        float2 *o = (float2 *)out + (b * (8 * h) + (8 * y + i)) * (int64_t)(8 * w) + (8 * (x0 + px) + j);
        *o = make_float2(ox, oy);
    }
}

extern "C" int raft_upsample_convex_f32(const float *flow, const float *mask, int B, int h, int w, float *out,
                                        void *stream) {
    RAFT_REQUIRE_PTR(flow);
    RAFT_REQUIRE_PTR(mask);
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    RAFT_REQUIRE(raft_aligned16(mask), RAFT_E_ALIGN);
    const int64_t npair = (int64_t)B * h * ((w + 1) / 2);
    upsample_convex_kernel<<<raft_ceil_div(npair, 4), 256, 0, (hipStream_t)stream>>>(flow, mask, B, h, w, out);
    return raft_launch_status();
}

// tf.image.resize(..., 'bilinear') with half-pixel centres, scale 8, then * 8.
__global__ void __launch_bounds__(256) upflow8_kernel(const float2 *__restrict__ flow, int B, int h, int w,
                                                      float2 *__restrict__ out) {
    const int64_t total = (int64_t)B * 8 * h * 8 * w;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int W8 = 8 * w, H8 = 8 * h;
    const int X = (int)(i % W8), Y = (int)((i / W8) % H8);
    const int64_t b = i / ((int64_t)W8 * H8);
    const float sx = ((float)X + 0.5f) * 0.125f - 0.5f, sy = ((float)Y + 0.5f) * 0.125f - 0.5f;
    const float fx = floorf(sx), fy = floorf(sy);
    const int x0 = max((int)fx, 0), x1 = min((int)ceilf(sx), w - 1);
    const int y0 = max((int)fy, 0), y1 = min((int)ceilf(sy), h - 1);
    const float lx = sx - fx, ly = sy - fy;
    const float2 *f = flow + b * h * (int64_t)w;
    const float2 tl = f[y0 * w + x0], tr = f[y0 * w + x1], bl = f[y1 * w + x0], br = f[y1 * w + x1];
    const float tx = tl.x + (tr.x - tl.x) * lx, ty = tl.y + (tr.y - tl.y) * lx;
    const float bx = bl.x + (br.x - bl.x) * lx, by = bl.y + (br.y - bl.y) * lx;
    out[i] = make_float2(8.f * (tx + (bx - tx) * ly), 8.f * (ty + (by - ty) * ly));
}

extern "C" int raft_upflow8_f32(const float *flow, int B, int h, int w, float *out, void *stream) {
    RAFT_REQUIRE_PTR(flow);
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    const int64_t total = (int64_t)B * 64 * h * w;
    upflow8_kernel<<<raft_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>((const float2 *)flow, B, h, w,
                                                                               (float2 *)out);
    return raft_launch_status();
}

// Streaming copy used by bench.py to measure the box's HBM copy bandwidth (read + write, 16 bytes per lane, one
// element per thread: the flat grid measured 6.0-6.1 TB/s on MI355X, a grid-stride loop 4.7-5.4, hipMemcpyAsync
// 4.8-5.3 -- tools/ablate/copy_bw.hip).
__global__ void __launch_bounds__(256) stream_copy_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                                          int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

extern "C" int raft_stream_copy_f32(const float *src, float *dst, int64_t n, void *stream) {
    RAFT_REQUIRE_PTR(src);
    RAFT_REQUIRE_PTR(dst);
    RAFT_REQUIRE(n > 0 && n % 4 == 0 && n / 1024 < 0x7fffffff, RAFT_E_SHAPE);
    RAFT_REQUIRE(raft_aligned16(src) && raft_aligned16(dst), RAFT_E_ALIGN);
    const int64_t n4 = n / 4;
    stream_copy_kernel<<<raft_ceil_div(n4, 256), 256, 0, (hipStream_t)stream>>>((const f32x4 *)src, (f32x4 *)dst, n4);
    return raft_launch_status();
}

// Measurement utility: what the fp32 matrix pipe of THIS box sustains.  Every wave runs `iters` trips of 8 independent
// v_mfma_f32_16x16x4_f32 accumulation chains on loop-invariant, lane-varying, non-zero operands (nothing else in the loop: no
// loads, no VALU), 2 waves per SIMD, `blocks` workgroups of 256 threads.  FLOPs = blocks * 4 * iters * 8 * 2048.
__global__ void __launch_bounds__(256, 2) mfma_probe_kernel(float *__restrict__ out, int iters) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const int t = threadIdx.x;
    float a[8], b[8];
    f32x4_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = 0.5f + 0.0013f * (float)((t * 37 + j * 11) & 255);
        b[j] = 1.25f - 0.0021f * (float)((t * 53 + j * 7) & 255);
        acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[(int64_t)blockIdx.x * 256 + t] = s;
}

extern "C" int raft_mfma_probe_f32(float *out, int blocks, int iters, void *stream) {
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(blocks > 0 && iters > 0, RAFT_E_SHAPE);
    mfma_probe_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(out, iters);
    return raft_launch_status();
}
