// Flow upsampling kernels for gfx950.
//   raft_upsample_convex_f32 : RAFT.upsample_flow  [reference model.py:39-66]
//   raft_upflow8_f32         : upflow8             [reference corr.py:93-96]
#include "common.h"

// One wavefront per coarse pixel: lane = i*8 + j (sub-row i, sub-col j).  The pixel's 576 mask
// logits are read with coalesced loads into LDS, then lane (i,j) reads its 9 taps at stride 9
// (odd stride => conflict-free ds_read_b32), does the softmax in registers, and blends the 3x3
// zero-padded neighbourhood of 8*flow (wave-uniform => scalar loads).
__global__ void __launch_bounds__(256) upsample_convex_kernel(const float *__restrict__ flow,
                                                              const float *__restrict__ mask, int B, int h, int w,
                                                              float *__restrict__ out) {
    __shared__ float sm[4][576];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t npix = (int64_t)B * h * w;
    const int64_t pix = (int64_t)blockIdx.x * 4 + wave;
    const bool active = pix < npix;
    if (active) {
        const float *m = mask + pix * 576;
#pragma unroll
        for (int k = 0; k < 9; ++k) sm[wave][lane + 64 * k] = m[lane + 64 * k];
    }
    __syncthreads();
    if (!active) return;
    const int x = (int)(pix % w), y = (int)((pix / w) % h);
    const int64_t b = pix / ((int64_t)w * h);

    float v[9], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        v[k] = sm[wave][lane * 9 + k];
        mx = fmaxf(mx, v[k]);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        v[k] = expf(v[k] - mx);
        s += v[k];
    }
    float ox = 0.f, oy = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;   // patch depth order (ky, kx, ch)
        float fx = 0.f, fy = 0.f;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
            const float *f = flow + ((b * h + yy) * (int64_t)w + xx) * 2;
            fx = 8.f * f[0];
            fy = 8.f * f[1];
        }
        const float wk = v[k] / s;
        ox += wk * fx;
        oy += wk * fy;
    }
    const int i = lane >> 3, j = lane & 7;
    float2 *o = (float2 *)out + (b * (8 * h) + (8 * y + i)) * (int64_t)(8 * w) + (8 * x + j);
    *o = make_float2(ox, oy);
}

extern "C" int raft_upsample_convex_f32(const float *flow, const float *mask, int B, int h, int w, float *out,
                                        void *stream) {
    RAFT_REQUIRE_PTR(flow);
    RAFT_REQUIRE_PTR(mask);
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    const int64_t npix = (int64_t)B * h * w;
    upsample_convex_kernel<<<raft_ceil_div(npix, 4), 256, 0, (hipStream_t)stream>>>(flow, mask, B, h, w, out);
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
