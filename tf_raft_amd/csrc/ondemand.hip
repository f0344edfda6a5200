// On-demand ("alternate") correlation lookup for gfx950: no stored O(N^2) volume.
//
// Average pooling over the target dims commutes with the dot product, so level l of the
// reference's pyramid (corr.py:106-114) equals <fmap1[q], avgpool_l(fmap2)[t]> / sqrt(C).  For each
// query pixel the (2r+2)^2 footprint correlations of every level are computed from fmap1[q] and the
// pooled fmap2 pyramid (raft_fmap_pyramid_f32) with wave-wide dot products, staged in LDS, and the
// window is then evaluated exactly like the volume lookup (same clamp / ceil-floor semantics,
// reference corr.py:116-152, 28-69).  The reference has no such path (README.md:109); this is the
// high-resolution configuration of BASELINE.json (1024x1024: the volume would be 1.43 GB / pair).
#include "common.h"
#include "lookup_common.h"

struct OnDemandArgs {
    const float *fmap1;      // (B, N, C)
    const float *f2pyr;      // (B, T, C) pooled fmap2 pyramid
    const float *coords;     // (B*N, 2)
    float *out;
    int64_t row_off[RAFT_MAX_LEVELS];   // first row of each level inside T
    int lh[RAFT_MAX_LEVELS], lw[RAFT_MAX_LEVELS], tx[RAFT_MAX_LEVELS];
    int64_t nq;
    int N, T, levels, ld_out;
    float sqrt_c;
};

template <int R, int V>   // V = C / 64 channels per lane
__global__ void __launch_bounds__(256) corr_lookup_ondemand_kernel(OnDemandArgs p) {
    constexpr int D = 2 * R + 1, FW = 2 * R + 2, FP = FW * FW;
    constexpr int C = 64 * V;
    __shared__ float sfp[4][RAFT_MAX_LEVELS][FP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    const bool active = q < p.nq;
    float cx0 = 0.f, cy0 = 0.f;
    if (active) {
        cx0 = p.coords[2 * q];
        cy0 = p.coords[2 * q + 1];
        const int64_t b = q / p.N;
        float f1[V];
        {
            const float *src = p.fmap1 + q * C + lane * V;
#pragma unroll
            for (int v = 0; v < V; ++v) f1[v] = src[v];
        }
        const float *f2b = p.f2pyr + b * (int64_t)p.T * C;
        for (int l = 0; l < p.levels; ++l) {
            const float sc = 1.0f / (float)(1 << l);
            const int w = p.lw[l], h = p.lh[l], tiles_x = p.tx[l];
            const AxisTap tx = axis_tap(cx0 * sc, -R, w);
            const AxisTap ty = axis_tap(cy0 * sc, -R, h);
            const float *lvl = f2b + p.row_off[l] * C + lane * V;
            for (int i = 0; i < FP; ++i) {
                const int fy = i / FW, fx = i - fy * FW;
                const int yy = min(ty.i0 + fy, h - 1), xx = min(tx.i0 + fx, w - 1);
                const float *row = lvl + (int64_t)raft_tiled_index(yy, xx, tiles_x) * C;   // rows in tile order
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) s = fmaf(f1[v], row[v], s);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
                if (lane == 0) sfp[wave][l][i] = s / p.sqrt_c;
            }
        }
    }
    __syncthreads();
    if (!active) return;
    {
#pragma clang fp contract(off)
        const int nout = p.levels * D * D;
        float *o = p.out + q * (int64_t)p.ld_out;
        for (int c = lane; c < nout; c += 64) {
            const int l = c / (D * D);
            const int t = c - l * (D * D);
            const int a = t / D, b = t - a * D;
            const float sc = 1.0f / (float)(1 << l);
            const int w = p.lw[l], h = p.lh[l];
            const float cx = cx0 * sc, cy = cy0 * sc;
            const AxisTap ox = axis_tap(cx, -R, w), oy = axis_tap(cy, -R, h);
            const AxisTap tx = axis_tap(cx, a - R, w), ty = axis_tap(cy, b - R, h);
            const float *f = sfp[wave][l];
            const int x0 = tx.i0 - ox.i0, x1 = tx.i1 - ox.i0;
            const int y0 = (ty.i0 - oy.i0) * FW, y1 = (ty.i1 - oy.i0) * FW;
            const float c00 = ty.w0 * tx.w0, c01 = ty.w0 * tx.w1, c10 = ty.w1 * tx.w0, c11 = ty.w1 * tx.w1;
            float v = c00 * f[y0 + x0] + c01 * f[y0 + x1];
            v = v + c10 * f[y1 + x0];
            v = v + c11 * f[y1 + x1];
            o[c] = v;
        }
    }
}

extern "C" int raft_corr_lookup_ondemand_f32(const float *fmap1, const float *fmap2_pyr, const float *coords, int B,
                                             int h, int w, int C, int levels, int radius, float *out, int ld_out,
                                             void *stream) {
    RAFT_REQUIRE_PTR(fmap1);
    RAFT_REQUIRE_PTR(fmap2_pyr);
    RAFT_REQUIRE_PTR(coords);
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    RAFT_REQUIRE(levels >= 1 && levels <= RAFT_MAX_LEVELS, RAFT_E_UNSUPPORTED);
    const int d = 2 * radius + 1;
    RAFT_REQUIRE(ld_out >= levels * d * d, RAFT_E_SHAPE);
    OnDemandArgs a;
    a.fmap1 = fmap1;
    a.f2pyr = fmap2_pyr;
    a.coords = coords;
    a.out = out;
    a.N = h * w;
    a.nq = (int64_t)B * h * w;
    a.levels = levels;
    a.ld_out = ld_out;
    a.sqrt_c = sqrtf((float)C);
    int64_t t = 0;
    int ch = h, cw = w;
    for (int l = 0; l < RAFT_MAX_LEVELS; ++l) {
        a.row_off[l] = 0; a.lh[l] = 1; a.lw[l] = 1; a.tx[l] = 1;
    }
    for (int l = 0; l < levels; ++l) {
        RAFT_REQUIRE(ch >= 1 && cw >= 1, RAFT_E_SHAPE);
        a.row_off[l] = t;
        a.lh[l] = ch;
        a.lw[l] = cw;
        a.tx[l] = raft_tiles_x(cw);
        t += raft_map_floats(ch, cw);
        ch /= 2;
        cw /= 2;
    }
    a.T = (int)t;
    const int blocks = raft_ceil_div(a.nq, 4);
    hipStream_t s = (hipStream_t)stream;
    if (radius == 4 && C == 256)
        corr_lookup_ondemand_kernel<4, 4><<<blocks, 256, 0, s>>>(a);
    else if (radius == 3 && C == 128)
        corr_lookup_ondemand_kernel<3, 2><<<blocks, 256, 0, s>>>(a);
    else if (radius == 4 && C == 128)
        corr_lookup_ondemand_kernel<4, 2><<<blocks, 256, 0, s>>>(a);
    else if (radius == 3 && C == 256)
        corr_lookup_ondemand_kernel<3, 4><<<blocks, 256, 0, s>>>(a);
    else
        return RAFT_E_UNSUPPORTED;
    return raft_launch_status();
}
