// On-demand ("alternate") correlation lookup for gfx950: no stored O(N^2) volume.
//
// Average pooling over the target dims commutes with the dot product, so level l of the
// reference's pyramid (corr.py:106-114) equals <fmap1[q], avgpool_l(fmap2)[t]> / sqrt(C).  For each
// query pixel the (2r+2)^2 footprint correlations of every level are computed from fmap1[q] and the
// pooled fmap2 pyramid (raft_fmap_pyramid_f32) with wave-wide dot products, staged in LDS, and the
// window is then evaluated exactly like the volume lookup (same clamp / ceil-floor semantics,
// reference corr.py:116-152, 28-69).  The reference has no such path (README.md:109); this is the
// high-resolution configuration of BASELINE.json (1024x1024: the volume would be 1.43 GB / pair).
#include <stdlib.h>

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

// ------------------------------------------------------------------------------------------------
// Blocked variant: a workgroup serves a 4 x 4 block of query pixels.  Neighbouring queries look at almost the same
// targets, so per level the block's footprints are covered by one bounding box of targets (<= 24 rows x 32 columns
// at the level's resolution, typically ~13 x 13) and the correlations  <fmap1[q], fmap2_l[t]>  of all 16 queries with
// every 16-target run of a bounding-box row come out of one 16 x 16 x C fp32-MFMA tile: fmap1 of the block stays in
// registers (lane = (query, k-quad)), a target row is read ONCE per block instead of once per query (the wave-per-
// query kernel above moves 400 KB of fmap2 rows per query through L2: that, not arithmetic, is its limit).  The tile
// results go to LDS as per-query correlation patches; the window evaluation is the same clamp / ceil-floor code.
// Per level the block picks the coarsest grouping whose boxes fit: the whole block (one MFMA pass), its four 2 x 2
// sub-blocks (four passes), or -- flow that diverges by many pixels inside 2 x 2 pixels -- the wave-wide dot products
// of the kernel above for each query, so the worst case costs what that kernel costs.
// ------------------------------------------------------------------------------------------------
template <int R, int C>
__global__ void __launch_bounds__(256, 2) corr_lookup_ondemand_block_kernel(OnDemandArgs p, int B, int H, int W) {
    constexpr int D = 2 * R + 1, FW = 2 * R + 2, KQ = C / 16;      // KQ b128 per lane = all k-steps of one operand row
    constexpr int BH = 24, BW = 32;                                  // box capacity (rows, columns): 48 KB of patches
    constexpr int V = C / 64;                                        // channels per lane of the wave-wide dot products
    __shared__ float sC[16][BH * BW];                                // correlation patch per query
    __shared__ int sorg[16][2];                                      // footprint origin of every query at this level
    __shared__ int sbox[16][2];                                      // origin of the box its patch is stored relative to
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int G = lane >> 4, LR = lane & 15;
    const int bxn = (W + 3) >> 2, byn = (H + 3) >> 2;
    const int blk = blockIdx.x;
    const int b = blk / (bxn * byn), by = (blk / bxn) % byn, bx = blk % bxn;
    // query qi of the block: (qy, qx), clamped duplicates at ragged edges (their stores are masked)
    auto query_of = [&](int qi, int &qy, int &qx, bool &valid) {
        const int yy = by * 4 + (qi >> 2), xx = bx * 4 + (qi & 3);
        valid = yy < H && xx < W;
        qy = yy < H ? yy : H - 1;
        qx = xx < W ? xx : W - 1;
    };
    int qy, qx;
    bool qvalid;
    query_of(LR, qy, qx, qvalid);
    const int64_t qlin = ((int64_t)b * H + qy) * W + qx;
    // A operand: lane (query LR, k-quad G) holds channels 16 kk + 4 G .. + 3, kk = 0 .. KQ-1
    f32x4 fa[KQ];
    {
        const float *src = p.fmap1 + qlin * C + 4 * G;
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) fa[kk] = *(const f32x4 *)(src + 16 * kk);
    }
    const float cx0 = p.coords[2 * qlin], cy0 = p.coords[2 * qlin + 1];
    const float *f2b = p.f2pyr + (int64_t)b * p.T * C;
    const float inv = 1.0f / p.sqrt_c;

    for (int l = 0; l < p.levels; ++l) {
        const float sc = 1.0f / (float)(1 << l);
        const int w = p.lw[l], h = p.lh[l], tiles_x = p.tx[l];
        if (tid < 16) {
            sorg[tid][0] = axis_tap(cx0 * sc, -R, w).i0;
            sorg[tid][1] = axis_tap(cy0 * sc, -R, h).i0;
        }
        __syncthreads();
        // bounding boxes of the block and of its 2 x 2 sub-blocks (sub-block of query qi: (qi >> 3) * 2 + ((qi >> 1) & 1))
        int lo_x[5], hi_x[5], lo_y[5], hi_y[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) { lo_x[k] = lo_y[k] = 1 << 30; hi_x[k] = hi_y[k] = 0; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ox = sorg[i][0], oy = sorg[i][1];
            const int ex = min(ox + FW - 1, w - 1), ey = min(oy + FW - 1, h - 1);
            const int sb = ((i >> 3) << 1) | ((i >> 1) & 1);
            lo_x[4] = min(lo_x[4], ox); hi_x[4] = max(hi_x[4], ex); lo_y[4] = min(lo_y[4], oy); hi_y[4] = max(hi_y[4], ey);
            lo_x[sb] = min(lo_x[sb], ox); hi_x[sb] = max(hi_x[sb], ex); lo_y[sb] = min(lo_y[sb], oy); hi_y[sb] = max(hi_y[sb], ey);
        }
        const bool fits16 = (hi_x[4] - lo_x[4] < BW) && (hi_y[4] - lo_y[4] < BH);
        bool fits4 = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) fits4 = fits4 && (hi_x[k] - lo_x[k] < BW) && (hi_y[k] - lo_y[k] < BH);
        // all three conditions are workgroup-uniform (every thread read the same 16 origins)
        if (fits16 || fits4) {
            const int npass = fits16 ? 1 : 4;
            for (int pass = 0; pass < npass; ++pass) {
                const int k = fits16 ? 4 : pass;
                const int bx0 = lo_x[k], by0 = lo_y[k], bw = hi_x[k] - lo_x[k] + 1, bh = hi_y[k] - lo_y[k] + 1;
                const int gx = (bw + 15) >> 4, ngroups = bh * gx;
                const float *lvl = f2b + p.row_off[l] * C + 4 * G;
                for (int g = wv; g < ngroups; g += 4) {
                    const int gy = g / gx, gxx = g - gy * gx;
                    const int yy = by0 + gy, xx = min(bx0 + gxx * 16 + LR, w - 1);
                    const float *row = lvl + (int64_t)raft_tiled_index(yy, xx, tiles_x) * C;   // target LR of the run
                    f32x4 fbv[KQ];
#pragma unroll
                    for (int kk = 0; kk < KQ; ++kk) fbv[kk] = *(const f32x4 *)(row + 16 * kk);
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < KQ; ++kk)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[kk][e], fbv[kk][e], acc, 0, 0, 0);
                    // acc[r] = <query 4G + r, target LR>; a sub-block pass keeps only its own queries' rows
                    const int col = gxx * 16 + LR;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qi = 4 * G + r;
                        const bool mine = fits16 || ((((qi >> 3) << 1) | ((qi >> 1) & 1)) == pass);
                        if (mine && col < BW) sC[qi][gy * BW + col] = acc[r] * inv;
                    }
                }
                if (tid < 16 && (fits16 || ((((tid >> 3) << 1) | ((tid >> 1) & 1)) == pass))) {
                    sbox[tid][0] = bx0;
                    sbox[tid][1] = by0;
                }
            }
        } else {
            // wave-wide dot products, four queries per wave (the algorithm of corr_lookup_ondemand_kernel)
            for (int k = 0; k < 4; ++k) {
                const int qi = wv * 4 + k;
                int yq, xq;
                bool vq;
                query_of(qi, yq, xq, vq);
                const int64_t ql = ((int64_t)b * H + yq) * W + xq;
                float f1[V];
                {
                    const float *src = p.fmap1 + ql * C + lane * V;
#pragma unroll
                    for (int v = 0; v < V; ++v) f1[v] = src[v];
                }
                const int ox = sorg[qi][0], oy = sorg[qi][1];
                const float *lvl = f2b + p.row_off[l] * C + lane * V;
                for (int i = 0; i < FW * FW; ++i) {
                    const int fy = i / FW, fx = i - fy * FW;
                    const int yy = min(oy + fy, h - 1), xx = min(ox + fx, w - 1);
                    const float *row = lvl + (int64_t)raft_tiled_index(yy, xx, tiles_x) * C;
                    float sacc = 0.f;
#pragma unroll
                    for (int v = 0; v < V; ++v) sacc = fmaf(f1[v], row[v], sacc);
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off, 64);
                    if (lane == 0) sC[qi][fy * BW + fx] = sacc * inv;
                }
                if (lane == 0) {
                    sbox[qi][0] = ox;
                    sbox[qi][1] = oy;
                }
            }
        }
        __syncthreads();
        {   // window evaluation of this level for the 16 queries, each relative to the origin of its own box
#pragma clang fp contract(off)
            for (int it = tid; it < 16 * D * D; it += 256) {
                const int qi = it / (D * D), c = it - qi * (D * D);
                const int a = c / D, bb = c - a * D;
                int yq, xq;
                bool vq;
                query_of(qi, yq, xq, vq);
                if (!vq) continue;
                const int64_t ql = ((int64_t)b * H + yq) * W + xq;
                const float cx = p.coords[2 * ql] * sc, cy = p.coords[2 * ql + 1] * sc;
                const AxisTap tx = axis_tap(cx, a - R, w), ty = axis_tap(cy, bb - R, h);
                const float *f = sC[qi];
                const int bx0 = sbox[qi][0], by0 = sbox[qi][1];
                // clamped duplicates of a footprint (taps beyond the map edge) were stored at their clamped position
                const int x0 = tx.i0 - bx0, x1 = tx.i1 - bx0;
                const int y0 = (ty.i0 - by0) * BW, y1 = (ty.i1 - by0) * BW;
                const float c00 = ty.w0 * tx.w0, c01 = ty.w0 * tx.w1, c10 = ty.w1 * tx.w0, c11 = ty.w1 * tx.w1;
                float v = c00 * f[y0 + x0] + c01 * f[y0 + x1];
                v = v + c10 * f[y1 + x0];
                v = v + c11 * f[y1 + x1];
                p.out[ql * (int64_t)p.ld_out + l * (D * D) + c] = v;
            }
        }
        __syncthreads();
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
    {   // blocked MFMA kernel (RAFT_ONDEMAND_BLOCK=0 selects the wave-per-query kernel below: A/B timing, parity tests)
        const bool block = raft_opt(RAFT_OPT_ONDEMAND_BLOCK, 1) != 0;
        const int nblk = B * ((h + 3) / 4) * ((w + 3) / 4);
        if (block && radius == 4 && C == 256) {
            corr_lookup_ondemand_block_kernel<4, 256><<<nblk, 256, 0, s>>>(a, B, h, w);
            return raft_launch_status();
        }
        if (block && radius == 3 && C == 128) {
            corr_lookup_ondemand_block_kernel<3, 128><<<nblk, 256, 0, s>>>(a, B, h, w);
            return raft_launch_status();
        }
    }
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
