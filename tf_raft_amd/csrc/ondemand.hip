// On-demand ("alternate") correlation lookup for gfx950: no stored O(N^2) volume.
//
// Average pooling over the target dims commutes with the dot product, so level l of the
// reference's pyramid (corr.py:106-114) equals <fmap1[q], avgpool_l(fmap2)[t]> / sqrt(C).  For each
// query pixel the (2r+2)^2 footprint correlations of every level are computed from fmap1[q] and the
// pooled fmap2 pyramid (raft_fmap_pyramid_f32) -- on fp32 MFMA for 4 x 8 blocks of queries (the default
// kernel, second half of this file), or with wave-wide dot products one query at a time (first kernel:
// other (radius, C) instances, A/B timing, and the fallback inside the blocked kernel) -- staged in LDS,
// and the window is then evaluated exactly like the volume lookup (same clamp / ceil-floor semantics,
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
// Blocked variant: a workgroup serves one pyramid level of a 4 x 8 block of query pixels -- two 4 x 4 sub-blocks, two
// waves each.  Neighbouring queries look at almost the same targets, so the block's footprints are covered by one
// bounding box of targets (typically 13 x 17 at level 0) and the correlations  <fmap1[q], fmap2_l[t]>  of a sub-block's
// 16 queries with 16 targets come out of one 16 x 16 x C fp32-MFMA tile: fmap1 of the sub-block stays in registers
// (lane = (query, k-quad)) and a target row is read once per block instead of once per query (the wave-per-query
// kernel above moves 400 KB of fmap2 rows per query through L2: that, not arithmetic, is its limit).
//   * The union box is linearised row-major and cut into runs of 16 targets with no per-row padding.
//   * A step stages two runs: every wave fetches 8 of the 32 rows, one whole row (C contiguous floats) per wave-wide
//     load at a scalar row address, into registers while the previous step's MFMAs issue; between the step's two
//     barriers the rows go to LDS (row stride C + 8 floats: the ds_read_b128 of the B-operand layout is conflict-free)
//     and both sub-blocks take their B operand from there, the two waves of a sub-block one run each.
//   * Only the (2r+2)^2 footprint of each query is kept (14 KB of LDS), so there is no box capacity: flow that
//     diverges inside the block costs more runs, smoothly, and only a union box of more than MAXG runs falls back to
//     the wave-wide dot products of the first kernel.
//   * The window evaluation is the same clamp / ceil-floor code as the volume lookup.
// History (profiles/r05y_ondemand_*): the first blocked kernel (4 x 4 queries per workgroup, B operand straight from
// global memory in 64-byte pieces, bounding-box patches in 50 KB of LDS) ran its MFMA pipe 30 % busy and took the same
// time with the MFMAs removed -- it waited for target rows, 16 KB per run, fetched by every block on its own.
// ------------------------------------------------------------------------------------------------
template <int R, int C>
__global__ void __launch_bounds__(256, 3) corr_lookup_ondemand_block_kernel(OnDemandArgs p, int B, int H, int W) {
    constexpr int D = 2 * R + 1, FW = 2 * R + 2, FWS = FW + 1, KQ = C / 16, V = C / 64;
    constexpr int RS = C + 8;                  // staged row stride in floats
    constexpr int LPR = C / 4;                 // lanes per staged row (16 bytes each)
    constexpr int RPI = 64 / LPR;              // rows per wave-wide load
    constexpr int NI = 8 / RPI;                // loads per wave per step: 8 of the step's rows
    constexpr int NPAR = 2;                    // waves per sub-block: each takes one run of a step
    constexpr int NT = 128 * NPAR;             // threads
    constexpr int SROWS = 16 * NPAR;           // target rows staged per step: one run per wave pair
    constexpr int MAXG = 96;                   // runs per level beyond which the union box is not worth staging
    __shared__ __attribute__((aligned(16))) float sT[SROWS * RS];   // the step's target rows (NPAR runs)
    __shared__ float sP[32][FW * FWS];         // footprint correlations of every query at this level
    __shared__ int sorg[32][2];                // footprint origin of every query at this level
    __shared__ int swb[2][4];                  // lo_x, lo_y, hi_x, hi_y of the two sub-blocks
    __shared__ float stap[32][2][D];           // clamped tap coordinate of every (query, axis, offset), minus the origin
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR
    const int G = lane >> 4, LR = lane & 15;
    const int sb = wv & 1, par = wv >> 1;      // sub-block (left / right 4 x 4), which run of a step this wave takes
    const int bxn = (W + 7) >> 3, byn = (H + 3) >> 2;
    const int blk = blockIdx.x;
    const int b = blk / (bxn * byn), by = (blk / bxn) % byn, bx = blk % bxn;
    // query qi = 16 * sub-block + 4 * row + column; clamped duplicates at ragged edges (their stores are masked)
    auto query_of = [&](int qi, int &qy, int &qx, bool &valid) {
        const int yy = by * 4 + ((qi & 15) >> 2), xx = bx * 8 + 4 * (qi >> 4) + (qi & 3);
        valid = yy < H && xx < W;
        qy = yy < H ? yy : H - 1;
        qx = xx < W ? xx : W - 1;
    };
    int qy, qx;
    bool qvalid;
    query_of(sb * 16 + LR, qy, qx, qvalid);
    const int64_t qlin = ((int64_t)b * H + qy) * W + qx;
    // A operand: lane (query LR of the sub-block, k-quad G) holds channels 16 kk + 4 G .. + 3, kk = 0 .. KQ-1
    f32x4 fa[KQ];
    {
        const float *src = p.fmap1 + qlin * C + 4 * G;
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) fa[kk] = *(const f32x4 *)(src + 16 * kk);
    }
    const float cx0 = p.coords[2 * qlin], cy0 = p.coords[2 * qlin + 1];
    const float *f2b = p.f2pyr + (int64_t)b * p.T * C;
    const float inv = 1.0f / p.sqrt_c;
    const int srow = lane / LPR, schunk = 4 * (lane % LPR);           // this lane's place in a wave-wide row load

    {   // one pyramid level per workgroup (blockIdx.y): four times the workgroups to fill the chip with
        const int l = blockIdx.y;
        const float sc = 1.0f / (float)(1 << l);
        const int w = p.lw[l], h = p.lh[l], tiles_x = p.tx[l];
        {   // footprint of query LR, bounding box of the sub-block by butterfly min / max over its 16 lanes
            const int ox = axis_tap(cx0 * sc, -R, w).i0, oy = axis_tap(cy0 * sc, -R, h).i0;
            int lx = ox, ly = oy, hx = min(ox + FW - 1, w - 1), hy = min(oy + FW - 1, h - 1);
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                lx = min(lx, __shfl_xor(lx, m, 64));
                ly = min(ly, __shfl_xor(ly, m, 64));
                hx = max(hx, __shfl_xor(hx, m, 64));
                hy = max(hy, __shfl_xor(hy, m, 64));
            }
            if (par == 0 && lane < 16) {
                sorg[sb * 16 + LR][0] = ox;
                sorg[sb * 16 + LR][1] = oy;
            }
            if (par == 0) {
                // the reference's clamp (corr.py:41-48), once per (query, axis, offset) instead of once per window value;
                // g - origin is exact (origin is an integer <= g), so floor / ceil / ceil-g / g-floor of the difference
                // are bit for bit those of axis_tap on g
                for (int k = G; k < 2 * D; k += 4) {
                    const int axis = k >= D, d = k - axis * D - R;
                    float g = (axis ? cy0 : cx0) * sc + (float)d;
                    g = fminf(fmaxf(g, 0.f), (float)((axis ? h : w) - 1));
                    stap[sb * 16 + LR][axis][k - axis * D] = g - (float)(axis ? oy : ox);
                }
            }
            if (par == 0 && lane == 0) {
                swb[sb][0] = lx; swb[sb][1] = ly; swb[sb][2] = hx; swb[sb][3] = hy;
            }
        }
        __syncthreads();
        // read from shared memory, so workgroup-uniform: say so (readfirstlane) to keep the box arithmetic on the scalar unit
        const int bx0 = __builtin_amdgcn_readfirstlane(min(swb[0][0], swb[1][0]));
        const int by0 = __builtin_amdgcn_readfirstlane(min(swb[0][1], swb[1][1]));
        const int bw = __builtin_amdgcn_readfirstlane(max(swb[0][2], swb[1][2])) - bx0 + 1;
        const int bh = __builtin_amdgcn_readfirstlane(max(swb[0][3], swb[1][3])) - by0 + 1;
        const int U = bw * bh, ngroups = (U + 15) >> 4;               // targets of the union box, row-major, runs of 16
        // t / bw without the integer-division sequence: t < 16 MAXG and bw <= 16 MAXG / FW, so (t + 0.5) / bw is at least
        // 0.5 / bw ~ 3e-3 away from an integer while the float product is within ~2e-5 of it
        const float rbw = 1.0f / (float)bw;
        auto row_col = [&](int t, int &ty, int &tx) {
            ty = (int)(((float)t + 0.5f) * rbw);
            tx = t - ty * bw;
        };
        // the same quotient in integers for wave-uniform t (kept on the scalar unit): m = ceil(2^20 / bw) is exact for
        // t * (m * bw - 2^20) < 2^20, and t < 16 MAXG, bw >= FW keep t * m below 2^31
        const unsigned mbw = ((1u << 20) + (unsigned)bw - 1u) / (unsigned)bw;
        auto row_col_uniform = [&](int t, int &ty, int &tx) {
            ty = (int)(((unsigned)t * mbw) >> 20);
            tx = t - ty * bw;
        };
        const float *lvl = f2b + p.row_off[l] * C;
        if (ngroups <= MAXG) {                                        // workgroup-uniform
            // origins of the four queries whose rows this lane's accumulator holds (query 4 G + r of the sub-block)
            int oxq[4], oyq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                oxq[r] = sorg[sb * 16 + 4 * G + r][0];
                oyq[r] = sorg[sb * 16 + 4 * G + r][1];
            }
            const int nsteps = (ngroups + NPAR - 1) / NPAR;
            // rows of step i: global -> registers one step ahead, registers -> LDS between the step's two barriers
            auto prefetch = [&](int i, f32x4 (&pre)[NI]) {
#pragma unroll
                for (int n = 0; n < NI; ++n) {
                    const int t = min(SROWS * i + 8 * wv + n * RPI + srow, U - 1);   // the tail repeats the last target
                    int ty, tx;
                    if (RPI == 1) row_col_uniform(t, ty, tx);          // one row per load: the row address is wave-uniform
                    else row_col(t, ty, tx);
                    pre[n] = *(const f32x4 *)(lvl + (int64_t)raft_tiled_index(by0 + ty, bx0 + tx, tiles_x) * C + schunk);
                }
            };
            auto step = [&](int i, f32x4 (&pre)[NI]) {
                __syncthreads();                                      // every wave is done with the previous step's rows
#pragma unroll
                for (int n = 0; n < NI; ++n)
                    *(f32x4 *)(sT + (8 * wv + n * RPI + srow) * RS + schunk) = pre[n];
                __syncthreads();
                if (i + 1 < nsteps) prefetch(i + 1, pre);             // in flight under this step's MFMAs
                const int g = NPAR * i + par;
                if (g < ngroups) {
                    const float *brow = sT + (16 * par + LR) * RS + 4 * G;     // B operand: lane (target LR, k-quad G)
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < KQ; kk += 2) {
                        const f32x4 fb0 = *(const f32x4 *)(brow + 16 * kk);
                        const f32x4 fb1 = *(const f32x4 *)(brow + 16 * kk + 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[kk][e], fb0[e], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[kk + 1][e], fb1[e], acc1, 0, 0, 0);
                        }
                    }
                    // acc[r] = <query 4 G + r, target 16 g + LR>: kept where the target lies in that query's footprint
                    const int t = 16 * g + LR;
                    int ty, tx;
                    row_col(t, ty, tx);
                    if (t < U) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int fy = by0 + ty - oyq[r], fx = bx0 + tx - oxq[r];
                            if ((unsigned)fy < (unsigned)FW && (unsigned)fx < (unsigned)FW)
                                sP[sb * 16 + 4 * G + r][fy * FWS + fx] = (acc0[r] + acc1[r]) * inv;
                        }
                    }
                }
            };
            f32x4 pre0[NI];
            prefetch(0, pre0);
            for (int i = 0; i < nsteps; ++i) step(i, pre0);
        } else {
            // wave-wide dot products, the sub-block's queries dealt to its waves (the algorithm of corr_lookup_ondemand_kernel)
            for (int k = par; k < 16; k += NPAR) {
                const int qi = sb * 16 + k;
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
                const float *lv = lvl + lane * V;
                for (int i = 0; i < FW * FW; ++i) {
                    const int fy = i / FW, fx = i - fy * FW;
                    const int yy = min(oy + fy, h - 1), xx = min(ox + fx, w - 1);
                    const float *row = lv + (int64_t)raft_tiled_index(yy, xx, tiles_x) * C;
                    float sacc = 0.f;
#pragma unroll
                    for (int v = 0; v < V; ++v) sacc = fmaf(f1[v], row[v], sacc);
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off, 64);
                    if (lane == 0) sP[qi][fy * FWS + fx] = sacc * inv;
                }
            }
        }
        __syncthreads();
        {   // window evaluation of this level for the 32 queries, each relative to its own footprint origin
#pragma clang fp contract(off)
            for (int it = tid; it < 32 * D * D; it += NT) {
                const int qi = it / (D * D), c = it - qi * (D * D);
                const int a = c / D, bb = c - a * D;
                int yq, xq;
                bool vq;
                query_of(qi, yq, xq, vq);
                if (!vq) continue;
                const int64_t ql = ((int64_t)b * H + yq) * W + xq;
                const float gx = stap[qi][0][a], gy = stap[qi][1][bb];
                const float fx0 = floorf(gx), fx1 = ceilf(gx), fy0 = floorf(gy), fy1 = ceilf(gy);
                const float wx0 = fx1 - gx, wx1 = gx - fx0, wy0 = fy1 - gy, wy1 = gy - fy0;
                const float *f = sP[qi];
                // taps beyond the map edge are clamped to it, i.e. to positions inside the footprint
                const int x0 = (int)fx0, x1 = (int)fx1;
                const int y0 = (int)fy0 * FWS, y1 = (int)fy1 * FWS;
                const float c00 = wy0 * wx0, c01 = wy0 * wx1, c10 = wy1 * wx0, c11 = wy1 * wx1;
                float v = c00 * f[y0 + x0] + c01 * f[y0 + x1];
                v = v + c10 * f[y1 + x0];
                v = v + c11 * f[y1 + x1];
                p.out[ql * (int64_t)p.ld_out + l * (D * D) + c] = v;
            }
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
    {   // blocked MFMA kernel (RAFT_ONDEMAND_BLOCK=0 selects the wave-per-query kernel below: A/B timing, parity tests)
        const bool block = raft_opt(RAFT_OPT_ONDEMAND_BLOCK, 1) != 0;
        const dim3 grid(B * ((h + 3) / 4) * ((w + 7) / 8), levels);
        if (block && radius == 4 && C == 256) {
            corr_lookup_ondemand_block_kernel<4, 256><<<grid, 256, 0, s>>>(a, B, h, w);
            return raft_launch_status();
        }
        if (block && radius == 3 && C == 128) {
            corr_lookup_ondemand_block_kernel<3, 128><<<grid, 256, 0, s>>>(a, B, h, w);
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
