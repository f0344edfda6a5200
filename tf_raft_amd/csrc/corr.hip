// Correlation volume build + pyramid lookup for gfx950 (MI355X).
//
//   raft_corr_build_f32   : fp32 MFMA "NT" GEMM  fmap1 (N x C) . fmap2_pyr (T x C)^T / sqrt(C),
//                           T = sum over levels of the tile-padded map sizes (pooled fmap2, rows in
//                           tile order), written straight into the per-level (B*N, map_l) maps of
//                           4x8-tiled floats.                [reference corr.py:100-114, 154-162]
//   raft_corr_lookup_f32  : 36 threads per query pixel; the (2r+2)^2 footprint of each level is
//                           staged in LDS, the (2r+1)^2 window is evaluated from it and written as
//                           contiguous channels.              [reference corr.py:116-152, 28-69]
//   raft_bilinear_sampler_f32, raft_coords_grid_f32           [reference corr.py:28-69, 72-90]
#include <stdlib.h>

#include "common.h"
#include "lookup_common.h"

// ------------------------------------------------------------------------------------------------
// geometry helpers (host)
// ------------------------------------------------------------------------------------------------
extern "C" int raft_corr_pyramid_layout(int B, int h, int w, int levels, int64_t *level_offsets,
                                        int *lh, int *lw) {
    RAFT_REQUIRE_PTR(level_offsets);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    RAFT_REQUIRE(levels >= 1 && levels <= RAFT_MAX_LEVELS, RAFT_E_UNSUPPORTED);
    int64_t off = 0;
    int ch = h, cw = w;
    const int64_t nq = (int64_t)B * h * w;
    for (int l = 0; l < levels; ++l) {
        RAFT_REQUIRE(ch >= 1 && cw >= 1, RAFT_E_SHAPE);
        level_offsets[l] = off;
        if (lh) lh[l] = ch;
        if (lw) lw[l] = cw;
        off += nq * raft_map_floats(ch, cw);   // whole 128-byte tiles: every map starts on a line
        ch /= 2;
        cw /= 2;
    }
    level_offsets[levels] = off;
    return RAFT_OK;
}

static int64_t pyr_cols(int h, int w, int levels) {
    int64_t t = 0;
    for (int l = 0; l < levels; ++l) {
        t += raft_map_floats(h, w);
        h /= 2;
        w /= 2;
    }
    return t;
}

extern "C" int64_t raft_corr_build_workspace_floats(int B, int h, int w, int C, int levels) {
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || levels < 1 || levels > RAFT_MAX_LEVELS) return 0;
    return (int64_t)B * pyr_cols(h, w, levels) * C;
}

// ------------------------------------------------------------------------------------------------
// fmap2 feature pyramid: level l = 2x2 VALID average of level l-1 (NHWC, channels vectorised x4).
// Workspace layout per batch element: [level0 | level1 | ...] x C floats, the rows (target positions) of a
// level in the TILED order of its correlation map (raft_tiled_index), padded positions zero: column n of the
// GEMM below is then float n of the tiled per-query map, so the GEMM epilogue writes whole 128-byte tiles.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fmap_tile_level0_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                                               int h, int w, int tiles_x, int map, int64_t tot_rows_per_b,
                                                               int c4, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % c4);
    int64_t r = i / c4;
    const int n = (int)(r % map);
    const int64_t b = r / map;
    int y, x;
    raft_untiled_yx(n, tiles_x, &y, &x);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y < h && x < w) v = src[((b * h + y) * (int64_t)w + x) * c4 + c];
    dst[(b * tot_rows_per_b + n) * c4 + c] = v;
}

__global__ void __launch_bounds__(256) fmap_pool_kernel(f32x4 *__restrict__ ws, int64_t tot_rows_per_b, int c4,
                                                        int64_t src_off, int stx, int64_t dst_off, int dh, int dw,
                                                        int dtx, int dmap, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % c4);
    int64_t r = i / c4;
    const int n = (int)(r % dmap);
    const int64_t b = r / dmap;
    int y, x;
    raft_untiled_yx(n, dtx, &y, &x);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (y < dh && x < dw) {
        const f32x4 *s = ws + (b * tot_rows_per_b + src_off) * c4 + c;
        f32x4 v00 = s[(int64_t)raft_tiled_index(2 * y, 2 * x, stx) * c4];
        f32x4 v01 = s[(int64_t)raft_tiled_index(2 * y, 2 * x + 1, stx) * c4];
        f32x4 v10 = s[(int64_t)raft_tiled_index(2 * y + 1, 2 * x, stx) * c4];
        f32x4 v11 = s[(int64_t)raft_tiled_index(2 * y + 1, 2 * x + 1, stx) * c4];
        o = ((v00 + v01) + (v10 + v11)) * 0.25f;
    }
    ws[(b * tot_rows_per_b + dst_off + n) * c4 + c] = o;
}

extern "C" int raft_fmap_pyramid_f32(const float *fmap2, int B, int h, int w, int C, int levels,
                                     float *fmap2_pyr, void *stream) {
    RAFT_REQUIRE_PTR(fmap2);
    RAFT_REQUIRE_PTR(fmap2_pyr);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0, RAFT_E_SHAPE);
    RAFT_REQUIRE(C % 4 == 0, RAFT_E_UNSUPPORTED);
    RAFT_REQUIRE(levels >= 1 && levels <= RAFT_MAX_LEVELS, RAFT_E_UNSUPPORTED);
    RAFT_REQUIRE(raft_aligned16(fmap2) && raft_aligned16(fmap2_pyr), RAFT_E_ALIGN);
    hipStream_t s = (hipStream_t)stream;
    const int c4 = C / 4;
    const int64_t tot = pyr_cols(h, w, levels);
    {
        const int map = raft_map_floats(h, w);
        int64_t total = (int64_t)B * map * c4;
        fmap_tile_level0_kernel<<<raft_ceil_div(total, 256), 256, 0, s>>>(
            (const f32x4 *)fmap2, (f32x4 *)fmap2_pyr, h, w, raft_tiles_x(w), map, tot, c4, total);
    }
    int64_t src_off = 0;
    int sh = h, sw = w;
    for (int l = 1; l < levels; ++l) {
        int dh = sh / 2, dw = sw / 2;
        RAFT_REQUIRE(dh >= 1 && dw >= 1, RAFT_E_SHAPE);
        int64_t dst_off = src_off + raft_map_floats(sh, sw);
        const int dmap = raft_map_floats(dh, dw);
        int64_t total = (int64_t)B * dmap * c4;
        fmap_pool_kernel<<<raft_ceil_div(total, 256), 256, 0, s>>>((f32x4 *)fmap2_pyr, tot, c4, src_off,
                                                                    raft_tiles_x(sw), dst_off, dh, dw,
                                                                    raft_tiles_x(dw), dmap, total);
        src_off = dst_off;
        sh = dh;
        sw = dw;
    }
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// corr_build: NT GEMM on v_mfma_f32_32x32x2_f32 (exact fp32).  128x128 output tile per workgroup
// (4 waves, 2x2, each 64x64 = 2x2 MFMA tiles), BK = 32, register-staged double-buffered LDS.
// A = fmap1[b] (N x C), B = fmap2_pyr[b] (T x C); both K-contiguous, so both tiles are
// [128 rows][32 k] with a 36-float row stride (conflict-free ds_read_b128 fragments).
// The k index inside an 8-wide sub-step is permuted between the two wave halves (lanes 0-31 take
// k = 0..3, lanes 32-63 take k = 4..7 of the sub-step); A and B use the same permutation, so the
// sum is unchanged.
// ------------------------------------------------------------------------------------------------
struct CorrGemmArgs {
    const float *a;        // fmap1  (B, N, C)
    const float *bmat;     // fmap2 pyramid workspace (B, T, C)
    float *pyr;            // corr pyramid
    PyramidGeom g;
    int64_t col_off[RAFT_MAX_LEVELS + 1];   // first column of each level inside T
    int N, T, C;
    float sqrt_c, rcp_sqrt_c;
    int rcp_exact;   // sqrt(C) is a power of two
    int xcd_rm, xcd_rn, xcd_maxreg, tiles_m, tiles_n, xcd_gw;   // XCD-aware tile order (xcd_rm == 0: plain 3-D grid)
    // pool1: pyramid level 1 has NO columns in the GEMM (col_off[1] == col_off[2]); it is pooled from the level-0 accumulators in the
    // epilogue.  brow_skip = rows of the fmap2 pyramid workspace to skip behind level 0 (its level-1 rows), tx0 = level-0 tiles per row.
    int pool1, brow_skip, tx0;
};

constexpr int CG_BM = 128, CG_BN = 128, CG_BK = 32, CG_LD = 36;
#ifndef RAFT_GEMM_ABL
#define RAFT_GEMM_ABL 0   // tools/ablate/lookup_layout.hip: 1 = no global loads / LDS writes after the first K step, 2 = no epilogue stores, 4 = no MFMAs
#endif

// XCD-aware tile order: tile `idx` of XCD `xcd` (idx = batch element * xcd_maxreg + position inside the XCD's region).  The
// region is walked in strips of xcd_gw n tiles, each m-major -- the strip's B panels (gw x 128 KB) are reused by every m row at
// once and the region's A panels (mh x 128 KB) return after mh * gw tiles, i.e. before the ~64 KB per tile of volume the L2
// writes in between has pushed them out.  false: the position lies beyond this (smaller) region.
__device__ __forceinline__ bool corr_xcd_tile(const CorrGemmArgs &p, int xcd, int idx, int &b, int &m0, int &n0) {
    b = idx / p.xcd_maxreg;
    const int t = idx - b * p.xcd_maxreg;
    const int rm = xcd / p.xcd_rn, rn = xcd - rm * p.xcd_rn;
    const int tm_lo = rm * p.tiles_m / p.xcd_rm, tm_hi = (rm + 1) * p.tiles_m / p.xcd_rm;
    const int tn_lo = rn * p.tiles_n / p.xcd_rn, tn_hi = (rn + 1) * p.tiles_n / p.xcd_rn;
    const int nw = tn_hi - tn_lo, mh = tm_hi - tm_lo;
    if (nw <= 0 || t >= mh * nw) return false;
    const int gw = p.xcd_gw, g = t / (mh * gw), t_in = t - g * mh * gw;
    const int wg = min(gw, nw - g * gw);
    m0 = (tm_lo + t_in / wg) * CG_BM;
    n0 = (tn_lo + g * gw + t_in % wg) * CG_BN;
    return true;
}

__global__ void __launch_bounds__(256) corr_gemm_kernel(CorrGemmArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (CG_BM + CG_BN) * CG_LD];
    float *sA = smem;
    float *sB = smem + 2 * CG_BM * CG_LD;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, half = lane >> 5, l31 = lane & 31;
    // Workgroup -> tile.  xcd_rm == 0: the plain (n tile, m tile, batch) grid.  Otherwise (1-D grid) XCD-aware: the hardware
    // places workgroup i on XCD i % 8, and each XCD has an L2 of its own (4 MB) -- so XCD x owns one REGION of the (m tile, n tile)
    // plane of every batch element (xcd_rm x xcd_rn = 8 regions: 7 x 19 tiles each at 448x512, corr_xcd_tile): over a batch
    // element an XCD reads A/4 + B/2 = 3.4 MB instead of (nearly) all of A and B.
    // (Two workgroups share a CU.  Delaying half of the first wave of workgroups by 11 - 30 us, so that co-resident pairs run
    // out of phase, changed nothing -- 358 - 364 against 358 - 367 us, profiles/r10i_corr_stagger.txt: lock-step is not what
    // keeps the MFMA pipe at 66 %.)
    int b, m0, n0;
    if (p.xcd_rm == 0) {
        b = blockIdx.z;
        m0 = blockIdx.y * CG_BM;
        n0 = blockIdx.x * CG_BN;
    } else if (!corr_xcd_tile(p, blockIdx.x & 7, blockIdx.x >> 3, b, m0, n0)) {
        return;                                                 // workgroup-uniform: region smaller than the largest one
    }
    const float *A = p.a + (int64_t)b * p.N * p.C;
    const float *Bm = p.bmat + (int64_t)b * (p.T + p.brow_skip) * p.C;

    // staging assignment: 128 rows x 8 float4 per operand tile -> 4 chunks per thread per operand
    const int srow = tid >> 3, sc4 = tid & 7;
    f32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = srow + 32 * i;
            // rows beyond the edge are clamped, not branched around (their products are never stored):
            // a conditional load becomes a branch + s_waitcnt per chunk and serialises the loads
            const int ma = min(m0 + r, p.N - 1);
            int nb = min(n0 + r, p.T - 1);
            nb += nb >= (int)p.col_off[1] ? p.brow_skip : 0;       // pool1: the workspace's level-1 rows take no part
            ra[i] = *(const f32x4 *)(A + (int64_t)ma * p.C + k0 + sc4 * 4);
            rb[i] = *(const f32x4 *)(Bm + (int64_t)nb * p.C + k0 + sc4 * 4);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = srow + 32 * i;
            *(f32x4 *)(sA + buf * CG_BM * CG_LD + r * CG_LD + sc4 * 4) = ra[i];
            *(f32x4 *)(sB + buf * CG_BN * CG_LD + r * CG_LD + sc4 * 4) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.C / CG_BK;
    gload(0);
    lstore(0);
    raft_barrier_lds();
    for (int s = 0; s < nk; ++s) {
        const int buf = s & 1;
        if (s + 1 < nk && !(RAFT_GEMM_ABL & 1)) gload((s + 1) * CG_BK);
        const float *cA = sA + buf * CG_BM * CG_LD;
        const float *cB = sB + buf * CG_BN * CG_LD;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kq = 2 * kk + half;
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *(const f32x4 *)(cA + (wm * 64 + t * 32 + l31) * CG_LD + kq * 4);
                fb[t] = *(const f32x4 *)(cB + (wn * 64 + t * 32 + l31) * CG_LD + kq * 4);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (RAFT_GEMM_ABL & 4)
                            acc[i][j][r] += fa[i][r] * fb[j][r];
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][r], fb[j][r], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nk && !(RAFT_GEMM_ABL & 1)) lstore(buf ^ 1);
        raft_barrier_lds();
    }

    // epilogue: lane owns column n (a target position of some level), 16 rows (queries) per tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        if (n >= p.T) continue;
        if ((RAFT_GEMM_ABL & 2) && acc[0][j][0] != 12345.678f) continue;
        int lvl = 0;
#pragma unroll
        for (int l = 1; l < RAFT_MAX_LEVELS; ++l)
            if (l < p.g.levels && n >= p.col_off[l]) lvl = l;
        const int64_t map = p.g.map[lvl];     // column n - col_off = float index inside the tiled map
        float *base = p.pyr + p.g.off[lvl] + (int64_t)b * p.N * map + (n - p.col_off[lvl]);
        // corr.py:161 divides by sqrt(C); for C a power of four (256, 64) the reciprocal is exact and the product is the same
        // float without the division sequence.  (Non-temporal stores measured no different in time or fabric reads:
        // profiles/r10c_corr_build_ab.txt.)  Tiles that lie inside N (all of them at 448 x 512: 3584 = 28 x 128) take the
        // branch-free path: one lane base, compile-time row offsets.
        if (m0 + CG_BM <= p.N && p.rcp_exact) {   // workgroup-uniform
            float *row0 = base + (int64_t)(m0 + wm * 64 + 4 * half) * map;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) row0[(int64_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * map] = acc[i][j][r] * p.rcp_sqrt_c;
            continue;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < p.N) base[(int64_t)m * map] = p.rcp_exact ? acc[i][j][r] * p.rcp_sqrt_c : acc[i][j][r] / p.sqrt_c;
            }
        }
    }
    // ---- pyramid level 1 pooled from the level-0 accumulators (reference corr.py:106-114: avg_pool2d of the level below).
    // The wave's 64 columns are two level-0 map tiles (tx even, tx + 1) of one tile row ty: lane l31 = (y % 4) * 8 + x % 8 holds the
    // target (y, x) of each.  Two DPP adds (lane ^ 1: the x pair; lane ^ 8: the y pair) leave the 2x2 sum in all four lanes of a
    // group, so the four lanes store it for four DIFFERENT query rows (replica rho -> register 4 q + rho): 16 dword stores per lane
    // instead of 64.  Level-1 tile (ty / 2, tx / 2), rows 2 (ty & 1) + (y % 4) / 2, columns 4 (tx & 1) + (x % 8) / 2: the wave writes
    // one 64-byte half tile per query, the workgroup of tile row ty ^ 1 the other half.
    const int nw0 = n0 + wn * 64;
    if (p.pool1 && nw0 + 64 <= (int)p.col_off[1] && !(RAFT_GEMM_ABL & 2)) {   // wave-uniform
        const int t = nw0 >> 5, ty = t / p.tx0, tx = t - ty * p.tx0;           // level-0 tile of j = 0 (tx even)
        const int tx1n = p.tx0 >> 1;
        const int rho = (l31 & 1) | ((l31 >> 2) & 2);
        const int64_t map1 = p.g.map[1];
        float *l1 = p.pyr + p.g.off[1] + (int64_t)b * p.N * map1 + ((ty >> 1) * tx1n + (tx >> 1)) * 32 + ((ty & 1) * 2 + (l31 >> 4)) * 8 +
                    ((l31 & 7) >> 1);
        const float sc = p.rcp_sqrt_c * 0.25f;                                  // exact: pool1 requires rcp_exact
        const bool whole = m0 + CG_BM <= p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float s4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][4 * q + e];
                        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1, 0, 3, 2]
                        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));   // row_ror 8
                        s4[e] = v;
                    }
                    const float v = rho == 0 ? s4[0] : (rho == 1 ? s4[1] : (rho == 2 ? s4[2] : s4[3]));
                    const int m = m0 + wm * 64 + i * 32 + rho + 8 * q + 4 * half;
                    if (whole || m < p.N) l1[(int64_t)m * map1 + j * 4] = v * sc;
                }
    }
}

// Round 4, measured and not kept (profiles/r10j_gemm_ablation.txt, r10k_corr_pipe_ab.txt): the phases of this kernel add up --
// MFMA + fragment reads alone 273 us (131 TF) at 4 pairs, operand loads +44, volume stores +55 -- and two co-resident workgroups
// do not hide them for each other.  A persistent variant (one workgroup per CU walking its XCD's tiles, the 64 stores per lane of
// a finished tile issued under the next tile's first four K steps, the next tile's first stage fetched under the last K step;
// bit-identical volume) was SLOWER, 387 against 357 us: gfx9's vmcnt counts stores and loads in one queue, so a wave that has
// stores in flight waits for them whenever it waits for its next operand tile.
extern "C" int raft_corr_build_f32(const float *fmap1, const float *fmap2, int B, int h, int w, int C,
                                   int levels, float *pyr, const int64_t *level_offsets, float *workspace,
                                   void *stream) {
    RAFT_REQUIRE_PTR(fmap1);
    RAFT_REQUIRE_PTR(fmap2);
    RAFT_REQUIRE_PTR(pyr);
    RAFT_REQUIRE_PTR(level_offsets);
    RAFT_REQUIRE_PTR(workspace);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0, RAFT_E_SHAPE);
    RAFT_REQUIRE(C % CG_BK == 0, RAFT_E_UNSUPPORTED);
    RAFT_REQUIRE(raft_aligned16(fmap1) && raft_aligned16(fmap2) && raft_aligned16(workspace), RAFT_E_ALIGN);
    CorrGemmArgs a;
    int rc = raft_make_geom(h, w, levels, level_offsets, &a.g);
    if (rc != RAFT_OK) return rc;
    rc = raft_fmap_pyramid_f32(fmap2, B, h, w, C, levels, workspace, stream);
    if (rc != RAFT_OK) return rc;
    a.a = fmap1;
    a.bmat = workspace;
    a.pyr = pyr;
    a.N = h * w;
    a.C = C;
    a.sqrt_c = sqrtf((float)C);
    a.rcp_sqrt_c = 1.0f / a.sqrt_c;
    {
        int e = 0;
        a.rcp_exact = frexpf(a.sqrt_c, &e) == 0.5f;   // mantissa 0.5 <=> power of two: x / 2^k == x * 2^-k exactly
    }
    // Level 1 from the level-0 accumulators (RAFT_CORR_POOL, default on) where every level-1 tile is covered by whole level-0 tiles
    // of one workgroup row: even map sizes, an even number of level-0 tiles per row and column, exact 1 / sqrt(C).
    a.tx0 = raft_tiles_x(w);
    const int ty0 = (int)(a.g.map[0] / 32) / a.tx0;
    a.pool1 = levels >= 2 && raft_opt(RAFT_OPT_CORR_POOL, 1) != 0 && a.rcp_exact && h % 2 == 0 && w % 2 == 0 && a.tx0 % 2 == 0 && ty0 % 2 == 0;
    a.brow_skip = a.pool1 ? a.g.map[1] : 0;
    int64_t t = 0;
    for (int l = 0; l < RAFT_MAX_LEVELS + 1; ++l) a.col_off[l] = 0;
    for (int l = 0; l < levels; ++l) {
        a.col_off[l] = t;
        t += (a.pool1 && l == 1) ? 0 : a.g.map[l];     // pooled: level 1 has no GEMM columns (col_off[1] == col_off[2])
    }
    a.col_off[levels] = t;
    a.T = (int)t;
    a.tiles_m = (int)raft_ceil_div(a.N, CG_BM);
    a.tiles_n = (int)raft_ceil_div(a.T, CG_BN);
    a.xcd_rm = a.xcd_rn = a.xcd_maxreg = 0;
    a.xcd_gw = 2;   // fabric reads per pair at 8 pairs: plain grid 91 MB, regions walked n-fastest 63, strips of 8 / 4 / 2 tiles 43 / 38 / 36 (profiles/r10h_corr_build_ab.txt)
    const int xcd_opt = raft_opt(RAFT_OPT_CORR_XCD, 1);       // 0: plain grid; 1: regions walked in 2-tile strips; n >= 2: strips of n tiles
    if (xcd_opt > 1) a.xcd_gw = xcd_opt;
    if (xcd_opt != 0 && a.tiles_m * a.tiles_n >= 64) {   // small maps: nothing to gain, plain grid
        a.xcd_rm = a.tiles_m >= 4 ? 4 : (a.tiles_m >= 2 ? 2 : 1);
        a.xcd_rn = 8 / a.xcd_rm;
        int mx = 0;
        for (int rm = 0; rm < a.xcd_rm; ++rm)
            for (int rn = 0; rn < a.xcd_rn; ++rn) {
                const int mh = (rm + 1) * a.tiles_m / a.xcd_rm - rm * a.tiles_m / a.xcd_rm;
                const int nw = (rn + 1) * a.tiles_n / a.xcd_rn - rn * a.tiles_n / a.xcd_rn;
                mx = mh * nw > mx ? mh * nw : mx;
            }
        a.xcd_maxreg = mx;
        RAFT_REQUIRE((int64_t)8 * mx * B < ((int64_t)1 << 31), RAFT_E_UNSUPPORTED);
        corr_gemm_kernel<<<dim3(8 * mx * B), 256, 0, (hipStream_t)stream>>>(a);
    } else {
        dim3 grid(a.tiles_n, a.tiles_m, B);
        corr_gemm_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
    }
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// corr_lookup
// ------------------------------------------------------------------------------------------------
struct LookupArgs {
    const float *pyr;
    const float *coords;   // (nq, 2) xy
    float *out;
    PyramidGeom g;
    int64_t nq;
    int ld_out;
};

// Strip kernel.  SP = 4 * (2r+1) threads serve one query from start to finish (r = 4: 36 threads per query,
// QB = 7 queries per 256-thread workgroup), so every index decomposition is a compile-time constant of the
// thread and nothing is looked up through tables:
//   * the thread loads its query's coordinate, derives the footprint origins of the levels in registers and
//     issues its 12 clamped footprint gathers at once (one dependent load after the coordinate, no barrier
//     first); the maps are 4x8-tiled (common.h), so a footprint pulls ~7 lines of a large map, not ~13;
//   * under those loads it evaluates the x tap and the y tap of its own (level, offset) pair
//     {i0 - origin, i1 - origin, w0, w1} as LDS byte offsets; the x tap stays in registers (the same thread
//     consumes it), the y tap goes to LDS; then the gathered footprint values are parked in LDS; ONE barrier;
//   * it then owns the strip (level l, x-offset a) = 2r+1 consecutive output channels: per output one
//     broadcast y-tap read, four footprint reads, 4 + 7 flops in the reference's order (bit-exact).
// STAGE 0: each thread stores its strip directly (2r+1 dwords, 4(2r+1) bytes apart across lanes);
// STAGE 1: the strips are transposed through LDS (aliasing the footprints, two more barriers) and written as
//          whole rows of 16-byte stores (needs levels == 4, ld_out % 4 == 0, a 16-byte aligned `out`).
// LDS = QB * (4 * FP * 4 + SP * 16) bytes = 15.2 KB at r = 4 and 58 VGPRs: 8 workgroups per CU, so a 448x512
// batch of 4 (2048 workgroups) is resident in a single round.  History (profiles/r03*): a table-driven
// 7-queries-per-workgroup kernel ran 18.6 us at B = 4 (854 VALU instructions per wave, 7 workgroups per CU);
// this one 12.0-12.9 us (367); a double-buffered multi-batch variant of it measured no better (13.4 us): the
// kernel is bound by the lines its gathers pull, not by phase lock-step.
#ifndef RAFT_LOOKUP_ABL
#define RAFT_LOOKUP_ABL 0   // tools/ablate/lookup_layout.hip can build this file with pieces of the kernel switched off (-DRAFT_LOOKUP_ABL=bits)
#endif

template <int R>
struct StripCfg {
    static constexpr int L = 4, D = 2 * R + 1, FW = 2 * R + 2, FP = FW * FW, SP = L * D, QB = 256 / SP;
    static constexpr int NR = (FP + SP - 1) / SP, NOUT = L * D * D;
};

// origins of the footprints + the clamped gathers of this thread's slots (all issued back to back)
template <int R>
__device__ __forceinline__ void strip_gather(const LookupArgs &p, int64_t q0, int qc, float2 cq, int j,
                                             float (&v)[4][StripCfg<R>::NR], int (&org)[4][2]) {
    using G = StripCfg<R>;
    int fyx[G::NR];
#pragma unroll
    for (int r = 0; r < G::NR; ++r) {
        const int s = min(j + G::SP * r, G::FP - 1);
        fyx[r] = ((s / G::FW) << 8) | (s % G::FW);
    }
#pragma unroll
    for (int l = 0; l < G::L; ++l) {
        org[l][0] = org[l][1] = 0;
#pragma unroll
        for (int r = 0; r < G::NR; ++r) v[l][r] = 0.f;
        if (l < p.g.levels) {                                // workgroup-uniform
            const float sc = 1.0f / (float)(1 << l);        // exact power of two: x * sc == x / 2^l
            const int w = p.g.lw[l], h = p.g.lh[l], tx = p.g.tx[l];
            const int ox = axis_tap(cq.x * sc, -R, w).i0, oy = axis_tap(cq.y * sc, -R, h).i0;
            org[l][0] = ox;
            org[l][1] = oy;
            // scalar 64-bit base of the workgroup's maps + a 32-bit per-lane byte offset (QB maps < 4 GiB)
            const char *img = (const char *)(p.pyr + p.g.off[l] + q0 * (int64_t)p.g.map[l]);
            const unsigned qoff = (unsigned)qc * (unsigned)p.g.map[l];
#pragma unroll
            for (int r = 0; r < G::NR; ++r) {
                const int y = min(oy + (fyx[r] >> 8), h - 1), x = min(ox + (fyx[r] & 255), w - 1);
                if (RAFT_LOOKUP_ABL & 1)
                    v[l][r] = cq.x + (float)(y + x);
                else
                    v[l][r] = *(const float *)(img + 4u * (qoff + (unsigned)raft_tiled_index(y, x, tx)));
            }
        }
    }
}

// x and y taps of (level l, offset d) relative to the footprint origin, as LDS byte offsets
template <int R>
__device__ __forceinline__ void strip_taps(const LookupArgs &p, float2 cq, int l, int d, const int (&org)[4][2],
                                           int4 &tx, int4 &ty) {
#pragma clang fp contract(off)
    using G = StripCfg<R>;
    const float sc = __int_as_float((127 - l) << 23);       // 2^-l
    const int w = max(p.g.lw[0] >> l, 1), h = max(p.g.lh[0] >> l, 1);   // floor-halved sizes: size_l = size_0 >> l
    int ox = org[0][0], oy = org[0][1];
#pragma unroll
    for (int k = 1; k < G::L; ++k) {
        ox = l == k ? org[k][0] : ox;
        oy = l == k ? org[k][1] : oy;
    }
    if (RAFT_LOOKUP_ABL & 8) {
        tx = make_int4(4 * d, 4 * d + 4, __float_as_int(cq.x), __float_as_int(cq.y));
        ty = make_int4(4 * G::FW * d, 4 * G::FW * (d + 1), tx.z, tx.w);
        return;
    }
    const AxisTap ax = axis_tap(cq.x * sc, d - R, w), ay = axis_tap(cq.y * sc, d - R, h);
    tx.x = (ax.i0 - ox) * 4;
    tx.y = (ax.i1 - ox) * 4;
    tx.z = __float_as_int(ax.w0);
    tx.w = __float_as_int(ax.w1);
    ty.x = (ay.i0 - oy) * (4 * G::FW);
    ty.y = (ay.i1 - oy) * (4 * G::FW);
    ty.z = __float_as_int(ay.w0);
    ty.w = __float_as_int(ay.w1);
}

// the strip (l, a): channels l*D*D + a*D + b, b = 0 .. D-1  (a offsets x, b offsets y: corr.py:133-143)
template <int R>
__device__ __forceinline__ void strip_eval(const float *f, const int (*ty4)[4], int4 tx, float (&o)[StripCfg<R>::D]) {
#pragma clang fp contract(off)   // keep mul/add unfused: same roundings as the unfused reference ops
    using G = StripCfg<R>;
    const float wx0 = __int_as_float(tx.z), wx1 = __int_as_float(tx.w);
    const char *f0 = (const char *)f + tx.x, *f1 = (const char *)f + tx.y;
    if (RAFT_LOOKUP_ABL & 2) {
#pragma unroll
        for (int b = 0; b < G::D; ++b) o[b] = wx0 + (float)b;
        return;
    }
    // (Round 4 tried keeping the lower row of tap b in registers for tap b + 1 -- i0(b + 1) == i1(b) away from borders and exact
    // integers: 20 instead of 36 footprint reads per strip behind a wave-uniform ballot.  12.7 against 13.4 us at 4 pairs, but
    // 23.5 - 23.9 against 22.7 us at 8 and 43.8 - 44.4 against 42.7 at 16, where the kernel is judged: the branch per tap keeps
    // the compiler from issuing the strip's LDS reads as one batch.  profiles/r10c_lookup_row_reuse_ab.txt; not kept.)
#pragma unroll
    for (int b = 0; b < G::D; ++b) {
        const int4 ty = *(const int4 *)ty4[b];
        const float wy0 = __int_as_float(ty.z), wy1 = __int_as_float(ty.w);
        const float c00 = wy0 * wx0, c01 = wy0 * wx1, c10 = wy1 * wx0, c11 = wy1 * wx1;
        float t = c00 * *(const float *)(f0 + ty.x) + c01 * *(const float *)(f1 + ty.x);
        t = t + c10 * *(const float *)(f0 + ty.y);
        t = t + c11 * *(const float *)(f1 + ty.y);
        o[b] = t;
    }
}

template <int R, int STAGE>
__global__ void __launch_bounds__(256) corr_lookup_strip_kernel(LookupArgs p) {
    using G = StripCfg<R>;
    constexpr int L = G::L, D = G::D, FP = G::FP, SP = G::SP, QB = G::QB, NR = G::NR, NOUT = G::NOUT;
    static_assert(QB * NOUT <= QB * L * FP, "staged rows alias the footprints");
    __shared__ __attribute__((aligned(16))) float sfp[QB * L * FP];
    __shared__ __attribute__((aligned(16))) int sty[QB * SP][4];
    const int tid = threadIdx.x;
    const int ql = tid / SP, j = tid - ql * SP;
    const int l = j / D, a = j - l * D;
    const int64_t q0 = (int64_t)blockIdx.x * QB;
    const int nq_here = (int)((p.nq - q0) < QB ? (p.nq - q0) : QB);
    const bool active = ql < nq_here;
    if (RAFT_LOOKUP_ABL & 16) return;
    const bool strip = active && l < p.g.levels;
    const int qc = active ? ql : nq_here - 1;               // clamp: every load below is unconditional
    const float2 cq = *(const float2 *)(p.coords + 2 * (q0 + qc));

    float v[L][NR];
    int org[L][2];
    strip_gather<R>(p, q0, qc, cq, j, v, org);
    int4 tx, ty;
    strip_taps<R>(p, cq, l, a, org, tx, ty);
    if (active) *(int4 *)sty[tid] = ty;
#pragma unroll
    for (int k = 0; k < L; ++k)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int s = j + SP * r;
            if (s < FP && active) sfp[(ql * L + k) * FP + s] = v[k][r];
        }
    __syncthreads();

    float o[D];
    if (strip) strip_eval<R>(sfp + (ql * L + l) * FP, sty + ql * SP + l * D, tx, o);
    if ((RAFT_LOOKUP_ABL & 4) && o[0] != 12345.678f) return;
    if (STAGE == 0) {
        if (strip) {
            float *dst = p.out + (q0 + ql) * (int64_t)p.ld_out + j * D;
#pragma unroll
            for (int b = 0; b < D; ++b) dst[b] = o[b];
        }
    } else {
        __syncthreads();                                     // every strip has read its footprints
        if (strip) {
#pragma unroll
            for (int b = 0; b < D; ++b) sfp[ql * NOUT + j * D + b] = o[b];
        }
        __syncthreads();
        static_assert(NOUT % 4 == 0, "rows are copied as 16-byte chunks");
        for (int i = tid; i < nq_here * (NOUT / 4); i += 256) {
            const int qi = i / (NOUT / 4), c4 = i - qi * (NOUT / 4);
            *(f32x4 *)(p.out + (q0 + qi) * (int64_t)p.ld_out + 4 * c4) = *(const f32x4 *)(sfp + qi * NOUT + 4 * c4);
        }
    }
}

template <int R>
static void launch_lookup(const LookupArgs &a, bool staged, hipStream_t s) {
    const int grid = raft_ceil_div(a.nq, StripCfg<R>::QB);
    if (staged)
        corr_lookup_strip_kernel<R, 1><<<grid, 256, 0, s>>>(a);
    else
        corr_lookup_strip_kernel<R, 0><<<grid, 256, 0, s>>>(a);
}

extern "C" int raft_corr_lookup_f32(const float *pyr, const int64_t *level_offsets, const float *coords, int B,
                                    int h, int w, int levels, int radius, float *out, int ld_out, void *stream) {
    RAFT_REQUIRE_PTR(pyr);
    RAFT_REQUIRE_PTR(level_offsets);
    RAFT_REQUIRE_PTR(coords);
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    LookupArgs a;
    int rc = raft_make_geom(h, w, levels, level_offsets, &a.g);
    if (rc != RAFT_OK) return rc;
    const int d = 2 * radius + 1;
    RAFT_REQUIRE(ld_out >= levels * d * d, RAFT_E_SHAPE);
    a.pyr = pyr;
    a.coords = coords;
    a.out = out;
    a.nq = (int64_t)B * h * w;
    a.ld_out = ld_out;
    hipStream_t s = (hipStream_t)stream;
    // rows staged through LDS and written as 16-byte stores wherever the output allows it, direct strip stores otherwise
    const bool staged = levels == 4 && (ld_out & 3) == 0 && raft_aligned16(out);
    if (radius == 4)
        launch_lookup<4>(a, staged, s);
    else if (radius == 3)
        launch_lookup<3>(a, staged, s);
    else
        return RAFT_E_UNSUPPORTED;
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Lookup fused into the first convolution of the motion encoder:
//   cor1 = relu(convc1(CorrBlock.retrieve(coords1)))        [reference corr.py:116-152 + update.py:91, 98]
// The (B, h, w, 324) lookup output never exists in HBM: a workgroup owns 28 queries (56 x 64 = 128 x 28: the feature maps
// of 448 x 512 frames cut into 512 x B / 4 equal workgroups, two per CU at B = 4 with no ragged last round), runs the
// strip lookup of corr_lookup_strip_kernel for them in two rounds of fourteen queries -- the SAME device functions, so the
// window values are bit-identical to raft_corr_lookup_f32 -- and leaves the values as rows of an LDS tile
// [32 queries][4 levels x 84]: K order = level-major, each level's 81 channels padded to 84 (the weights are packed to
// match, tf_raft_amd/packing.py pack_convc1_fused; rows 28..31 and the pad columns are zero).  The tile is then the A
// operand of a [32 x 336] . [336 x 256] fp32-MFMA product (16x16x4; 512 threads, wave w owns channels 32 w .. 32 w + 31:
// 2 row blocks x 2 column blocks; weight fragments straight from L2, two 16-channel blocks ahead), bias + relu in the
// epilogue.  The gathers of BOTH rounds are issued before anything else (one exposed round trip per workgroup); with two
// workgroups = 16 waves per CU one workgroup's MFMA phase runs under the other's lookup phase.  Versus lookup + convc1 as two kernels this removes the 18.6 MB (B = 4)
// write and re-read of the lookup output and one kernel boundary per iteration.
// ------------------------------------------------------------------------------------------------
struct FusedLookupArgs {
    LookupArgs lk;          // lk.out / lk.ld_out unused
    const float *wp;        // [84 k-quads][npad][4]
    const float *bias;      // [npad]
    float *out;             // (nq, ldo): cor1
    int ldo, npad, nvalid;
};

constexpr int FL_ROUNDS = 2, FL_QB = 14, FL_THREADS = 512, FL_LVLK = 84, FL_K = 4 * FL_LVLK, FL_LDA = FL_K + 8, FL_ROWS = 32;   // row stride 344: conflict-free b128 A-fragment reads (tools/bank_check.py lane groups; 340 is 2-way)

// ABL (diagnostics, tools/fused_probe.py): 0 = the kernel; 1 = no lookup phase (A tile left zero); 2 = no MFMA phase
template <int R, int ABL = 0>
__global__ void __launch_bounds__(FL_THREADS, 1) lookup_convc1_kernel(FusedLookupArgs p) {
    using G = StripCfg<R>;
    constexpr int L = G::L, D = G::D, FP = G::FP, SP = G::SP, NR = G::NR;
    static_assert(R == 4 && FL_QB * SP <= FL_THREADS && D * D <= FL_LVLK, "14 queries per round, 81 channels per level padded to 84");
    constexpr int QW = FL_QB * FL_ROUNDS;                                // 28 queries per workgroup
    __shared__ __attribute__((aligned(16))) float sA[FL_ROWS * FL_LDA];
    __shared__ __attribute__((aligned(16))) float sfp[FL_QB * L * FP];
    __shared__ __attribute__((aligned(16))) int sty[FL_QB * SP][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ql = min(tid / SP, FL_QB - 1), j = tid - (tid / SP) * SP;  // threads 504..511 shadow query 13 (never stored)
    const bool lk_thread = tid < FL_QB * SP;
    const int l = j / D, a = j - l * D;
    const int64_t qbase = (int64_t)blockIdx.x * QW;

    if (ABL == 1) {
        for (int i = tid; i < FL_ROWS * FL_LDA / 4; i += FL_THREADS) ((f32x4 *)sA)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
    }
    // ---- phase A: the strip lookup, two rounds of fourteen queries; the gathers of BOTH rounds are issued up front
    float v[FL_ROUNDS][L][NR];
    int org[FL_ROUNDS][L][2];
    float2 cq[FL_ROUNDS];
#pragma unroll
    for (int rd = 0; rd < (ABL == 1 ? 0 : FL_ROUNDS); ++rd) {
        const int64_t q0 = qbase + rd * FL_QB;
        const int64_t left = p.lk.nq - q0;
        const int nq_here = (int)(left < FL_QB ? (left < 1 ? 1 : left) : FL_QB);   // >= 1: loads stay in range
        const int64_t q0c = left < 1 ? p.lk.nq - 1 : q0;                           // a round past the end re-reads the last query
        const int qc = ql < nq_here ? ql : nq_here - 1;
        cq[rd] = *(const float2 *)(p.lk.coords + 2 * (q0c + qc));
        strip_gather<R>(p.lk, q0c, qc, cq[rd], j, v[rd], org[rd]);
    }
    if (ABL != 1)
        for (int i = tid; i < FL_ROWS * FL_LDA / 4; i += FL_THREADS) ((f32x4 *)sA)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rd = 0; rd < (ABL == 1 ? 0 : FL_ROUNDS); ++rd) {
        const int64_t left = p.lk.nq - (qbase + rd * FL_QB);
        const int nq_here = (int)(left < 0 ? 0 : (left < FL_QB ? left : FL_QB));
        const bool active = lk_thread && ql < nq_here;
        int4 tx, ty;
        strip_taps<R>(p.lk, cq[rd], l, a, org[rd], tx, ty);
        if (rd) __syncthreads();                                           // the previous round's strips have been read
        if (active) *(int4 *)sty[ql * SP + j] = ty;
#pragma unroll
        for (int k = 0; k < L; ++k)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int s = j + SP * r;
                if (s < FP && active) sfp[(ql * L + k) * FP + s] = v[rd][k][r];
            }
        __syncthreads();
        if (active) {
            float o[D];
            strip_eval<R>(sfp + (ql * L + l) * FP, sty + ql * SP + l * D, tx, o);
            float *dst = sA + (rd * FL_QB + ql) * FL_LDA + l * FL_LVLK + a * D;
#pragma unroll
            for (int b = 0; b < D; ++b) dst[b] = o[b];
        }
    }
    __syncthreads();

    // ---- phase B: cor1 tile = relu(A . W + bias); wave wv owns channels 32 wv .. 32 wv + 31 (2 row x 2 column blocks)
    const int r16 = lane & 15, g4 = lane >> 4;
    const int n0 = wv * 32;
    if (ABL == 2) {   // keep the lookup alive: one value per thread
        if (tid < FL_ROWS && qbase + tid < p.lk.nq) p.out[(qbase + tid) * p.ldo] = sA[tid * FL_LDA + 5] + sA[tid * FL_LDA + 200];
        return;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 *wq = (const f32x4 *)p.wp + (int64_t)g4 * p.npad + n0 + r16;      // k-quad 4 blk + g4, column n0 + 16 cb + r16
    constexpr int NBLK = FL_K / 16, PF = 3;                                 // 21 blocks of 16 channels; fragments PF blocks ahead
    f32x4 fb[PF][2];
#pragma unroll
    for (int u = 0; u < PF - 1; ++u)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) fb[u][cb] = wq[(int64_t)u * 4 * p.npad + cb * 16];
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        if (blk + PF - 1 < NBLK) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) fb[(blk + PF - 1) % PF][cb] = wq[(int64_t)(blk + PF - 1) * 4 * p.npad + cb * 16];
        }
        f32x4 fa[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) fa[rb] = *(const f32x4 *)(sA + (rb * 16 + r16) * FL_LDA + blk * 16 + g4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[rb][e], fb[blk % PF][cb][e], acc[rb][cb], 0, 0, 0);
    }
    // D[row = query 16 rb + 4 g4 + e][col = channel n0 + 16 cb + r16]
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int n = n0 + cb * 16 + r16;
        const float bias = p.bias[n];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = rb * 16 + g4 * 4 + e;
                const int64_t q = qbase + row;
                if (row < QW && q < p.lk.nq && n < p.nvalid) p.out[q * p.ldo + n] = fmaxf(acc[rb][cb][e] + bias, 0.f);
            }
    }
}

extern "C" int raft_lookup_convc1_f32(const float *pyr, const int64_t *level_offsets, const float *coords, int B, int h, int w,
                                      const float *wp, const float *bias, int npad, int nvalid, float *out, int ldo,
                                      void *stream) {
    RAFT_REQUIRE_PTR(pyr);
    RAFT_REQUIRE_PTR(level_offsets);
    RAFT_REQUIRE_PTR(coords);
    RAFT_REQUIRE_PTR(wp);
    RAFT_REQUIRE_PTR(bias);
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    RAFT_REQUIRE(npad == 256 && nvalid > 0 && nvalid <= npad && ldo >= nvalid, RAFT_E_UNSUPPORTED);
    RAFT_REQUIRE(raft_aligned16(wp), RAFT_E_ALIGN);
    FusedLookupArgs a;
    RAFT_TRY(raft_make_geom(h, w, 4, level_offsets, &a.lk.g));
    a.lk.pyr = pyr;
    a.lk.coords = coords;
    a.lk.out = nullptr;
    a.lk.nq = (int64_t)B * h * w;
    a.lk.ld_out = 0;
    a.wp = wp;
    a.bias = bias;
    a.out = out;
    a.ldo = ldo;
    a.npad = npad;
    a.nvalid = nvalid;
    const int grid = raft_ceil_div(a.lk.nq, FL_QB * FL_ROUNDS);
#ifdef RAFT_FUSED_PROBE
    // diagnostic builds only (tools/fused_probe.py compiles its own copy with -DRAFT_FUSED_PROBE): RAFT_LOOKUP_FUSED
    // 11 / 12 select the phase-ablated kernels, which do NOT compute the function.  The library never contains them:
    // every setting of every switch of the shipped .so computes the same result (include/raft_hip.h).
    const int abl = raft_opt(RAFT_OPT_LOOKUP_FUSED, 1);
    if (abl == 11) {
        lookup_convc1_kernel<4, 1><<<grid, FL_THREADS, 0, (hipStream_t)stream>>>(a);
        return raft_launch_status();
    }
    if (abl == 12) {
        lookup_convc1_kernel<4, 2><<<grid, FL_THREADS, 0, (hipStream_t)stream>>>(a);
        return raft_launch_status();
    }
#endif
    lookup_convc1_kernel<4, 0><<<grid, FL_THREADS, 0, (hipStream_t)stream>>>(a);
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// standalone bilinear_sampler and coords_grid
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bilinear_sampler_kernel(const float *__restrict__ image,
                                                               const float *__restrict__ coords, int64_t total,
                                                               int h, int w, int kk, float *__restrict__ out) {
#pragma clang fp contract(off)
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t n = i / kk;
    const AxisTap tx = axis_tap(coords[2 * i], 0, w), ty = axis_tap(coords[2 * i + 1], 0, h);
    const float *img = image + n * ((int64_t)h * w);
    const float c00 = ty.w0 * tx.w0, c01 = ty.w0 * tx.w1, c10 = ty.w1 * tx.w0, c11 = ty.w1 * tx.w1;
    float v = c00 * img[ty.i0 * w + tx.i0] + c01 * img[ty.i0 * w + tx.i1];
    v = v + c10 * img[ty.i1 * w + tx.i0];
    v = v + c11 * img[ty.i1 * w + tx.i1];
    out[i] = v;
}

extern "C" int raft_bilinear_sampler_f32(const float *image, const float *coords, int64_t n, int h, int w, int kh,
                                         int kw, float *out, void *stream) {
    RAFT_REQUIRE_PTR(image);
    RAFT_REQUIRE_PTR(coords);
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(n > 0 && h > 0 && w > 0 && kh > 0 && kw > 0, RAFT_E_SHAPE);
    const int64_t total = n * kh * kw;
    bilinear_sampler_kernel<<<raft_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(image, coords, total, h, w,
                                                                                      kh * kw, out);
    return raft_launch_status();
}

__global__ void __launch_bounds__(256) coords_grid_kernel(float2 *__restrict__ coords, int h, int w, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % w), y = (int)((i / w) % h);
    coords[i] = make_float2((float)x, (float)y);
}

extern "C" int raft_coords_grid_f32(float *coords, int B, int h, int w, void *stream) {
    RAFT_REQUIRE_PTR(coords);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    const int64_t total = (int64_t)B * h * w;
    coords_grid_kernel<<<raft_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>((float2 *)coords, h, w, total);
    return raft_launch_status();
}
