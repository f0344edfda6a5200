// Correlation volume build + pyramid lookup for gfx950 (MI355X).
//
//   raft_corr_build_f32   : fp32 MFMA "NT" GEMM  fmap1 (N x C) . fmap2_pyr (T x C)^T / sqrt(C),
//                           T = sum over levels of lh*lw (pooled fmap2), written straight into the
//                           per-level (B*N, lh, lw) maps.   [reference corr.py:100-114, 154-162]
//   raft_corr_lookup_f32  : one wavefront per query pixel; the (2r+2)^2 footprint of each level is
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
        int64_t sz = nq * ch * cw;
        sz = (sz + 3) & ~(int64_t)3;   // keep every level 16-byte aligned
        off += sz;
        ch /= 2;
        cw /= 2;
    }
    level_offsets[levels] = off;
    return RAFT_OK;
}

static int64_t pyr_cols(int h, int w, int levels) {
    int64_t t = 0;
    for (int l = 0; l < levels; ++l) {
        t += (int64_t)h * w;
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
// fmap2 feature pyramid: level l = 2x2 VALID average of level l-1 (NHWC, channels vectorised x4)
// workspace layout per batch element: [level0 (h*w rows) | level1 | ...] x C floats
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fmap_copy_level0_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                                               int64_t rows_per_b, int64_t tot_rows_per_b, int c4,
                                                               int64_t total) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int64_t per_b = rows_per_b * c4;
    int64_t b = i / per_b, r = i % per_b;
    dst[b * tot_rows_per_b * c4 + r] = src[i];
}

__global__ void __launch_bounds__(256) fmap_pool_kernel(f32x4 *__restrict__ ws, int64_t tot_rows_per_b, int c4,
                                                        int64_t src_off, int sh, int sw, int64_t dst_off, int dh,
                                                        int dw, int64_t total) {
    (void)sh;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int c = (int)(i % c4);
    int64_t p = i / c4;
    int x = (int)(p % dw);
    int y = (int)((p / dw) % dh);
    int64_t b = p / ((int64_t)dw * dh);
    const f32x4 *s = ws + (b * tot_rows_per_b + src_off) * c4;
    f32x4 v00 = s[((int64_t)(2 * y) * sw + 2 * x) * c4 + c];
    f32x4 v01 = s[((int64_t)(2 * y) * sw + 2 * x + 1) * c4 + c];
    f32x4 v10 = s[((int64_t)(2 * y + 1) * sw + 2 * x) * c4 + c];
    f32x4 v11 = s[((int64_t)(2 * y + 1) * sw + 2 * x + 1) * c4 + c];
    f32x4 r = ((v00 + v01) + (v10 + v11)) * 0.25f;
    ws[(b * tot_rows_per_b + dst_off + (int64_t)y * dw + x) * c4 + c] = r;
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
        int64_t total = (int64_t)B * h * w * c4;
        fmap_copy_level0_kernel<<<raft_ceil_div(total, 256), 256, 0, s>>>(
            (const f32x4 *)fmap2, (f32x4 *)fmap2_pyr, (int64_t)h * w, tot, c4, total);
    }
    int64_t src_off = 0;
    int sh = h, sw = w;
    for (int l = 1; l < levels; ++l) {
        int dh = sh / 2, dw = sw / 2;
        RAFT_REQUIRE(dh >= 1 && dw >= 1, RAFT_E_SHAPE);
        int64_t dst_off = src_off + (int64_t)sh * sw;
        int64_t total = (int64_t)B * dh * dw * c4;
        fmap_pool_kernel<<<raft_ceil_div(total, 256), 256, 0, s>>>((f32x4 *)fmap2_pyr, tot, c4, src_off, sh, sw,
                                                                    dst_off, dh, dw, total);
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
    float sqrt_c;
};

constexpr int CG_BM = 128, CG_BN = 128, CG_BK = 32, CG_LD = 36;

__global__ void __launch_bounds__(256) corr_gemm_kernel(CorrGemmArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (CG_BM + CG_BN) * CG_LD];
    float *sA = smem;
    float *sB = smem + 2 * CG_BM * CG_LD;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const int m0 = blockIdx.y * CG_BM, n0 = blockIdx.x * CG_BN;
    const float *A = p.a + (int64_t)b * p.N * p.C;
    const float *Bm = p.bmat + (int64_t)b * p.T * p.C;

    // staging assignment: 128 rows x 8 float4 per operand tile -> 4 chunks per thread per operand
    const int srow = tid >> 3, sc4 = tid & 7;
    f32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = srow + 32 * i;
            // rows beyond the edge are clamped, not branched around (their products are never stored):
            // a conditional load becomes a branch + s_waitcnt per chunk and serialises the loads
            const int ma = min(m0 + r, p.N - 1), nb = min(n0 + r, p.T - 1);
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
    __syncthreads();
    for (int s = 0; s < nk; ++s) {
        const int buf = s & 1;
        if (s + 1 < nk) gload((s + 1) * CG_BK);
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
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][r], fb[j][r], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane owns column n (a target position of some level), 16 rows (queries) per tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        if (n >= p.T) continue;
        int lvl = 0;
#pragma unroll
        for (int l = 1; l < RAFT_MAX_LEVELS; ++l)
            if (l < p.g.levels && n >= p.col_off[l]) lvl = l;
        const int64_t map = (int64_t)p.g.lh[lvl] * p.g.lw[lvl];
        float *base = p.pyr + p.g.off[lvl] + (int64_t)b * p.N * map + (n - p.col_off[lvl]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < p.N) base[(int64_t)m * map] = acc[i][j][r] / p.sqrt_c;
            }
        }
    }
}

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
    int64_t t = 0;
    for (int l = 0; l < RAFT_MAX_LEVELS + 1; ++l) a.col_off[l] = 0;
    for (int l = 0; l < levels; ++l) {
        a.col_off[l] = t;
        t += (int64_t)a.g.lh[l] * a.g.lw[l];
    }
    a.col_off[levels] = t;
    a.T = (int)t;
    a.sqrt_c = sqrtf((float)C);
    dim3 grid(raft_ceil_div(a.T, CG_BN), raft_ceil_div(a.N, CG_BM), B);
    corr_gemm_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
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

// v1 (default): the 2 x levels x (2r+1) axis taps {i0, i1, w0, w1} are evaluated ONCE per query by
// 72 lanes and parked in LDS next to the footprint; every output then needs two 16-B tap reads,
// four footprint reads and seven flops.  (v0 below recomputes four axis taps per output and is
// VALU-issue bound; it is kept for A/B timing: RAFT_LOOKUP_V0=1.)
template <int R>
__global__ void __launch_bounds__(256) corr_lookup_kernel(LookupArgs p) {
#pragma clang fp contract(off)   // keep mul/add unfused: same roundings as the unfused reference ops
    constexpr int D = 2 * R + 1, FW = 2 * R + 2, FP = FW * FW, NT = RAFT_MAX_LEVELS * 2 * D;
    __shared__ float sfp[4][RAFT_MAX_LEVELS][FP];
    __shared__ __attribute__((aligned(16))) int stap[4][NT + 8][4];   // {i0 - origin, i1 - origin, w0, w1}
    __shared__ int sorg[4][RAFT_MAX_LEVELS][2];                        // footprint origin (x, y) per level
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    const bool active = q < p.nq;
    const int levels = p.g.levels;
    float cx0 = 0.f, cy0 = 0.f;
    if (active) {
        cx0 = p.coords[2 * q];
        cy0 = p.coords[2 * q + 1];
        // ---- axis taps: entry t = (l*2 + axis)*D + d
        for (int t = lane; t < levels * 2 * D; t += 64) {
            const int l = t / (2 * D), r = t - l * (2 * D);
            const int axis = r / D, d = r - axis * D;
            const float sc = 1.0f / (float)(1 << l);   // exact power of two: x * sc == x / 2^l
            const int size = axis ? p.g.lh[l] : p.g.lw[l];
            const float c = (axis ? cy0 : cx0) * sc;
            const AxisTap org = axis_tap(c, -R, size), tp = axis_tap(c, d - R, size);
            stap[wave][t][0] = tp.i0 - org.i0;
            stap[wave][t][1] = tp.i1 - org.i0;
            stap[wave][t][2] = __float_as_int(tp.w0);
            stap[wave][t][3] = __float_as_int(tp.w1);
            if (d == 0) sorg[wave][l][axis] = org.i0;
        }
    }
    __syncthreads();
    if (active) {
        // ---- footprint staging: (fy, fx) of this lane's two footprint slots do not depend on the level
        const int i1 = lane + 64;
        const int fy0 = lane / FW, fx0 = lane - fy0 * FW;
        const int fy1 = i1 / FW, fx1 = i1 - fy1 * FW;
#pragma unroll
        for (int l = 0; l < RAFT_MAX_LEVELS; ++l) {
            if (l >= levels) break;
            const int w = p.g.lw[l], h = p.g.lh[l];
            const int ox = sorg[wave][l][0], oy = sorg[wave][l][1];
            const float *img = p.pyr + p.g.off[l] + q * ((int64_t)h * w);
            sfp[wave][l][lane] = img[min(oy + fy0, h - 1) * w + min(ox + fx0, w - 1)];
            if (i1 < FP) sfp[wave][l][i1] = img[min(oy + fy1, h - 1) * w + min(ox + fx1, w - 1)];
        }
    }
    __syncthreads();
    if (!active) return;
    const int nout = levels * D * D;
    float *o = p.out + q * (int64_t)p.ld_out;
    for (int c = lane; c < nout; c += 64) {
        const int l = c / (D * D);
        const int t = c - l * (D * D);
        const int a = t / D, b = t - a * D;              // a offsets x, b offsets y (corr.py:133-143)
        const int4 tx = *(const int4 *)stap[wave][(l * 2 + 0) * D + a];
        const int4 ty = *(const int4 *)stap[wave][(l * 2 + 1) * D + b];
        const float wx0 = __int_as_float(tx.z), wx1 = __int_as_float(tx.w);
        const float wy0 = __int_as_float(ty.z), wy1 = __int_as_float(ty.w);
        const float *f = sfp[wave][l];
        const int y0 = ty.x * FW, y1 = ty.y * FW;
        const float c00 = wy0 * wx0, c01 = wy0 * wx1, c10 = wy1 * wx0, c11 = wy1 * wx1;
        float v = c00 * f[y0 + tx.x] + c01 * f[y0 + tx.y];
        v = v + c10 * f[y1 + tx.x];
        v = v + c11 * f[y1 + tx.y];
        o[c] = v;
    }
}

// v2 (default): a WORKGROUP handles QB = 7 queries cooperatively, in three phases separated by barriers:
//   1. axis taps {i0 - origin, i1 - origin, w0, w1} of all QB x levels x 2 x (2r+1) window positions
//      (one item per thread; the query coordinate is a broadcast load) -> LDS;
//   2. the (2r+2)^2 footprints: QB x (2r+2)^2 items per level spread over all 256 threads, every load
//      of every level issued before the first one is consumed (12 independent gathers in flight per
//      thread instead of 2 dependent rounds per wave), clamped addresses so no load is conditional;
//   3. the outputs: QB x levels x (2r+1)^2 items, consecutive threads = consecutive channels of a
//      query (coalesced stores), each from two 16-byte tap reads + four footprint reads in LDS.
// Index decompositions (channel -> level / window position, footprint slot -> row / column) come from
// small LDS tables built once per workgroup.  With QB = 7 a 448x512 batch of 4 is 2048 workgroups =
// exactly the 8 workgroups per CU the chip keeps resident.  (A single-barrier variant with 36 threads
// per query deriving the origins in registers measured slower: 20.2 vs 18.9 us at B = 4.)
template <int R, int QB>
__global__ void __launch_bounds__(256) corr_lookup_wg_kernel(LookupArgs p) {
#pragma clang fp contract(off)   // keep mul/add unfused: same roundings as the unfused reference ops
    constexpr int L = RAFT_MAX_LEVELS, D = 2 * R + 1, FW = 2 * R + 2, FP = FW * FW, NT = L * 2 * D;
    constexpr int NI = (QB * FP + 255) / 256, NOUT = L * D * D;
    __shared__ float sfp[QB][L][FP];
    __shared__ __attribute__((aligned(16))) int stap[QB][NT][4];
    __shared__ int sorg[QB][L][2];
    __shared__ int sdec[NOUT];     // output channel c -> (x-tap entry, y-tap entry, level) packed 10 + 10 + 4 bits
    __shared__ int sfpd[FP];       // footprint slot -> (fy << 8) | fx
    const int tid = threadIdx.x;
    const int64_t q0 = (int64_t)blockIdx.x * QB;
    const int nq_here = (int)((p.nq - q0) < QB ? (p.nq - q0) : QB);
    const int levels = p.g.levels;

    // ---- index tables (the divisions by 81 / 9 / 10 are done once per workgroup, not once per item)
    for (int c = tid; c < NOUT; c += 256) {
        const int l = c / (D * D), t = c - l * (D * D);
        const int a = t / D, b = t - a * D;              // a offsets x, b offsets y (corr.py:133-143)
        sdec[c] = ((l * 2 + 0) * D + a) | (((l * 2 + 1) * D + b) << 10) | (l << 20);
    }
    if (tid < FP) sfpd[tid] = ((tid / FW) << 8) | (tid % FW);

    // ---- phase 1: axis taps; entry t = (l*2 + axis)*D + d
    for (int it = tid; it < QB * NT; it += 256) {
        const int qi = it / NT, t = it - qi * NT;
        const int l = t / (2 * D), r = t - l * (2 * D);
        const int axis = r / D, d = r - axis * D;
        if (qi < nq_here && l < levels) {
            const float cq = p.coords[2 * (q0 + qi) + axis];
            const float sc = 1.0f / (float)(1 << l);   // exact power of two: x * sc == x / 2^l
            const int size = axis ? p.g.lh[l] : p.g.lw[l];
            const float c = cq * sc;
            const AxisTap org = axis_tap(c, -R, size), tp = axis_tap(c, d - R, size);
            stap[qi][t][0] = tp.i0 - org.i0;
            stap[qi][t][1] = tp.i1 - org.i0;
            stap[qi][t][2] = __float_as_int(tp.w0);
            stap[qi][t][3] = __float_as_int(tp.w1);
            if (d == 0) sorg[qi][l][axis] = org.i0;
        }
    }
    __syncthreads();

    // ---- phase 2: footprints (all loads first, then the LDS writes)
    float v[L][NI];
    int qk[NI], fk[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int it = tid + 256 * k;
        int qi = it / FP;
        fk[k] = sfpd[it - qi * FP];
        qk[k] = qi < nq_here ? qi : nq_here - 1;          // clamp: the loads are unconditional
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        if (l < levels) {
            const int w = p.g.lw[l], h = p.g.lh[l];
            const float *lvl = p.pyr + p.g.off[l] + q0 * ((int64_t)h * w);
            const int map = h * w;
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int2 o = *(const int2 *)sorg[qk[k]][l];
                v[l][k] = lvl[qk[k] * map + min(o.y + (fk[k] >> 8), h - 1) * w + min(o.x + (fk[k] & 255), w - 1)];
            }
        }
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        if (l < levels) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int it = tid + 256 * k;
                if (it < QB * FP) (&sfp[0][0][0])[(it / FP) * (L * FP) + l * FP + (it % FP)] = v[l][k];
            }
        }
    }
    __syncthreads();

    // ---- phase 3: outputs
    const int nout = levels * D * D;
    float *obase = p.out + q0 * (int64_t)p.ld_out;
    for (int it = tid; it < nq_here * nout; it += 256) {
        const int qi = it / nout, c = it - qi * nout;
        const int dec = sdec[c];
        const int l = dec >> 20;
        const int4 tx = *(const int4 *)stap[qi][dec & 1023];
        const int4 ty = *(const int4 *)stap[qi][(dec >> 10) & 1023];
        const float wx0 = __int_as_float(tx.z), wx1 = __int_as_float(tx.w);
        const float wy0 = __int_as_float(ty.z), wy1 = __int_as_float(ty.w);
        const float *f = sfp[qi][l];
        const int y0 = ty.x * FW, y1 = ty.y * FW;
        const float c00 = wy0 * wx0, c01 = wy0 * wx1, c10 = wy1 * wx0, c11 = wy1 * wx1;
        float o = c00 * f[y0 + tx.x] + c01 * f[y0 + tx.y];
        o = o + c10 * f[y1 + tx.x];
        o = o + c11 * f[y1 + tx.y];
        obase[qi * p.ld_out + c] = o;
    }
}

template <int R>
__global__ void __launch_bounds__(256) corr_lookup_v0_kernel(LookupArgs p) {
#pragma clang fp contract(off)   // keep mul/add unfused: same roundings as the unfused reference ops
    constexpr int D = 2 * R + 1, FW = 2 * R + 2, FP = FW * FW;
    __shared__ float sfp[4][RAFT_MAX_LEVELS][FP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    const bool active = q < p.nq;
    float cx0 = 0.f, cy0 = 0.f;
    if (active) {
        cx0 = p.coords[2 * q];
        cy0 = p.coords[2 * q + 1];
    }
    if (active) {
#pragma unroll
        for (int l = 0; l < RAFT_MAX_LEVELS; ++l) {
            if (l >= p.g.levels) break;
            const float sc = 1.0f / (float)(1 << l);   // exact power of two: x * sc == x / 2^l
            const int w = p.g.lw[l], h = p.g.lh[l];
            const AxisTap tx = axis_tap(cx0 * sc, -R, w);
            const AxisTap ty = axis_tap(cy0 * sc, -R, h);
            const float *img = p.pyr + p.g.off[l] + q * ((int64_t)h * w);
            for (int i = lane; i < FP; i += 64) {
                const int fy = i / FW, fx = i - fy * FW;
                const int yy = min(ty.i0 + fy, h - 1), xx = min(tx.i0 + fx, w - 1);
                sfp[wave][l][i] = img[yy * w + xx];
            }
        }
    }
    __syncthreads();
    if (!active) return;
    const int nout = p.g.levels * D * D;
    float *o = p.out + q * (int64_t)p.ld_out;
    for (int c = lane; c < nout; c += 64) {
        const int l = c / (D * D);
        const int t = c - l * (D * D);
        const int a = t / D, b = t - a * D;              // a offsets x, b offsets y (corr.py:133-143)
        const float sc = 1.0f / (float)(1 << l);
        const int w = p.g.lw[l], h = p.g.lh[l];
        const float cx = cx0 * sc, cy = cy0 * sc;
        const AxisTap ox = axis_tap(cx, -R, w), oy = axis_tap(cy, -R, h);   // footprint origin
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
    const int blocks = raft_ceil_div(a.nq, 4);
    hipStream_t s = (hipStream_t)stream;
    const char *ver = getenv("RAFT_LOOKUP_VERSION");   // A/B timing switch only: 0, 1 = wave-per-query kernels
    const int version = ver ? atoi(ver) : 2;
    constexpr int QB = 7;
    const int wg_blocks = raft_ceil_div(a.nq, QB);
    if (radius == 4) {
        if (version == 0)
            corr_lookup_v0_kernel<4><<<blocks, 256, 0, s>>>(a);
        else if (version == 1)
            corr_lookup_kernel<4><<<blocks, 256, 0, s>>>(a);
        else
            corr_lookup_wg_kernel<4, QB><<<wg_blocks, 256, 0, s>>>(a);
    } else if (radius == 3) {
        if (version == 0)
            corr_lookup_v0_kernel<3><<<blocks, 256, 0, s>>>(a);
        else if (version == 1)
            corr_lookup_kernel<3><<<blocks, 256, 0, s>>>(a);
        else
            corr_lookup_wg_kernel<3, QB><<<wg_blocks, 256, 0, s>>>(a);
    } else {
        return RAFT_E_UNSUPPORTED;
    }
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
