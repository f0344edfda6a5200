// First slice of the training step (reference tf_raft/model.py:126-144 under tf.GradientTape), gfx950:
//
//   raft_sequence_loss_grad_f32    d sequence_loss / d prediction_i                       [losses.py:4-21]
//   raft_corr_lookup_backward_f32  backward of CorrBlock.retrieve + bilinear_sampler: gradient w.r.t. the lookup
//                                  coordinates and w.r.t. the correlation pyramid           [corr.py:116-152, 28-69]
//   raft_conv2d_wgrad_f32          d Conv2D / d kernel and d bias (stride 1, 'same')       [update.py:10-11, 91-95, 138-140]
//   raft_relu_backward_f32         dy * (y > 0)
// (d Conv2D / d input needs no kernel of its own: it is the forward convolution of dy with the spatially flipped,
//  in/out-transposed kernel -- tf_raft_amd/packing.py pack_conv_dgrad + raft_conv2d_f32.)
//
// All reductions are deterministic: no atomics anywhere; a correlation map is owned by its query (one wavefront), weight
// gradients are split over pixel slices into a workspace and added in slice order by a second kernel.
#include "common.h"
#include "lookup_common.h"

// ------------------------------------------------------------------------------------------------
// sequence_loss backward: loss = sum_i w_i * mean over 2*npix elements of m * |p_i - g|, w_i = gamma^(n-i-1)
//   d loss / d p_i[e] = upstream * w_i * m * sign(p_i[e] - g[e]) / (2 * npix)        (sign(0) = 0, as tf.abs)
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int MAX_PRED = 64;
struct LossWeights {
    float w[MAX_PRED];
};

__global__ void __launch_bounds__(256) sequence_loss_grad_kernel(const float2 *__restrict__ gt, const unsigned char *__restrict__ valid,
                                                                  const float2 *__restrict__ preds, int64_t pred_stride, int n_pred,
                                                                  int64_t npix, float max_flow, LossWeights lw,
                                                                  float2 *__restrict__ d_preds) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const float2 g = gt[i];
    const float mag = sqrtf(g.x * g.x + g.y * g.y);                       // losses.py:11
    const float m = (valid[i] != 0 && mag < max_flow) ? 1.0f : 0.0f;
    auto sgn = [](float v) { return v > 0.f ? 1.0f : (v < 0.f ? -1.0f : 0.0f); };
    for (int k = 0; k < n_pred; ++k) {
        const float2 p = preds[(int64_t)k * pred_stride + i];
        d_preds[(int64_t)k * pred_stride + i] = make_float2(lw.w[k] * m * sgn(p.x - g.x), lw.w[k] * m * sgn(p.y - g.y));
    }
}
}   // namespace

extern "C" int raft_sequence_loss_grad_f32(const float *flow_gt, const unsigned char *valid, const float *preds,
                                           int64_t pred_stride, int n_predictions, int64_t npix, double gamma, float max_flow,
                                           float upstream, float *d_preds, void *stream) {
    RAFT_REQUIRE_PTR(flow_gt);
    RAFT_REQUIRE_PTR(valid);
    RAFT_REQUIRE_PTR(preds);
    RAFT_REQUIRE_PTR(d_preds);
    RAFT_REQUIRE(npix > 0 && n_predictions > 0 && pred_stride >= npix * 2, RAFT_E_SHAPE);
    RAFT_REQUIRE(n_predictions <= MAX_PRED && (pred_stride & 1) == 0, RAFT_E_UNSUPPORTED);
    RAFT_REQUIRE(((uintptr_t)flow_gt & 7) == 0 && ((uintptr_t)preds & 7) == 0 && ((uintptr_t)d_preds & 7) == 0, RAFT_E_ALIGN);
    LossWeights lw = {};
    for (int i = 0; i < n_predictions; ++i) {
        double w = 1.0;
        for (int k = 0; k < n_predictions - i - 1; ++k) w *= gamma;
        lw.w[i] = (float)(w * (double)upstream / (2.0 * (double)npix));
    }
    sequence_loss_grad_kernel<<<raft_ceil_div(npix, 256), 256, 0, (hipStream_t)stream>>>(
        (const float2 *)flow_gt, valid, (const float2 *)preds, pred_stride / 2, n_predictions, npix, max_flow, lw,
        (float2 *)d_preds);
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// corr_lookup backward.  Forward (corr.py:116-152): for level l, tap (a, b):
//   gx = clamp(x / 2^l + (a - r), 0, w_l - 1), gy likewise with b;   out = wy0 wx0 F[y0][x0] + wy0 wx1 F[y0][x1] + ...
// with wx0 = ceil(gx) - gx, wx1 = gx - floor(gx).  Under tf.GradientTape floor / ceil have zero gradient and
// clip_by_value passes the gradient inside [min, max] (ends included), so
//   d out / d gx = [0 <= x/2^l + a - r <= w_l - 1] * (wy0 (F[y0][x1] - F[y0][x0]) + wy1 (F[y1][x1] - F[y1][x0])),
//   d out / d F[yc][xc] = wyc * wxc,       d gx / d x = 2^-l.
// One wavefront per query, levels in turn: the (2r+2)^2 footprint F and the level's (2r+1)^2 upstream gradients are
// parked in LDS, the per-axis tap tables (index relative to the footprint origin, weights, in-range flag) are built once,
// then  T[fx][b] = sum_a Wx[fx][a] g[a][b],  dF[fy][fx] = sum_b Wy[fy][b] T[fx][b]  (Wx[fx][a] = the weight tap a puts on
// footprint column fx) and the coordinate gradient is reduced over the wave in a fixed order.
// d_pyr is ACCUMULATED (+=): a map is touched by its own query's wavefront only, and the loop's iterations are ordered
// by the stream, so plain read-modify-write is race-free and deterministic.
// ------------------------------------------------------------------------------------------------
namespace {
struct LookupBwdArgs {
    const float *pyr;
    const float *coords;
    const float *d_out;
    float *d_coords;
    float *d_pyr;     // may be NULL
    PyramidGeom g;
    int64_t nq;
    int ld_out;
};

template <int R>
__global__ void __launch_bounds__(256) corr_lookup_backward_kernel(LookupBwdArgs p) {
    constexpr int D = 2 * R + 1, FW = 2 * R + 2, FP = FW * FW, WPB = 4;   // 4 queries (wavefronts) per workgroup
    __shared__ float sF[WPB][FP], sG[WPB][D * D], sT[WPB][FW * D], sW[WPB][2][D][4];
    __shared__ int sI[WPB][2][D][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * WPB + wv;
    if (q >= p.nq) return;                                               // wave-uniform; no workgroup barrier below
    const float2 cq = *(const float2 *)(p.coords + 2 * q);
    float acc_x = 0.f, acc_y = 0.f;
    for (int l = 0; l < p.g.levels; ++l) {
        const float sc = __int_as_float((127 - l) << 23);               // 2^-l
        const int w = p.g.lw[l], h = p.g.lh[l], tx = p.g.tx[l];
        const int ox = axis_tap(cq.x * sc, -R, w).i0, oy = axis_tap(cq.y * sc, -R, h).i0;
        const int64_t mbase = p.g.off[l] + q * (int64_t)p.g.map[l];
        // footprint (clamped reads like the forward) and the level's upstream gradients
        for (int s = lane; s < FP; s += 64) {
            const int y = min(oy + s / FW, h - 1), x = min(ox + s % FW, w - 1);
            sF[wv][s] = p.pyr[mbase + raft_tiled_index(y, x, tx)];
        }
        for (int s = lane; s < D * D; s += 64) sG[wv][s] = p.d_out[q * (int64_t)p.ld_out + l * D * D + s];
        // per-axis tap tables: lanes 0..D-1 the x taps, lanes 32..32+D-1 the y taps
        if ((lane & 31) < D) {
            const int axis = lane >> 5, d = lane & 31;
            const float c = (axis ? cq.y : cq.x) * sc;
            const int size = axis ? h : w, o = axis ? oy : ox;
            const AxisTap t = axis_tap(c, d - R, size);
            const float graw = c + (float)(d - R);
            const float in = (graw >= 0.f && graw <= (float)(size - 1)) ? 1.0f : 0.0f;   // clip_by_value gradient
            sI[wv][axis][d][0] = t.i0 - o;
            sI[wv][axis][d][1] = t.i1 - o;
            sW[wv][axis][d][0] = t.w0;
            sW[wv][axis][d][1] = t.w1;
            sW[wv][axis][d][2] = in;
            sW[wv][axis][d][3] = 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0): this wave's LDS writes are visible to it
        // coordinate gradient: every tap (a, b) -- a offsets x, b offsets y (corr.py:133-143), channel a * D + b
        for (int s = lane; s < D * D; s += 64) {
            const int a = s / D, b = s - a * D;
            const int x0 = sI[wv][0][a][0], x1 = sI[wv][0][a][1], y0 = sI[wv][1][b][0], y1 = sI[wv][1][b][1];
            const float wx0 = sW[wv][0][a][0], wx1 = sW[wv][0][a][1], inx = sW[wv][0][a][2];
            const float wy0 = sW[wv][1][b][0], wy1 = sW[wv][1][b][1], iny = sW[wv][1][b][2];
            const float v00 = sF[wv][y0 * FW + x0], v01 = sF[wv][y0 * FW + x1], v10 = sF[wv][y1 * FW + x0], v11 = sF[wv][y1 * FW + x1];
            const float g = sG[wv][s];
            acc_x += g * sc * inx * (wy0 * (v01 - v00) + wy1 * (v11 - v10));
            acc_y += g * sc * iny * (wx0 * (v10 - v00) + wx1 * (v11 - v01));
        }
        if (p.d_pyr) {
            // T[fx][b] = sum_a Wx[fx][a] * g[a][b]
            for (int s = lane; s < FW * D; s += 64) {
                const int fx = s / D, b = s - fx * D;
                float t = 0.f;
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const float wgt = (sI[wv][0][a][0] == fx ? sW[wv][0][a][0] : 0.f) + (sI[wv][0][a][1] == fx ? sW[wv][0][a][1] : 0.f);
                    t = fmaf(wgt, sG[wv][a * D + b], t);
                }
                sT[wv][s] = t;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            // dF[fy][fx] = sum_b Wy[fy][b] * T[fx][b]; positions past the border are clamped duplicates of the border
            // pixel in the forward's footprint and are never referenced by a tap: only in-range positions are written
            for (int s = lane; s < FP; s += 64) {
                const int fy = s / FW, fx = s - fy * FW;
                float dv = 0.f;
#pragma unroll
                for (int b = 0; b < D; ++b) {
                    const float wgt = (sI[wv][1][b][0] == fy ? sW[wv][1][b][0] : 0.f) + (sI[wv][1][b][1] == fy ? sW[wv][1][b][1] : 0.f);
                    dv = fmaf(wgt, sT[wv][fx * D + b], dv);
                }
                const int y = oy + fy, x = ox + fx;
                if (y < h && x < w) p.d_pyr[mbase + raft_tiled_index(y, x, tx)] += dv;
            }
        }
        __builtin_amdgcn_wave_barrier();                                 // the next level overwrites the LDS tables
    }
    // fixed-order butterfly: deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        acc_x += __shfl_xor(acc_x, o, 64);
        acc_y += __shfl_xor(acc_y, o, 64);
    }
    if (lane == 0) *(float2 *)(p.d_coords + 2 * q) = make_float2(acc_x, acc_y);
}
}   // namespace

extern "C" int raft_corr_lookup_backward_f32(const float *pyr, const int64_t *level_offsets, const float *coords,
                                             const float *d_out, int ld_out, int B, int h, int w, int levels, int radius,
                                             float *d_coords, float *d_pyr, void *stream) {
    RAFT_REQUIRE_PTR(pyr);
    RAFT_REQUIRE_PTR(level_offsets);
    RAFT_REQUIRE_PTR(coords);
    RAFT_REQUIRE_PTR(d_out);
    RAFT_REQUIRE_PTR(d_coords);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    LookupBwdArgs a;
    RAFT_TRY(raft_make_geom(h, w, levels, level_offsets, &a.g));
    const int d = 2 * radius + 1;
    RAFT_REQUIRE(ld_out >= levels * d * d, RAFT_E_SHAPE);
    a.pyr = pyr;
    a.coords = coords;
    a.d_out = d_out;
    a.d_coords = d_coords;
    a.d_pyr = d_pyr;
    a.nq = (int64_t)B * h * w;
    a.ld_out = ld_out;
    const int grid = raft_ceil_div(a.nq, 4);
    if (radius == 4)
        corr_lookup_backward_kernel<4><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    else if (radius == 3)
        corr_lookup_backward_kernel<3><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    else
        return RAFT_E_UNSUPPORTED;
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// relu backward
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) relu_backward_kernel(const float *__restrict__ y, const float *__restrict__ dy,
                                                            float *__restrict__ dx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

extern "C" int raft_relu_backward_f32(const float *y, const float *dy, float *dx, int64_t n, void *stream) {
    RAFT_REQUIRE_PTR(y);
    RAFT_REQUIRE_PTR(dy);
    RAFT_REQUIRE_PTR(dx);
    RAFT_REQUIRE(n > 0, RAFT_E_SHAPE);
    relu_backward_kernel<<<raft_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(y, dy, dx, n);
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Conv2D weight gradient (stride 1, 'same'), fp32 MFMA 16x16x4:
//   dK[ky][kx][ci][co] = sum over pixels  x[b, y + ky - pt, x + kx - pl, ci] * dy[b, y, x, co]
// a GEMM whose reduction dimension is the PIXELS.  A workgroup owns 64 input channels x 64 output channels x all kh*kw
// taps and a slice of the 4 x 16 pixel tiles; wave w takes input channels 16 w .. 16 w + 15: kh*kw x 4 accumulator tiles.
// Per pixel tile the x halo ((4 + kh - 1) x (16 + kw - 1) pixels x 64 channels) and the dy tile (64 pixels x 64
// channels) are staged in LDS with an 80-float pixel stride (the four k-groups of an MFMA operand read four consecutive
// pixels: 16 banks apart, conflict-free); an MFMA k-step is four consecutive pixels of a tile row, its A operand the
// tap-shifted x values, its B operand dy -- dy fragments are read once per k-step and reused by every tap.
// Partial sums go to workspace[slice][tap][ci][co]; wgrad_reduce_kernel adds the slices in order (deterministic) and
// writes the Keras-layout kernel gradient.  The bias gradient (column sums of dy) takes the same two steps.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int WG_TH = 4, WG_TW = 16, WG_PS = 80, WG_MAXT = 9, WG_MAXSEG = 32;

// x / dy may be SEGMENTED: nseg tensors of (B, H, W, .) each (the iterations of the prediction loop share their weights, so
// one launch sums the pixel reduction over all of them: raft_conv2d_wgrad_multi_f32); tile t lives in segment t / tiles_seg
struct WgradArgs {
    const float *x[WG_MAXSEG], *dy[WG_MAXSEG];
    float *part;          // [S][T][cin][cout]
    float *bias_part;     // [S][cout] column sums of dy per slice (written by the workgroups of input-channel block 0), or NULL
    int ldx, ldy, cin, cout, B, H, W, kh, kw, S, tiles_y, tiles_x, nseg, tiles_seg;
};

template <int KH, int KW>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(WgradArgs p) {
    constexpr int T = KH * KW, HH = WG_TH + KH - 1, HW = WG_TW + KW - 1;
    constexpr int PT = (KH - 1) / 2, PL = (KW - 1) / 2;
    constexpr int NX = (HH * HW * 16 + 255) / 256, NY = WG_TH * WG_TW * 16 / 256;   // 16-byte items per thread and tile
    static_assert(T <= WG_MAXT, "accumulator budget");
    __shared__ __attribute__((aligned(16))) float sX[HH * HW * WG_PS];
    __shared__ __attribute__((aligned(16))) float sY[WG_TH * WG_TW * WG_PS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int cib = blockIdx.x, cob = blockIdx.y, sl = blockIdx.z;
    const int ci0 = cib * 64, co0 = cob * 64;
    const int ntiles = p.nseg * p.tiles_seg;
    const int t_lo = (int)((long)ntiles * sl / p.S), t_hi = (int)((long)ntiles * (sl + 1) / p.S);

    f32x4 acc[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the NEXT tile's halo / dy items are fetched into registers while the current tile is multiplied out of LDS
    f32x4 rx[NX], ry[NY];
    const int c4 = tid & 15;
    auto fetch = [&](int tile) {
        const int seg = tile / p.tiles_seg, tl = tile - seg * p.tiles_seg;
        const int txi = tl % p.tiles_x, tyi = (tl / p.tiles_x) % p.tiles_y, b = tl / (p.tiles_x * p.tiles_y);
        const int y0 = tyi * WG_TH, x0 = txi * WG_TW;
        const float *xs = p.x[seg], *ds = p.dy[seg];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int hp = (tid + 256 * i) >> 4;
            const int yy = y0 - PT + hp / HW, xx = x0 - PL + hp % HW;
            const int c = ci0 + c4 * 4;
            rx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (hp < HH * HW && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W && c < p.cin)
                rx[i] = *(const f32x4 *)(xs + ((int64_t)(b * p.H + yy) * p.W + xx) * p.ldx + c);
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int pp = (tid + 256 * i) >> 4;
            const int yy = y0 + pp / WG_TW, xx = x0 + pp % WG_TW;
            const int c = co0 + c4 * 4;
            ry[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (yy < p.H && xx < p.W && c < p.cout)
                ry[i] = *(const f32x4 *)(ds + ((int64_t)(b * p.H + yy) * p.W + xx) * p.ldy + c);
        }
    };
    // bias gradient = column sums of dy: the workgroups of input-channel block 0 add up the dy items they stage anyway (a
    // thread's items all belong to channel quad tid & 15: four vector adds per tile, joined once at the end), instead of a
    // second pass over dy
    const bool do_bias = p.bias_part != nullptr && cib == 0;     // workgroup-uniform
    f32x4 bacc = {0.f, 0.f, 0.f, 0.f};
    if (t_lo < t_hi) fetch(t_lo);
    for (int tile = t_lo; tile < t_hi; ++tile) {
        __syncthreads();                                   // previous tile's fragments have been read
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int hp = (tid + 256 * i) >> 4;
            if (hp < HH * HW) *(f32x4 *)(sX + hp * WG_PS + c4 * 4) = rx[i];
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) *(f32x4 *)(sY + ((tid + 256 * i) >> 4) * WG_PS + c4 * 4) = ry[i];
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < NY; ++i) bacc += ry[i];
        }
        __syncthreads();
        if (tile + 1 < t_hi) fetch(tile + 1);
        // k-steps: 4 consecutive pixels of a tile row; lane (r, g): pixel 4 * step + g
#pragma unroll 2
        for (int step = 0; step < WG_TH * WG_TW / 4; ++step) {
            const int pp = step * 4 + g, py = pp / WG_TW, px = pp % WG_TW;
            float bf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = sY[pp * WG_PS + j * 16 + r];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int ky = t / KW, kx = t % KW;
                const float av = sX[((py + ky) * HW + px + kx) * WG_PS + wv * 16 + r];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bf[j], acc[t][j], 0, 0, 0);
            }
        }
    }
    // D[row = ci][col = co]: lane (col = r, rows 4 g + e)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = ci0 + wv * 16 + 4 * g + e, co = co0 + j * 16 + r;
                if (ci < p.cin && co < p.cout)
                    p.part[(((int64_t)sl * T + t) * p.cin + ci) * p.cout + co] = acc[t][j][e];
            }
    if (do_bias) {
        __syncthreads();
        *(f32x4 *)(sY + tid * 4) = bacc;                  // [pixel group tid >> 4][channel quad tid & 15][4]
        __syncthreads();
        const int co = co0 + tid;
        if (tid < 64 && co < p.cout) {
            float t = 0.f;
#pragma unroll
            for (int gq = 0; gq < 16; ++gq) t += sY[(gq * 16 + (tid >> 2)) * 4 + (tid & 3)];
            p.bias_part[(int64_t)sl * p.cout + co] = t;
        }
    }
}

// out[i] = sum over the S slices of part[k][i], in a fixed order: thread = (element blockIdx.x * 64 + tid % 64, quarter tid / 64 of
// the slices), eight independent loads in flight per thread, the quarters joined through LDS (one thread per element walking
// all slices with dependent adds was latency-bound: 49 us per call with 512 slices)
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *__restrict__ part, int S, int64_t n, float *__restrict__ out) {
    __shared__ float sh[4][64];
    const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + el;
    const int per = (S + 3) / 4, k0 = sg * per, k1 = (k0 + per < S) ? k0 + per : S;
    float s = 0.f;
    if (i < n) {
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(k + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < k1; ++k) s += part[(int64_t)k * n + i];
    }
    sh[sg][el] = s;
    __syncthreads();
    if (sg == 0 && i < n) out[i] = (sh[0][el] + sh[1][el]) + (sh[2][el] + sh[3][el]);
}

// column sums of dy: partial[blk][co] over a pixel range, then the same ordered reduction.  Thread = (pixel lane tid / 64,
// channel tid % 64): 64-channel chunks, four pixels in flight per chunk, the four lanes joined through LDS in fixed order.
// Segmented like the kernel gradient: pixel m of the concatenation lives in segment m / Mseg.
struct BiasGradArgs {
    const float *dy[WG_MAXSEG];
    int64_t Mseg;
    int nseg, ldy, cout, nblk;
    float *part;
};
__global__ void __launch_bounds__(256) bias_grad_partial_kernel(BiasGradArgs p) {
    __shared__ float sh[4][64];
    const int blk = blockIdx.x, cl = threadIdx.x & 63, pr = threadIdx.x >> 6;
    const int64_t M = p.Mseg * p.nseg;
    const int64_t lo = M * blk / p.nblk, hi = M * (blk + 1) / p.nblk;
    for (int c0 = 0; c0 < p.cout; c0 += 64) {
        const int c = c0 + cl;
        float s = 0.f;
        if (c < p.cout)
            for (int seg = (int)(lo / p.Mseg); seg < p.nseg && (int64_t)seg * p.Mseg < hi; ++seg) {   // the block's pieces, in order
                const int64_t base = (int64_t)seg * p.Mseg;
                const int64_t a = (lo > base ? lo : base) - base, b = (hi < base + p.Mseg ? hi : base + p.Mseg) - base;
                const float *d = p.dy[seg];
                for (int64_t m = a + pr; m < b; m += 4) s += d[m * p.ldy + c];
            }
        sh[pr][cl] = s;
        __syncthreads();
        if (pr == 0 && c < p.cout) p.part[(int64_t)blk * p.cout + c] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
        __syncthreads();
    }
}

int wgrad_slices(int cin, int cout, int B, int H, int W) {
    const int ntiles = B * ((H + WG_TH - 1) / WG_TH) * ((W + WG_TW - 1) / WG_TW);
    const int blocks = ((cin + 63) / 64) * ((cout + 63) / 64);
    int S = (512 + blocks - 1) / blocks;                   // about two workgroups per CU: every slice costs a pass of the
    if (S > ntiles) S = ntiles;                            // ordered second-stage sum over the whole kernel gradient
    // (no cap below that: the 64 -> 64 layers of the encoders are ONE channel block, and 64 slices were 64 workgroups for 256 CUs
    // -- 1.4 ms per launch at the training crop; 512 slices of 147 KB cost the second stage 15 us)
    return S < 1 ? 1 : S;
}
constexpr int BIAS_BLOCKS = 512;   // >= the largest slice count: the per-slice column sums of dy live behind the kernel partials
}   // namespace

extern "C" int64_t raft_conv2d_wgrad_workspace_floats(int cin, int cout, int B, int H, int W, int kh, int kw) {
    if (cin <= 0 || cout <= 0 || B <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0) return 0;
    return (int64_t)wgrad_slices(cin, cout, B, H, W) * kh * kw * cin * cout + (int64_t)BIAS_BLOCKS * cout;
}

// nseg (x, dy) pairs of identical geometry: d_kernel / d_bias = the sums over all of them, in ONE pixel reduction
static int wgrad_launch(const float *const *xs, const float *const *dys, int nseg, int ldx, int cin, int ldy, int cout, int B,
                        int H, int W, int kh, int kw, float *d_kernel, float *d_bias, float *workspace, void *stream) {
    RAFT_REQUIRE_PTR(xs);
    RAFT_REQUIRE_PTR(dys);
    RAFT_REQUIRE_PTR(d_kernel);
    RAFT_REQUIRE_PTR(workspace);
    RAFT_REQUIRE(nseg >= 1 && nseg <= WG_MAXSEG, RAFT_E_SHAPE);
    RAFT_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0 && cout > 0 && ldx >= cin && ldy >= cout, RAFT_E_SHAPE);
    RAFT_REQUIRE(cin % 4 == 0 && cout % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, RAFT_E_UNSUPPORTED);
    hipStream_t s = (hipStream_t)stream;
    WgradArgs a = {};
    for (int i = 0; i < nseg; ++i) {
        RAFT_REQUIRE_PTR(xs[i]);
        RAFT_REQUIRE_PTR(dys[i]);
        RAFT_REQUIRE(raft_aligned16(xs[i]) && raft_aligned16(dys[i]), RAFT_E_ALIGN);
        a.x[i] = xs[i];
        a.dy[i] = dys[i];
    }
    a.part = workspace;
    const int64_t n = (int64_t)kh * kw * cin * cout;
    a.S = wgrad_slices(cin, cout, B * nseg, H, W);
    a.bias_part = d_bias ? workspace + (int64_t)a.S * n : nullptr;
    a.ldx = ldx; a.ldy = ldy; a.cin = cin; a.cout = cout; a.B = B; a.H = H; a.W = W; a.kh = kh; a.kw = kw;
    a.tiles_y = (H + WG_TH - 1) / WG_TH;
    a.tiles_x = (W + WG_TW - 1) / WG_TW;
    a.nseg = nseg;
    a.tiles_seg = B * a.tiles_y * a.tiles_x;
    const dim3 grid((cin + 63) / 64, (cout + 63) / 64, a.S);
    if (kh == 1 && kw == 1)
        conv_wgrad_kernel<1, 1><<<grid, 256, 0, s>>>(a);
    else if (kh == 3 && kw == 3)
        conv_wgrad_kernel<3, 3><<<grid, 256, 0, s>>>(a);
    else if (kh == 1 && kw == 5)
        conv_wgrad_kernel<1, 5><<<grid, 256, 0, s>>>(a);
    else if (kh == 5 && kw == 1)
        conv_wgrad_kernel<5, 1><<<grid, 256, 0, s>>>(a);
    else
        return RAFT_E_UNSUPPORTED;
    RAFT_TRY(raft_launch_status());
    wgrad_reduce_kernel<<<raft_ceil_div(n, 64), 256, 0, s>>>(workspace, a.S, n, d_kernel);
    RAFT_TRY(raft_launch_status());
    if (d_bias) {
        wgrad_reduce_kernel<<<raft_ceil_div(cout, 64), 256, 0, s>>>(a.bias_part, a.S, cout, d_bias);
        RAFT_TRY(raft_launch_status());
    }
    return RAFT_OK;
}

extern "C" int raft_conv2d_wgrad_f32(const float *x, int ldx, int cin, const float *dy, int ldy, int cout, int B, int H, int W,
                                     int kh, int kw, float *d_kernel, float *d_bias, float *workspace, void *stream) {
    return wgrad_launch(&x, &dy, 1, ldx, cin, ldy, cout, B, H, W, kh, kw, d_kernel, d_bias, workspace, stream);
}

// The same gradient summed over nseg (<= 32) pairs (xs[i], dys[i]) of identical geometry -- host arrays of device pointers --
// in one pixel reduction: the 12 iterations of a training step share the update block's weights, and one launch over all of
// them replaces 12 launches + 12 second-stage sums + 11 accumulations per layer.  Workspace:
// raft_conv2d_wgrad_workspace_floats(cin, cout, B * nseg, H, W, kh, kw).
extern "C" int raft_conv2d_wgrad_multi_f32(const float *const *xs, const float *const *dys, int nseg, int ldx, int cin, int ldy,
                                           int cout, int B, int H, int W, int kh, int kw, float *d_kernel, float *d_bias,
                                           float *workspace, void *stream) {
    return wgrad_launch(xs, dys, nseg, ldx, cin, ldy, cout, B, H, W, kh, kw, d_kernel, d_bias, workspace, stream);
}

// ------------------------------------------------------------------------------------------------
// Elementwise pieces of the update block in training mode (reference update.py:51-67: SepConvGRU): the inference kernels
// apply the gates in their convolution epilogues and keep nothing; the training forward runs the convolutions with a
// linear epilogue and these kernels, so that z, r, q are available to the backward.
//   gate_zr      z = sigmoid(a[:, :C]), r = sigmoid(a[:, C:2C]), rh = r * h
//   gate_q       q = tanh(a), h' = (1 - z) h + z q
//   gate_q_bwd   dh' -> dz_pre = dh' (q - h) z (1 - z),  dq_pre = dh' z (1 - q^2),  dh = dh' (1 - z)
//   gate_r_bwd   d(rh) -> dr_pre = d(rh) h r (1 - r),  dh += d(rh) r
//   axpby        out = alpha a + beta b   (gradient accumulation, the 0.25 of the mask head)
// ------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

__global__ void __launch_bounds__(256) gate_zr_kernel(const float *__restrict__ a, const float *__restrict__ h, int C, int64_t M,
                                                      float *__restrict__ z, float *__restrict__ r, float *__restrict__ rh) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * C) return;
    const int64_t m = i / C;
    const int c = (int)(i - m * C);
    const float zv = sigmoidf_(a[m * 2 * C + c]), rv = sigmoidf_(a[m * 2 * C + C + c]);
    z[i] = zv;
    r[i] = rv;
    rh[i] = rv * h[i];
}

__global__ void __launch_bounds__(256) gate_q_kernel(const float *__restrict__ a, const float *__restrict__ z, const float *__restrict__ h,
                                                     int64_t n, float *__restrict__ q, float *__restrict__ hn) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float qv = tanhf(a[i]), zv = z[i];
    q[i] = qv;
    hn[i] = (1.0f - zv) * h[i] + zv * qv;
}

__global__ void __launch_bounds__(256) gate_q_bwd_kernel(const float *__restrict__ dhn, const float *__restrict__ z, const float *__restrict__ q,
                                                         const float *__restrict__ h, int64_t n, float *__restrict__ dz_pre,
                                                         float *__restrict__ dq_pre, float *__restrict__ dh) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = dhn[i], zv = z[i], qv = q[i];
    dz_pre[i] = g * (qv - h[i]) * zv * (1.0f - zv);
    dq_pre[i] = g * zv * (1.0f - qv * qv);
    dh[i] = g * (1.0f - zv);
}

__global__ void __launch_bounds__(256) gate_r_bwd_kernel(const float *__restrict__ drh, const float *__restrict__ r, const float *__restrict__ h,
                                                         int64_t n, float *__restrict__ dr_pre, float *__restrict__ dh) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = drh[i], rv = r[i];
    dr_pre[i] = g * h[i] * rv * (1.0f - rv);
    dh[i] += g * rv;
}

__global__ void __launch_bounds__(256) axpby_kernel(float alpha, const float *__restrict__ a, float beta, const float *__restrict__ b,
                                                    float *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}
}   // namespace

extern "C" int raft_gru_gate_zr_f32(const float *a_zr, const float *h, int C, int64_t M, float *z, float *r, float *rh, void *stream) {
    RAFT_REQUIRE_PTR(a_zr); RAFT_REQUIRE_PTR(h); RAFT_REQUIRE_PTR(z); RAFT_REQUIRE_PTR(r); RAFT_REQUIRE_PTR(rh);
    RAFT_REQUIRE(C > 0 && M > 0, RAFT_E_SHAPE);
    gate_zr_kernel<<<raft_ceil_div(M * C, 256), 256, 0, (hipStream_t)stream>>>(a_zr, h, C, M, z, r, rh);
    return raft_launch_status();
}

extern "C" int raft_gru_gate_q_f32(const float *a_q, const float *z, const float *h, int64_t n, float *q, float *h_new, void *stream) {
    RAFT_REQUIRE_PTR(a_q); RAFT_REQUIRE_PTR(z); RAFT_REQUIRE_PTR(h); RAFT_REQUIRE_PTR(q); RAFT_REQUIRE_PTR(h_new);
    RAFT_REQUIRE(n > 0, RAFT_E_SHAPE);
    gate_q_kernel<<<raft_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(a_q, z, h, n, q, h_new);
    return raft_launch_status();
}

extern "C" int raft_gru_gate_q_backward_f32(const float *d_h_new, const float *z, const float *q, const float *h, int64_t n,
                                            float *dz_pre, float *dq_pre, float *dh, void *stream) {
    RAFT_REQUIRE_PTR(d_h_new); RAFT_REQUIRE_PTR(z); RAFT_REQUIRE_PTR(q); RAFT_REQUIRE_PTR(h);
    RAFT_REQUIRE_PTR(dz_pre); RAFT_REQUIRE_PTR(dq_pre); RAFT_REQUIRE_PTR(dh);
    RAFT_REQUIRE(n > 0, RAFT_E_SHAPE);
    gate_q_bwd_kernel<<<raft_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(d_h_new, z, q, h, n, dz_pre, dq_pre, dh);
    return raft_launch_status();
}

extern "C" int raft_gru_gate_r_backward_f32(const float *d_rh, const float *r, const float *h, int64_t n, float *dr_pre, float *dh,
                                            void *stream) {
    RAFT_REQUIRE_PTR(d_rh); RAFT_REQUIRE_PTR(r); RAFT_REQUIRE_PTR(h); RAFT_REQUIRE_PTR(dr_pre); RAFT_REQUIRE_PTR(dh);
    RAFT_REQUIRE(n > 0, RAFT_E_SHAPE);
    gate_r_bwd_kernel<<<raft_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(d_rh, r, h, n, dr_pre, dh);
    return raft_launch_status();
}

extern "C" int raft_axpby_f32(float alpha, const float *a, float beta, const float *b, float *out, int64_t n, void *stream) {
    RAFT_REQUIRE_PTR(a); RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(n > 0, RAFT_E_SHAPE);
    axpby_kernel<<<raft_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(alpha, a, beta, b, out, n);
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// convf1 (7x7, Cin = 2; reference update.py:93) backward.  0.4 % of the block's FLOPs: plain deterministic kernels.
//   dgrad  d_flow[p][c] = sum_{ky,kx,n} dy[p - (ky-3, kx-3)][n] * K[ky][kx][c][n]      one wavefront per pixel, lanes over n
//   wgrad  dK[k][n] = sum_p flow[p + tap(k)][c(k)] * dy[p][n]                            pixel slices -> ordered second-stage sum
// ------------------------------------------------------------------------------------------------
namespace {
template <int COUT>
__global__ void __launch_bounds__(256) conv7x7_c2_dgrad_kernel(const float *__restrict__ dy, int ldy, const float *__restrict__ wk,
                                                               int B, int H, int W, float *__restrict__ dflow) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * 4 + wv;
    if (p >= (int64_t)B * H * W) return;
    const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
    float a0 = 0.f, a1 = 0.f;
    for (int ky = 0; ky < 7; ++ky) {
        const int yy = y - (ky - 3);
        if ((unsigned)yy >= (unsigned)H) continue;
        for (int kx = 0; kx < 7; ++kx) {
            const int xx = x - (kx - 3);
            if ((unsigned)xx >= (unsigned)W) continue;
            const float *d = dy + (((int64_t)b * H + yy) * W + xx) * ldy;
            const float *k0 = wk + ((ky * 7 + kx) * 2) * COUT;
            for (int n = lane; n < COUT; n += 64) {
                const float g = d[n];
                a0 = fmaf(g, k0[n], a0);
                a1 = fmaf(g, k0[COUT + n], a1);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_xor(a0, o, 64);
        a1 += __shfl_xor(a1, o, 64);
    }
    if (lane == 0) ((float2 *)dflow)[p] = make_float2(a0, a1);
}

constexpr int C7_SLICES = 256;
template <int COUT>
__global__ void __launch_bounds__(256) conv7x7_c2_wgrad_partial_kernel(const float *__restrict__ flow, const float *__restrict__ dy, int ldy,
                                                                       int B, int H, int W, float *__restrict__ part) {
    // thread = (k-group, n): n = tid % COUT, k walks 98 / (256 / COUT) values; pixels of the slice in order
    constexpr int KG = 256 / COUT, KPT = (98 + KG - 1) / KG;
    const int n = threadIdx.x % COUT, kg = threadIdx.x / COUT;
    const int64_t M = (int64_t)B * H * W, lo = M * blockIdx.x / C7_SLICES, hi = M * (blockIdx.x + 1) / C7_SLICES;
    float acc[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) acc[i] = 0.f;
    for (int64_t p = lo; p < hi; ++p) {
        const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
        const float g = dy[p * ldy + n];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int k = kg + i * KG;
            if (k < 98) {
                const int ky = k / 14, rem = k - ky * 14, kx = rem >> 1, c = rem & 1;
                const int yy = y + ky - 3, xx = x + kx - 3;
                const float f = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? flow[(((int64_t)b * H + yy) * W + xx) * 2 + c] : 0.f;
                acc[i] = fmaf(f, g, acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int k = kg + i * KG;
        if (k < 98) part[((int64_t)blockIdx.x * 98 + k) * COUT + n] = acc[i];
    }
}
}   // namespace

extern "C" int64_t raft_conv7x7_c2_wgrad_workspace_floats(int cout) { return cout > 0 ? (int64_t)C7_SLICES * (98 + 1) * cout : 0; }

extern "C" int raft_conv7x7_c2_backward_f32(const float *flow, const float *dy, int ldy, const float *kernel, int cout, int B, int H,
                                            int W, float *d_flow, float *d_kernel, float *d_bias, float *workspace, void *stream) {
    RAFT_REQUIRE_PTR(flow); RAFT_REQUIRE_PTR(dy); RAFT_REQUIRE_PTR(kernel); RAFT_REQUIRE_PTR(d_flow);
    RAFT_REQUIRE_PTR(d_kernel); RAFT_REQUIRE_PTR(d_bias); RAFT_REQUIRE_PTR(workspace);
    RAFT_REQUIRE(B > 0 && H > 0 && W > 0 && ldy >= cout, RAFT_E_SHAPE);
    RAFT_REQUIRE(cout == 64 || cout == 128, RAFT_E_UNSUPPORTED);
    hipStream_t s = (hipStream_t)stream;
    const int64_t M = (int64_t)B * H * W;
    if (cout == 128) {
        conv7x7_c2_dgrad_kernel<128><<<raft_ceil_div(M, 4), 256, 0, s>>>(dy, ldy, kernel, B, H, W, d_flow);
        conv7x7_c2_wgrad_partial_kernel<128><<<C7_SLICES, 256, 0, s>>>(flow, dy, ldy, B, H, W, workspace);
    } else {
        conv7x7_c2_dgrad_kernel<64><<<raft_ceil_div(M, 4), 256, 0, s>>>(dy, ldy, kernel, B, H, W, d_flow);
        conv7x7_c2_wgrad_partial_kernel<64><<<C7_SLICES, 256, 0, s>>>(flow, dy, ldy, B, H, W, workspace);
    }
    RAFT_TRY(raft_launch_status());
    wgrad_reduce_kernel<<<raft_ceil_div(98 * cout, 64), 256, 0, s>>>(workspace, C7_SLICES, (int64_t)98 * cout, d_kernel);
    float *bp = workspace + (int64_t)C7_SLICES * 98 * cout;
    {
        BiasGradArgs ba = {};
        ba.dy[0] = dy;
        ba.Mseg = M; ba.nseg = 1; ba.ldy = ldy; ba.cout = cout; ba.nblk = C7_SLICES; ba.part = bp;
        bias_grad_partial_kernel<<<C7_SLICES, 256, 0, s>>>(ba);
    }
    wgrad_reduce_kernel<<<raft_ceil_div(cout, 64), 256, 0, s>>>(bp, C7_SLICES, cout, d_bias);
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Convex upsampling backward (reference model.py:39-66):  up[8y+i, 8x+j, c] = sum_k p_k * 8 * flow[y+ky-1, x+kx-1, c],
// p = softmax_k(mask[y, x, (i*8+j)*9 + k]), zero padding outside the map.  For the upstream gradient g = d_up[8y+i, 8x+j]:
//   s_k = 8 * <flow_nb(k), g>,   d_mask[.., k] = p_k * (s_k - sum_k' p_k' s_k'),
//   d_flow[nb(k)] += 8 * sum_{i,j} p_k * g.
// One wavefront per coarse pixel, lane = sub-pixel (i, j).  The scatter into the neighbours is made deterministic in two
// steps: the wave writes its nine 2-vectors to T[pixel][k] (reduced over the lanes in a fixed butterfly), a second kernel
// GATHERS d_flow[p] = sum_k T[p - offset(k)][k].
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) upsample_convex_bwd_kernel(const float *__restrict__ flow, const float *__restrict__ mask,
                                                                  const float *__restrict__ d_up, int B, int h, int w,
                                                                  float *__restrict__ d_mask, float *__restrict__ T) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * 4 + wv;
    if (p >= (int64_t)B * h * w) return;
    const int x = (int)(p % w), y = (int)((p / w) % h);
    const int64_t b = p / ((int64_t)w * h);
    const int i = lane >> 3, j = lane & 7;
    const float *m = mask + p * 576 + lane * 9;
    float mk[9], mx = -3.4e38f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        mk[k] = m[k];
        mx = fmaxf(mx, mk[k]);
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        mk[k] = __expf(mk[k] - mx);
        sum += mk[k];
    }
    const float inv = 1.0f / sum;
    const float2 g = *(const float2 *)(d_up + (((b * 8 * h + 8 * y + i) * (int64_t)(8 * w)) + 8 * x + j) * 2);
    float s[9], sbar = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        mk[k] *= inv;
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        float2 f = make_float2(0.f, 0.f);
        if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) f = *(const float2 *)(flow + ((b * h + yy) * (int64_t)w + xx) * 2);
        s[k] = 8.0f * (f.x * g.x + f.y * g.y);
        sbar = fmaf(mk[k], s[k], sbar);
    }
    float *dm = d_mask + p * 576 + lane * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        dm[k] = mk[k] * (s[k] - sbar);
        float tx = 8.0f * mk[k] * g.x, ty = 8.0f * mk[k] * g.y;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            tx += __shfl_xor(tx, o, 64);
            ty += __shfl_xor(ty, o, 64);
        }
        if (lane == 0) *(float2 *)(T + (p * 9 + k) * 2) = make_float2(tx, ty);
    }
}

__global__ void __launch_bounds__(256) upsample_convex_bwd_gather_kernel(const float *__restrict__ T, int B, int h, int w,
                                                                         float *__restrict__ d_flow) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (int64_t)B * h * w) return;
    const int x = (int)(p % w), y = (int)((p / w) % h);
    const int64_t b = p / ((int64_t)w * h);
    float ax = 0.f, ay = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {   // pixel q = p - offset(k) used p as its neighbour k
        const int yy = y - (k / 3 - 1), xx = x - (k % 3 - 1);
        if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
            const float2 t = *(const float2 *)(T + ((((b * h + yy) * (int64_t)w + xx)) * 9 + k) * 2);
            ax += t.x;
            ay += t.y;
        }
    }
    *(float2 *)(d_flow + p * 2) = make_float2(ax, ay);
}
}   // namespace

extern "C" int64_t raft_upsample_convex_backward_workspace_floats(int B, int h, int w) {
    return (B > 0 && h > 0 && w > 0) ? (int64_t)B * h * w * 18 : 0;
}

extern "C" int raft_upsample_convex_backward_f32(const float *flow, const float *mask, const float *d_up, int B, int h, int w,
                                                 float *d_flow, float *d_mask, float *workspace, void *stream) {
    RAFT_REQUIRE_PTR(flow); RAFT_REQUIRE_PTR(mask); RAFT_REQUIRE_PTR(d_up);
    RAFT_REQUIRE_PTR(d_flow); RAFT_REQUIRE_PTR(d_mask); RAFT_REQUIRE_PTR(workspace);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    hipStream_t s = (hipStream_t)stream;
    const int64_t M = (int64_t)B * h * w;
    upsample_convex_bwd_kernel<<<raft_ceil_div(M, 4), 256, 0, s>>>(flow, mask, d_up, B, h, w, d_mask, workspace);
    RAFT_TRY(raft_launch_status());
    upsample_convex_bwd_gather_kernel<<<raft_ceil_div(M, 256), 256, 0, s>>>(workspace, B, h, w, d_flow);
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Optimizer side of train_step (reference model.py:131-136): tf.clip_by_global_norm + tfa.optimizers.AdamW.
//   raft_sumsq_f32       sum of squares of a tensor, float64, deterministic (partials per workgroup + ordered final sum);
//                        accumulate = 1 adds to *out (the global norm runs over all gradient tensors)
//   raft_adamw_step_f32  tensorflow-addons 0.11.1 DecoupledWeightDecayExtension + Keras Adam (TF 2.3), one tensor:
//                          var -= weight_decay * var                  (decoupled, NOT scaled by the learning rate)
//                          g = grad * grad_scale                      (grad_scale = clip_norm / max(global_norm, clip_norm))
//                          m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2
//                          var -= lr_t * m / (sqrt(v) + eps),  lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) computed by the caller
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int SUMSQ_WGS = 512;
__global__ void __launch_bounds__(256) sumsq_partial_kernel(const float *__restrict__ x, int64_t n, double *__restrict__ part) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = (double)x[i];
        acc += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void sumsq_final_kernel(const double *__restrict__ part, int nparts, int accumulate, double *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = accumulate ? out[0] : 0.0;
        for (int k = 0; k < nparts; ++k) s += part[k];
        out[0] = s;
    }
}
__global__ void __launch_bounds__(256) adamw_kernel(float *__restrict__ var, const float *__restrict__ grad, float *__restrict__ m,
                                                    float *__restrict__ v, int64_t n, float lr_t, float b1, float b2, float eps,
                                                    float wd, const double *__restrict__ gnorm_sq, float clip_norm) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float scale = 1.0f;
    if (gnorm_sq) {   // tf.clip_by_global_norm: g * clip_norm / max(global_norm, clip_norm)
        const float gn = (float)sqrt(gnorm_sq[0]);
        scale = clip_norm / fmaxf(gn, clip_norm);
    }
    const float g = grad[i] * scale;
    float w = var[i];
    w -= wd * w;
    const float mi = b1 * m[i] + (1.0f - b1) * g;
    const float vi = b2 * v[i] + (1.0f - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    var[i] = w - lr_t * mi / (sqrtf(vi) + eps);
}
}   // namespace

// ---- the same two steps over MANY tensors per launch (a model has 154 trainable tensors: 154 x 3 launches per step otherwise).
// Up to MT_MAX tensors travel in the kernel arguments; blockIdx.y = tensor, blockIdx.x = one of MT_BLOCKS strided workgroups.
constexpr int MT_MAX = 64, MT_BLOCKS = 16;
struct MultiTensorArgs {
    float *p[MT_MAX];             // var (adamw) / x (sumsq)
    const float *g[MT_MAX];       // grad
    float *m[MT_MAX], *v[MT_MAX];
    int64_t n[MT_MAX];
    int count;
};
__global__ void __launch_bounds__(256) sumsq_multi_partial_kernel(MultiTensorArgs a, double *__restrict__ part) {
    const int t = blockIdx.y;
    const float *x = a.p[t];
    const int64_t n = a.n[t];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)MT_BLOCKS * 256) {
        const double v = (double)x[i];
        acc += v * v;
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(int64_t)t * MT_BLOCKS + blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(256) adamw_multi_kernel(MultiTensorArgs a, float lr_t, float b1, float b2, float eps, float wd,
                                                          const double *__restrict__ gnorm_sq, float clip_norm) {
    const int t = blockIdx.y;
    float *var = a.p[t], *m = a.m[t], *v = a.v[t];
    const float *grad = a.g[t];
    const int64_t n = a.n[t];
    float scale = 1.0f;
    if (gnorm_sq) scale = clip_norm / fmaxf((float)sqrt(gnorm_sq[0]), clip_norm);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float g = grad[i] * scale;   // the arithmetic of adamw_kernel, statement for statement
        float w = var[i];
        w -= wd * w;
        const float mi = b1 * m[i] + (1.0f - b1) * g;
        const float vi = b2 * v[i] + (1.0f - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        var[i] = w - lr_t * mi / (sqrtf(vi) + eps);
    }
}

extern "C" int64_t raft_sumsq_workspace_doubles(void) { return SUMSQ_WGS; }

// Sum of squares over `count` tensors (host arrays of device pointers / element counts): float64, deterministic (per-tensor,
// per-workgroup partials in `workspace` -- count * 16 doubles -- then ONE ordered final sum).  *out is overwritten.
extern "C" int64_t raft_sumsq_multi_workspace_doubles(int count) { return count > 0 ? (int64_t)count * MT_BLOCKS : 0; }
extern "C" int raft_sumsq_multi_f32(const float *const *xs, const int64_t *ns, int count, double *out, double *workspace, void *stream) {
    RAFT_REQUIRE_PTR(xs); RAFT_REQUIRE_PTR(ns); RAFT_REQUIRE_PTR(out); RAFT_REQUIRE_PTR(workspace);
    RAFT_REQUIRE(count > 0, RAFT_E_SHAPE);
    hipStream_t s = (hipStream_t)stream;
    for (int lo = 0; lo < count; lo += MT_MAX) {
        MultiTensorArgs a = {};
        a.count = count - lo < MT_MAX ? count - lo : MT_MAX;
        for (int k = 0; k < a.count; ++k) {
            RAFT_REQUIRE_PTR(xs[lo + k]);
            RAFT_REQUIRE(ns[lo + k] > 0, RAFT_E_SHAPE);
            a.p[k] = const_cast<float *>(xs[lo + k]);
            a.n[k] = ns[lo + k];
        }
        sumsq_multi_partial_kernel<<<dim3(MT_BLOCKS, a.count), 256, 0, s>>>(a, workspace + (int64_t)lo * MT_BLOCKS);
        RAFT_TRY(raft_launch_status());
    }
    sumsq_final_kernel<<<1, 64, 0, s>>>(workspace, count * MT_BLOCKS, 0, out);
    return raft_launch_status();
}

// raft_adamw_step_f32 over `count` tensors per launch (chunks of 64): the same arithmetic per element.
extern "C" int raft_adamw_step_multi_f32(float *const *vars, const float *const *grads, float *const *ms, float *const *vs,
                                         const int64_t *ns, int count, float lr_t, float beta1, float beta2, float epsilon,
                                         float weight_decay, const double *global_norm_sq, float clip_norm, void *stream) {
    RAFT_REQUIRE_PTR(vars); RAFT_REQUIRE_PTR(grads); RAFT_REQUIRE_PTR(ms); RAFT_REQUIRE_PTR(vs); RAFT_REQUIRE_PTR(ns);
    RAFT_REQUIRE(count > 0, RAFT_E_SHAPE);
    hipStream_t s = (hipStream_t)stream;
    for (int lo = 0; lo < count; lo += MT_MAX) {
        MultiTensorArgs a = {};
        a.count = count - lo < MT_MAX ? count - lo : MT_MAX;
        int64_t nmax = 0;
        for (int k = 0; k < a.count; ++k) {
            RAFT_REQUIRE_PTR(vars[lo + k]); RAFT_REQUIRE_PTR(grads[lo + k]); RAFT_REQUIRE_PTR(ms[lo + k]); RAFT_REQUIRE_PTR(vs[lo + k]);
            RAFT_REQUIRE(ns[lo + k] > 0, RAFT_E_SHAPE);
            a.p[k] = vars[lo + k]; a.g[k] = grads[lo + k]; a.m[k] = ms[lo + k]; a.v[k] = vs[lo + k]; a.n[k] = ns[lo + k];
            if (ns[lo + k] > nmax) nmax = ns[lo + k];
        }
        int gx = (int)raft_ceil_div(nmax, 256 * 8);   // ~8 elements per thread of the largest tensor; small tensors finish early
        if (gx < 1) gx = 1;
        if (gx > 256) gx = 256;
        adamw_multi_kernel<<<dim3(gx, a.count), 256, 0, s>>>(a, lr_t, beta1, beta2, epsilon, weight_decay, global_norm_sq, clip_norm);
        RAFT_TRY(raft_launch_status());
    }
    return RAFT_OK;
}

extern "C" int raft_sumsq_f32(const float *x, int64_t n, int accumulate, double *out, double *workspace, void *stream) {
    RAFT_REQUIRE_PTR(x); RAFT_REQUIRE_PTR(out); RAFT_REQUIRE_PTR(workspace);
    RAFT_REQUIRE(n > 0, RAFT_E_SHAPE);
    hipStream_t s = (hipStream_t)stream;
    const int64_t g = raft_ceil_div(n, 256);
    const int grid = (int)(g < SUMSQ_WGS ? g : SUMSQ_WGS);
    sumsq_partial_kernel<<<grid, 256, 0, s>>>(x, n, workspace);
    RAFT_TRY(raft_launch_status());
    sumsq_final_kernel<<<1, 64, 0, s>>>(workspace, grid, accumulate, out);
    return raft_launch_status();
}

extern "C" int raft_adamw_step_f32(float *var, const float *grad, float *m, float *v, int64_t n, float lr_t, float beta1, float beta2,
                                   float epsilon, float weight_decay, const double *global_norm_sq, float clip_norm, void *stream) {
    RAFT_REQUIRE_PTR(var); RAFT_REQUIRE_PTR(grad); RAFT_REQUIRE_PTR(m); RAFT_REQUIRE_PTR(v);
    RAFT_REQUIRE(n > 0, RAFT_E_SHAPE);
    adamw_kernel<<<raft_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(var, grad, m, v, n, lr_t, beta1, beta2, epsilon, weight_decay,
                                                                         global_norm_sq, clip_norm);
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Generic strided batched GEMM on fp32 MFMA 16x16x4 (training-side plumbing: the backward of the volume build needs
// NN and TN products of activations):  C[b][m][n] = alpha * sum_k A(b, m, k) * B(b, k, n) + beta * C[b][m][n],
//   A(b, m, k) = a[b * sab + m * sam + k * sak],  B(b, k, n) = bm[b * sbb + k * sbk + n * sbn],  C row-major with ldc.
// Workgroup = 64 x 64 output tile, wave w = rows 16 w .. 16 w + 15 x 4 column blocks; 16-deep K chunks staged in LDS
// through the generic strides (zero fill past the edges).  Correctness-first: ~10-20 TF, enough for a step that runs
// once per training iteration next to ~300 convolutions.
// ------------------------------------------------------------------------------------------------
namespace {
struct GemmArgs {
    const float *a, *b;
    float *c;
    int M, N, K;
    int64_t sab, sam, sak, sbb, sbk, sbn, scb;
    int ldc;
    float alpha, beta;
};

__global__ void __launch_bounds__(256) gemm_strided_kernel(GemmArgs p) {
    __shared__ float sA[64][17], sB[16][65];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const float *A = p.a + (int64_t)blockIdx.z * p.sab, *B = p.b + (int64_t)blockIdx.z * p.sbb;
    float *Cm = p.c + (int64_t)blockIdx.z * p.scb;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < p.K; k0 += 16) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            {   // A tile: 64 rows x 16 k; consecutive threads walk the faster-varying of (m, k)
                const bool kfast = p.sak <= p.sam;
                const int mm = kfast ? e / 16 : e % 64, kk = kfast ? e % 16 : e / 64;
                const int m = m0 + mm, k = k0 + kk;
                sA[mm][kk] = (m < p.M && k < p.K) ? A[m * p.sam + k * p.sak] : 0.f;
            }
            {   // B tile: 16 k x 64 cols
                const bool nfast = p.sbn <= p.sbk;
                const int kk = nfast ? e / 64 : e % 16, nn = nfast ? e % 64 : e / 16;
                const int k = k0 + kk, n = n0 + nn;
                sB[kk][nn] = (k < p.K && n < p.N) ? B[k * p.sbk + n * p.sbn] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float av = sA[wv * 16 + r][q * 4 + g];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sB[q * 4 + g][j * 16 + r], acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + wv * 16 + 4 * g + e, n = n0 + j * 16 + r;
            if (m < p.M && n < p.N) {
                float *dst = Cm + (int64_t)m * p.ldc + n;
                *dst = p.alpha * acc[j][e] + (p.beta != 0.f ? p.beta * *dst : 0.f);
            }
        }
}

int launch_gemm(const GemmArgs &a, int batch, hipStream_t s) {
    const dim3 grid(raft_ceil_div(a.N, 64), raft_ceil_div(a.M, 64), batch);
    gemm_strided_kernel<<<grid, 256, 0, s>>>(a);
    return raft_launch_status();
}

// dF[l-1](child) += 0.25 * dF[l](parent): the adjoint of fmap_pool_kernel (2x2 VALID average over tiled rows)
__global__ void __launch_bounds__(256) fmap_pool_backward_kernel(float *__restrict__ ws, int64_t tot_rows_per_b, int C, int64_t child_off,
                                                                 int ch, int cw, int ctx, int64_t parent_off, int ph, int pw, int ptx,
                                                                 int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int x = (int)(r % cw);
    r /= cw;
    const int y = (int)(r % ch);
    const int64_t b = r / ch;
    if ((y >> 1) >= ph || (x >> 1) >= pw) return;                       // dropped by the VALID pooling
    const float g = ws[(b * tot_rows_per_b + parent_off + raft_tiled_index(y >> 1, x >> 1, ptx)) * C + c];
    ws[(b * tot_rows_per_b + child_off + raft_tiled_index(y, x, ctx)) * C + c] += 0.25f * g;
}

__global__ void __launch_bounds__(256) fmap_untile_level0_kernel(const float *__restrict__ ws, int64_t tot_rows_per_b, int C, int h, int w,
                                                                 int tiles_x, float *__restrict__ out, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int x = (int)(r % w);
    r /= w;
    const int y = (int)(r % h);
    const int64_t b = r / h;
    out[i] = ws[(b * tot_rows_per_b + raft_tiled_index(y, x, tiles_x)) * C + c];
}

__global__ void __launch_bounds__(256) state_backward_kernel(const float *__restrict__ net0, const float *__restrict__ inp, const float *__restrict__ d_net0,
                                                             const float *__restrict__ d_inp, int hdim, int cdim, int64_t M,
                                                             float *__restrict__ d_cnet) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per = hdim + cdim;
    if (i >= M * per) return;
    const int64_t m = i / per;
    const int c = (int)(i - m * per);
    if (c < hdim) {
        const float t = net0[m * hdim + c];
        d_cnet[i] = d_net0[m * hdim + c] * (1.0f - t * t);
    } else {
        d_cnet[i] = inp[m * cdim + c - hdim] > 0.f ? d_inp[m * cdim + c - hdim] : 0.f;
    }
}
}   // namespace

extern "C" int raft_gemm_f32(const float *a, int64_t sab, int64_t sam, int64_t sak, const float *b, int64_t sbb, int64_t sbk, int64_t sbn,
                             float *c, int64_t scb, int ldc, int batch, int M, int N, int K, float alpha, float beta, void *stream) {
    RAFT_REQUIRE_PTR(a); RAFT_REQUIRE_PTR(b); RAFT_REQUIRE_PTR(c);
    RAFT_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0 && ldc >= N, RAFT_E_SHAPE);
    GemmArgs g = {a, b, c, M, N, K, sab, sam, sak, sbb, sbk, sbn, scb, ldc, alpha, beta};
    return launch_gemm(g, batch, (hipStream_t)stream);
}

// Backward of raft_corr_build_f32: level l of the volume is fmap1 . pooled_l(fmap2)^T / sqrt(C), so
//   d_fmap1 = sum_l dP_l . F2_l / sqrt(C)            (NN products over the level's tiled map columns)
//   dF2_l   = dP_l^T . fmap1 / sqrt(C)               (TN)
// followed by the adjoint of the 2x2 average pooling chain (level 3 -> 0) and the un-tiling of level 0.
extern "C" int raft_corr_build_backward_f32(const float *fmap1, const float *fmap2_pyr, const float *d_pyr, const int64_t *level_offsets,
                                            int B, int h, int w, int C, int levels, float *d_fmap1, float *d_fmap2, float *workspace,
                                            void *stream) {
    RAFT_REQUIRE_PTR(fmap1); RAFT_REQUIRE_PTR(fmap2_pyr); RAFT_REQUIRE_PTR(d_pyr); RAFT_REQUIRE_PTR(level_offsets);
    RAFT_REQUIRE_PTR(d_fmap1); RAFT_REQUIRE_PTR(d_fmap2); RAFT_REQUIRE_PTR(workspace);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0, RAFT_E_SHAPE);
    PyramidGeom g;
    RAFT_TRY(raft_make_geom(h, w, levels, level_offsets, &g));
    hipStream_t s = (hipStream_t)stream;
    const int N = h * w;
    int64_t col_off[RAFT_MAX_LEVELS + 1], T = 0;
    for (int l = 0; l < levels; ++l) {
        col_off[l] = T;
        T += g.map[l];
    }
    const float rs = 1.0f / sqrtf((float)C);
    for (int l = 0; l < levels; ++l) {
        const int K = g.map[l];
        GemmArgs a1 = {d_pyr + g.off[l], fmap2_pyr + col_off[l] * C, d_fmap1, N, C, K,
                       (int64_t)N * K, K, 1, T * C, C, 1, (int64_t)N * C, C, rs, l ? 1.0f : 0.0f};
        RAFT_TRY(launch_gemm(a1, B, s));
        GemmArgs a2 = {d_pyr + g.off[l], fmap1, workspace + col_off[l] * C, K, C, N,
                       (int64_t)N * K, 1, K, (int64_t)N * C, C, 1, T * C, C, rs, 0.0f};
        RAFT_TRY(launch_gemm(a2, B, s));
    }
    for (int l = levels - 1; l >= 1; --l) {
        const int64_t total = (int64_t)B * g.lh[l - 1] * g.lw[l - 1] * C;
        fmap_pool_backward_kernel<<<raft_ceil_div(total, 256), 256, 0, s>>>(workspace, T, C, col_off[l - 1], g.lh[l - 1], g.lw[l - 1],
                                                                            g.tx[l - 1], col_off[l], g.lh[l], g.lw[l], g.tx[l], total);
        RAFT_TRY(raft_launch_status());
    }
    const int64_t total = (int64_t)B * N * C;
    fmap_untile_level0_kernel<<<raft_ceil_div(total, 256), 256, 0, s>>>(workspace, T, C, h, w, g.tx[0], d_fmap2, total);
    return raft_launch_status();
}

// model.py:84-86 backward: d cnet = [d_net0 * (1 - net0^2) | d_inp * (inp > 0)]
extern "C" int raft_prepare_state_backward_f32(const float *net0, const float *inp, const float *d_net0, const float *d_inp, int hdim,
                                               int cdim, int64_t M, float *d_cnet, void *stream) {
    RAFT_REQUIRE_PTR(net0); RAFT_REQUIRE_PTR(inp); RAFT_REQUIRE_PTR(d_net0); RAFT_REQUIRE_PTR(d_inp); RAFT_REQUIRE_PTR(d_cnet);
    RAFT_REQUIRE(hdim > 0 && cdim > 0 && M > 0, RAFT_E_SHAPE);
    state_backward_kernel<<<raft_ceil_div(M * (hdim + cdim), 256), 256, 0, (hipStream_t)stream>>>(net0, inp, d_net0, d_inp, hdim, cdim, M, d_cnet);
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Normalisation layers of the encoders in TRAINING form (reference extractor.py:6-16): tfa InstanceNormalization
// (per sample and channel over H x W) and Keras BatchNormalization with batch statistics (per channel over B x H x W),
// both eps = 1e-3 and biased variance.  x is viewed as (G groups, P pixels, C channels): G = B, P = H*W for instance
// norm; G = 1, P = B*H*W for batch norm.
//   stats     mean[g][c], rstd[g][c] = 1 / sqrt(var + eps)            (float64 partial sums over pixel slices, ordered)
//   apply     y = (x - mean) * rstd * gamma + beta   [+ relu]
//   backward  dx = gamma rstd (dy - mean_P(dy) - xhat mean_P(dy xhat)),  dgamma[c] = sum dy xhat,  dbeta[c] = sum dy
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int NORM_SLICES = 256;   // workgroups per group: batch norm has ONE group over B*H*W pixels

// part[(g * NORM_SLICES + s) * C + c] = {sum a, sum b} over the slice's pixels; mode 0: (x, x^2); mode 1: (dy, dy * xhat)
__global__ void __launch_bounds__(256) norm_partial_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ mean,
                                                           const float *__restrict__ rstd, int64_t P, int C, int mode,
                                                           double2 *__restrict__ part) {
    __shared__ double2 sh[4][64];
    const int g = blockIdx.x / NORM_SLICES, s = blockIdx.x % NORM_SLICES;
    const int cl = threadIdx.x & 63, pr = threadIdx.x >> 6;
    const int64_t lo = P * s / NORM_SLICES, hi = P * (s + 1) / NORM_SLICES;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + cl;
        double a = 0.0, b = 0.0;
        if (c < C) {
            const float mu = mode ? mean[(int64_t)g * C + c] : 0.f, rs = mode ? rstd[(int64_t)g * C + c] : 0.f;
            for (int64_t p = lo + pr; p < hi; p += 4) {
                const int64_t i = ((int64_t)g * P + p) * C + c;
                if (mode == 0) {
                    const double v = (double)x[i];
                    a += v;
                    b += v * v;
                } else {
                    const float d = dy[i];
                    a += (double)d;
                    b += (double)(d * ((x[i] - mu) * rs));
                }
            }
        }
        sh[pr][cl] = make_double2(a, b);
        __syncthreads();
        if (pr == 0 && c < C) {
            const double2 t0 = sh[0][cl], t1 = sh[1][cl], t2 = sh[2][cl], t3 = sh[3][cl];
            part[((int64_t)g * NORM_SLICES + s) * C + c] = make_double2((t0.x + t1.x) + (t2.x + t3.x), (t0.y + t1.y) + (t2.y + t3.y));
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) norm_stats_final_kernel(const double2 *__restrict__ part, int G, int C, double inv_count, float eps,
                                                               float *__restrict__ mean, float *__restrict__ rstd, float *__restrict__ var_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G * C) return;
    const int g = i / C, c = i - g * C;
    double a = 0.0, b = 0.0;
#pragma unroll 8
    for (int s = 0; s < NORM_SLICES; ++s) {
        const double2 t = part[((int64_t)g * NORM_SLICES + s) * C + c];
        a += t.x;
        b += t.y;
    }
    const double mu = a * inv_count;
    double var = b * inv_count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[i] = (float)mu;
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (var_out) var_out[i] = (float)var;
}

__global__ void __launch_bounds__(256) norm_apply_kernel(const float *__restrict__ x, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta, int64_t P, int C,
                                                         int relu, int64_t total, float *__restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int64_t g = i / ((int64_t)P * C);
    float v = (x[i] - mean[g * C + c]) * rstd[g * C + c] * gamma[c] + beta[c];
    y[i] = relu ? fmaxf(v, 0.f) : v;
}

// sums[(g * C + c)] = {sum dy, sum dy xhat} over the group (ordered over slices); dgamma / dbeta over the groups
__global__ void __launch_bounds__(256) norm_bwd_final_kernel(const double2 *__restrict__ part, int G, int C, double2 *__restrict__ sums,
                                                             float *__restrict__ dgamma, float *__restrict__ dbeta) {
    // thread = (channel blockIdx.x * 64 + tid % 64, quarter tid / 64 of the slices): independent loads in flight, the four
    // quarters joined through LDS in fixed order (one thread per channel walking G x NORM_SLICES records took 85 us per call)
    static_assert(NORM_SLICES % 4 == 0, "slice quarters");
    __shared__ double sh[2][4][64];
    const int cl = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double tg = 0.0, tb = 0.0;
    for (int g = 0; g < G; ++g) {
        double a = 0.0, b = 0.0;
        if (c < C) {
#pragma unroll 8
            for (int s = sg * (NORM_SLICES / 4); s < (sg + 1) * (NORM_SLICES / 4); ++s) {
                const double2 t = part[((int64_t)g * NORM_SLICES + s) * C + c];
                a += t.x;
                b += t.y;
            }
        }
        sh[0][sg][cl] = a;
        sh[1][sg][cl] = b;
        __syncthreads();
        if (sg == 0 && c < C) {
            a = (sh[0][0][cl] + sh[0][1][cl]) + (sh[0][2][cl] + sh[0][3][cl]);
            b = (sh[1][0][cl] + sh[1][1][cl]) + (sh[1][2][cl] + sh[1][3][cl]);
            sums[(int64_t)g * C + c] = make_double2(a, b);
            tb += a;
            tg += b;
        }
        __syncthreads();
    }
    if (sg == 0 && c < C) {
        dgamma[c] = (float)tg;
        dbeta[c] = (float)tb;
    }
}

__global__ void __launch_bounds__(256) norm_bwd_dx_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ mean,
                                                          const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                          const double2 *__restrict__ sums, int64_t P, int C, double inv_count, int64_t total,
                                                          float *__restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int64_t g = i / ((int64_t)P * C);
    const float rs = rstd[g * C + c], xh = (x[i] - mean[g * C + c]) * rs;
    const double2 t = sums[g * C + c];
    dx[i] = gamma[c] * rs * (dy[i] - (float)(t.x * inv_count) - xh * (float)(t.y * inv_count));
}
}   // namespace

extern "C" int64_t raft_norm_workspace_doubles(int G, int C) { return (G > 0 && C > 0) ? (int64_t)2 * (NORM_SLICES + 1) * G * C : 0; }

extern "C" int raft_norm_forward_f32(const float *x, int G, int64_t P, int C, const float *gamma, const float *beta, float eps, int relu,
                                     float *y, float *mean, float *rstd, float *var, double *workspace, void *stream) {
    RAFT_REQUIRE_PTR(x); RAFT_REQUIRE_PTR(gamma); RAFT_REQUIRE_PTR(beta); RAFT_REQUIRE_PTR(y);
    RAFT_REQUIRE_PTR(mean); RAFT_REQUIRE_PTR(rstd); RAFT_REQUIRE_PTR(workspace);
    RAFT_REQUIRE(G > 0 && P > 0 && C > 0, RAFT_E_SHAPE);
    hipStream_t s = (hipStream_t)stream;
    norm_partial_kernel<<<G * NORM_SLICES, 256, 0, s>>>(x, nullptr, nullptr, nullptr, P, C, 0, (double2 *)workspace);
    RAFT_TRY(raft_launch_status());
    norm_stats_final_kernel<<<raft_ceil_div((int64_t)G * C, 256), 256, 0, s>>>((const double2 *)workspace, G, C, 1.0 / (double)P, eps, mean, rstd, var);
    RAFT_TRY(raft_launch_status());
    const int64_t total = (int64_t)G * P * C;
    norm_apply_kernel<<<raft_ceil_div(total, 256), 256, 0, s>>>(x, mean, rstd, gamma, beta, P, C, relu, total, y);
    return raft_launch_status();
}

extern "C" int raft_norm_backward_f32(const float *x, const float *dy, const float *mean, const float *rstd, const float *gamma, int G,
                                      int64_t P, int C, float *dx, float *dgamma, float *dbeta, double *workspace, void *stream) {
    RAFT_REQUIRE_PTR(x); RAFT_REQUIRE_PTR(dy); RAFT_REQUIRE_PTR(mean); RAFT_REQUIRE_PTR(rstd); RAFT_REQUIRE_PTR(gamma);
    RAFT_REQUIRE_PTR(dx); RAFT_REQUIRE_PTR(dgamma); RAFT_REQUIRE_PTR(dbeta); RAFT_REQUIRE_PTR(workspace);
    RAFT_REQUIRE(G > 0 && P > 0 && C > 0, RAFT_E_SHAPE);
    hipStream_t s = (hipStream_t)stream;
    double2 *part = (double2 *)workspace, *sums = part + (int64_t)NORM_SLICES * G * C;
    norm_partial_kernel<<<G * NORM_SLICES, 256, 0, s>>>(x, dy, mean, rstd, P, C, 1, part);
    RAFT_TRY(raft_launch_status());
    norm_bwd_final_kernel<<<raft_ceil_div(C, 64), 256, 0, s>>>(part, G, C, sums, dgamma, dbeta);
    RAFT_TRY(raft_launch_status());
    const int64_t total = (int64_t)G * P * C;
    norm_bwd_dx_kernel<<<raft_ceil_div(total, 256), 256, 0, s>>>(x, dy, mean, rstd, gamma, sums, P, C, 1.0 / (double)P, total, dx);
    return raft_launch_status();
}

// out = [relu](alpha a + beta b): the residual join of a ResBlock (extractor.py:49)
extern "C" int raft_axpby_relu_f32(float alpha, const float *a, float beta, const float *b, float *out, int64_t n, void *stream);
namespace {
__global__ void __launch_bounds__(256) axpby_relu_kernel(float alpha, const float *__restrict__ a, float beta, const float *__restrict__ b,
                                                         float *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = fmaxf(alpha * a[i] + (b ? beta * b[i] : 0.f), 0.f);
}
}   // namespace
extern "C" int raft_axpby_relu_f32(float alpha, const float *a, float beta, const float *b, float *out, int64_t n, void *stream) {
    RAFT_REQUIRE_PTR(a); RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(n > 0, RAFT_E_SHAPE);
    axpby_relu_kernel<<<raft_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(alpha, a, beta, b, out, n);
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// upflow8 backward (reference corr.py:93-96: 8 * tf.image.resize(flow, 8x, 'bilinear'), half-pixel centres): the adjoint as a
// deterministic GATHER -- low-resolution pixel (y, x) collects, from every output pixel within reach, the weight that
// pixel's interpolation puts on it (src = (dst + 0.5) / 8 - 0.5, lo = max(floor(src), 0), hi = min(ceil(src), n - 1),
// lerp = src - floor(src); both taps of a clamped output land on the same index and simply add).
// ------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void resize_tap(int dst, int n, int *lo, int *hi, float *lerp) {
    const float src = ((float)dst + 0.5f) * 0.125f - 0.5f;
    const float fl = floorf(src);
    *lo = max((int)fl, 0);
    *hi = min((int)ceilf(src), n - 1);
    *lerp = src - fl;
}
__global__ void __launch_bounds__(256) upflow8_bwd_kernel(const float *__restrict__ d_up, int B, int h, int w, float *__restrict__ d_flow) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * h * w) return;
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const int64_t b = i / ((int64_t)w * h);
    const int H8 = 8 * h, W8 = 8 * w;
    float ax = 0.f, ay = 0.f;
    for (int Y = max(0, 8 * y - 8); Y <= min(H8 - 1, 8 * y + 15); ++Y) {
        int ylo, yhi;
        float yl;
        resize_tap(Y, h, &ylo, &yhi, &yl);
        const float wy = (ylo == y ? 1.0f - yl : 0.f) + (yhi == y ? yl : 0.f);
        if (wy == 0.f) continue;
        for (int X = max(0, 8 * x - 8); X <= min(W8 - 1, 8 * x + 15); ++X) {
            int xlo, xhi;
            float xl;
            resize_tap(X, w, &xlo, &xhi, &xl);
            const float wx = (xlo == x ? 1.0f - xl : 0.f) + (xhi == x ? xl : 0.f);
            if (wx == 0.f) continue;
            const float2 g = *(const float2 *)(d_up + ((b * H8 + Y) * (int64_t)W8 + X) * 2);
            ax = fmaf(wy * wx, g.x, ax);
            ay = fmaf(wy * wx, g.y, ay);
        }
    }
    *(float2 *)(d_flow + i * 2) = make_float2(8.0f * ax, 8.0f * ay);
}
}   // namespace

extern "C" int raft_upflow8_backward_f32(const float *d_up, int B, int h, int w, float *d_flow, void *stream) {
    RAFT_REQUIRE_PTR(d_up); RAFT_REQUIRE_PTR(d_flow);
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    upflow8_bwd_kernel<<<raft_ceil_div((int64_t)B * h * w, 256), 256, 0, (hipStream_t)stream>>>(d_up, B, h, w, d_flow);
    return raft_launch_status();
}


// ------------------------------------------------------------------------------------------------
// bf16 storage of the activation tape (BASELINE config 5: bf16 storage, fp32 arithmetic): the training forward keeps what the
// backward needs as bf16 (round to nearest even) and widens it again before use; every kernel still computes in fp32.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) f32_to_bf16_kernel(const float *__restrict__ x, unsigned short *__restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned u = __float_as_uint(x[i]);
    // NaN stays NaN (quiet bit), everything else rounds to nearest, ties to even
    y[i] = ((u & 0x7fffffffu) > 0x7f800000u) ? (unsigned short)((u >> 16) | 0x40u) : (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__global__ void __launch_bounds__(256) bf16_to_f32_kernel(const unsigned short *__restrict__ x, float *__restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = __uint_as_float((unsigned)x[i] << 16);
}
// counter-based uniform in [0, 1): two rounds of a 64-bit mix (splitmix64 finaliser) of (seed, index)
__device__ __forceinline__ float dropout_uniform(uint64_t seed, int64_t i) {
    uint64_t z = (uint64_t)i + seed * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}
__global__ void __launch_bounds__(256) dropout_kernel(const float *__restrict__ x, int64_t n, float rate, float scale, uint64_t seed,
                                                      float *__restrict__ y, unsigned char *__restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool keep = dropout_uniform(seed, i) >= rate;
    mask[i] = keep ? 1 : 0;
    y[i] = keep ? x[i] * scale : 0.f;
}
__global__ void __launch_bounds__(256) dropout_backward_kernel(const float *__restrict__ dy, const unsigned char *__restrict__ mask,
                                                               int64_t n, float scale, float *__restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dx[i] = mask[i] ? dy[i] * scale : 0.f;
}
}   // namespace

extern "C" int raft_f32_to_bf16(const float *x, void *y, int64_t n, void *stream) {
    RAFT_REQUIRE_PTR(x); RAFT_REQUIRE_PTR(y);
    RAFT_REQUIRE(n > 0, RAFT_E_SHAPE);
    f32_to_bf16_kernel<<<raft_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(x, (unsigned short *)y, n);
    return raft_launch_status();
}

extern "C" int raft_bf16_to_f32(const void *x, float *y, int64_t n, void *stream) {
    RAFT_REQUIRE_PTR(x); RAFT_REQUIRE_PTR(y);
    RAFT_REQUIRE(n > 0, RAFT_E_SHAPE);
    bf16_to_f32_kernel<<<raft_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>((const unsigned short *)x, y, n);
    return raft_launch_status();
}

// Keras Dropout in training mode (reference extractor.py:109-111, 127-128): y = x * keep / (1 - rate), keep ~ Bernoulli(1 - rate)
// from a counter-based generator of (seed, element index) -- reproducible, no state; mask[i] = keep (bytes, for the backward).
// TensorFlow's own random stream cannot be reproduced: the parity of this layer is distributional.
extern "C" int raft_dropout_f32(const float *x, int64_t n, float rate, uint64_t seed, float *y, unsigned char *mask, void *stream) {
    RAFT_REQUIRE_PTR(x); RAFT_REQUIRE_PTR(y); RAFT_REQUIRE_PTR(mask);
    RAFT_REQUIRE(n > 0 && rate >= 0.f && rate < 1.f, RAFT_E_SHAPE);
    dropout_kernel<<<raft_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(x, n, rate, 1.0f / (1.0f - rate), seed, y, mask);
    return raft_launch_status();
}

extern "C" int raft_dropout_backward_f32(const float *dy, const unsigned char *mask, int64_t n, float rate, float *dx, void *stream) {
    RAFT_REQUIRE_PTR(dy); RAFT_REQUIRE_PTR(mask); RAFT_REQUIRE_PTR(dx);
    RAFT_REQUIRE(n > 0 && rate >= 0.f && rate < 1.f, RAFT_E_SHAPE);
    dropout_backward_kernel<<<raft_ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(dy, mask, n, 1.0f / (1.0f - rate), dx);
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Training-time weight packing on the device, ONE launch per (kernel, use): the master weights live on the device and change
// every step, and repacking them with tensor-library ops was ~10 small kernels per layer and use (zero-fills, slice copies, casts,
// an einsum, a transposing copy: ~1200 launches = 7 ms of a 55 ms step, profiles/r11p_train_step_trace.txt).
//   kernel (kh, kw, cin, cout) Keras layout, fp32;  dgrad: use K'[ky][kx][co][ci] = K[kh-1-ky][kw-1-kx][ci][co] instead
//   mode 0: taps = kh * kw, U = K';  mode 1 (3x3): taps = 16, U[a][b] = sum_uv G[a][u] G[b][v] K'[u][v];  mode 2 (1x5 / 5x1):
//   taps = gt, U[t] = sum_j G[t][j] K'[j]  -- float64 accumulation, ONE rounding (packing.winograd_kernel / winograd1d_kernel)
//   wp[t][k / 4][n][k % 4] = U[t][k][n] for k < K, n < N, else 0  (K, N = cin, cout; swapped for dgrad); bias_out[n] = bias[n] or 0
// ------------------------------------------------------------------------------------------------
struct PackTrainArgs {
    const float *kernel, *bias;
    float *wp, *bias_out;
    int kh, kw, cin, cout, dgrad, mode, gt, kpad, npad;
    double g[40];   // mode 1: G (4 x 3); mode 2: G (gt x 5); row-major
};
__global__ void __launch_bounds__(256) pack_train_kernel(PackTrainArgs p) {
    const int taps = p.mode == 0 ? p.kh * p.kw : (p.mode == 1 ? 16 : p.gt);
    const int64_t total = (int64_t)taps * p.kpad * p.npad;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < p.npad) p.bias_out[i] = (p.bias != nullptr && !p.dgrad && i < p.cout) ? p.bias[i] : 0.f;
    if (i >= total) return;
    // i = ((t * kpad/4 + k4) * npad + n) * 4 + e : consecutive threads write consecutive floats of wp
    const int e = (int)(i & 3);
    const int64_t q = i >> 2;
    const int n = (int)(q % p.npad);
    const int64_t r = q / p.npad;
    const int k4 = (int)(r % (p.kpad >> 2)), t = (int)(r / (p.kpad >> 2));
    const int k = 4 * k4 + e;
    const int K = p.dgrad ? p.cout : p.cin, N = p.dgrad ? p.cin : p.cout;
    float v = 0.f;
    if (k < K && n < N) {
        auto src = [&](int ky, int kx) -> double {   // K'[ky][kx][k][n]
            if (p.dgrad) return (double)p.kernel[(((int64_t)(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)) * p.cin + n) * p.cout + k];
            return (double)p.kernel[(((int64_t)ky * p.kw + kx) * p.cin + k) * p.cout + n];
        };
        if (p.mode == 0) {
            v = (float)src(t / p.kw, t % p.kw);
        } else if (p.mode == 1) {
            const int a = t >> 2, b = t & 3;
            double acc = 0.0;
            for (int u = 0; u < 3; ++u) {
                double row = 0.0;
                for (int w = 0; w < 3; ++w) row += p.g[b * 3 + w] * src(u, w);
                acc += p.g[a * 3 + u] * row;
            }
            v = (float)acc;
        } else {
            double acc = 0.0;
            for (int j = 0; j < 5; ++j) acc += p.g[t * 5 + j] * (p.kh == 1 ? src(0, j) : src(j, 0));
            v = (float)acc;
        }
    }
    p.wp[i] = v;
}

extern "C" int raft_pack_train_conv_f32(const float *kernel, const float *bias, int kh, int kw, int cin, int cout, int dgrad,
                                        int mode, const double *g, int gt, int kpad, int npad, float *wp, float *bias_out,
                                        void *stream) {
    RAFT_REQUIRE_PTR(kernel);
    RAFT_REQUIRE_PTR(wp);
    RAFT_REQUIRE_PTR(bias_out);
    RAFT_REQUIRE(kh > 0 && kw > 0 && cin > 0 && cout > 0 && kpad > 0 && npad > 0, RAFT_E_SHAPE);
    RAFT_REQUIRE(kpad % 4 == 0 && kpad >= (dgrad ? cout : cin) && npad >= (dgrad ? cin : cout), RAFT_E_SHAPE);
    RAFT_REQUIRE(mode == 0 || (mode == 1 && kh == 3 && kw == 3) || (mode == 2 && ((kh == 1 && kw == 5) || (kh == 5 && kw == 1)) && gt > 0 && gt <= 8),
                 RAFT_E_UNSUPPORTED);
    RAFT_REQUIRE(!dgrad || (kh % 2 == 1 && kw % 2 == 1), RAFT_E_UNSUPPORTED);
    PackTrainArgs a = {};
    a.kernel = kernel; a.bias = bias; a.wp = wp; a.bias_out = bias_out;
    a.kh = kh; a.kw = kw; a.cin = cin; a.cout = cout; a.dgrad = dgrad; a.mode = mode; a.gt = gt; a.kpad = kpad; a.npad = npad;
    if (mode != 0) {
        RAFT_REQUIRE_PTR(g);
        const int ng = mode == 1 ? 12 : gt * 5;
        for (int i = 0; i < ng; ++i) a.g[i] = g[i];
    }
    const int taps = mode == 0 ? kh * kw : (mode == 1 ? 16 : gt);
    const int64_t total = (int64_t)taps * kpad * npad;
    RAFT_REQUIRE(total / 256 < 0x7fffffff, RAFT_E_UNSUPPORTED);
    pack_train_kernel<<<raft_ceil_div(total > npad ? total : npad, 256), 256, 0, (hipStream_t)stream>>>(a);
    return raft_launch_status();
}
