// Evaluation-side reductions over the flow predictions (reference tf_raft/losses/losses.py), gfx950.
//
//   raft_flow_metrics_f32   end_point_error (losses.py:24-43; what test_step feeds its metrics with, model.py:146-158):
//                           valid' = valid & (|flow_gt| < max_flow); EPE = |pred - gt|_2 over valid' pixels;
//                           out = {mean EPE, rate EPE<1, rate EPE<3, rate EPE<5, number of valid' pixels}
//   raft_sequence_loss_f32  sequence_loss (losses.py:4-21): sum_i gamma^(n-i-1) * mean(valid' * |pred_i - gt|), the
//                           mean running over ALL B*H*W*2 elements; the n predictions are read in one pass
//
// Both are single passes over HBM: per pixel 8 B of ground truth, 1 B of mask and 8 B per prediction.  Deterministic:
// every thread accumulates its grid-stride pixels in float64, a workgroup reduces with shuffles + LDS into one partial
// record, and a second one-workgroup kernel adds the partial records in index order (no atomics).
// The per-pixel arithmetic keeps the reference's operation order (x*x + y*y, then sqrt; fp32, no contraction).
#include "common.h"

namespace {

constexpr int METRIC_WGS = 1024;      // partial records (>= 4 workgroups per CU)
constexpr int METRIC_VALS = 5;
constexpr int MAX_PRED = 64;

struct LossWeights {
    float w[MAX_PRED];
};

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// reduce NV per-thread doubles over the 256 threads of a workgroup into part[blockIdx.x][NV]
template <int NV>
__device__ inline void block_reduce_store(double (&v)[NV], double *__restrict__ part) {
    __shared__ double sh[4][NV];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) sh[w][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) part[(int64_t)blockIdx.x * NV + threadIdx.x] =
        ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

__device__ inline bool pixel_valid(float2 g, unsigned char v, float max_flow) {
#pragma clang fp contract(off)
    const float mag = sqrtf(g.x * g.x + g.y * g.y);      // losses.py:11 / 28
    return v != 0 && mag < max_flow;
}

__global__ void __launch_bounds__(256) flow_metrics_partial_kernel(const float2 *__restrict__ gt,
                                                                    const unsigned char *__restrict__ valid,
                                                                    const float2 *__restrict__ pred, int64_t npix,
                                                                    float max_flow, double *__restrict__ part) {
#pragma clang fp contract(off)
    double acc[METRIC_VALS] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        const float2 g = gt[i], p = pred[i];
        if (!pixel_valid(g, valid[i], max_flow)) continue;
        const float dx = p.x - g.x, dy = p.y - g.y;
        const float epe = sqrtf(dx * dx + dy * dy);       // losses.py:31
        acc[0] += (double)epe;
        acc[1] += epe < 1.0f ? 1.0 : 0.0;
        acc[2] += epe < 3.0f ? 1.0 : 0.0;
        acc[3] += epe < 5.0f ? 1.0 : 0.0;
        acc[4] += 1.0;
    }
    block_reduce_store<METRIC_VALS>(acc, part);
}

__global__ void __launch_bounds__(256) flow_metrics_final_kernel(const double *__restrict__ part, int nparts,
                                                                  float *__restrict__ out) {
    __shared__ double tot[METRIC_VALS];
    if (threadIdx.x < METRIC_VALS) {
        double s = 0.0;
        for (int k = 0; k < nparts; ++k) s += part[(int64_t)k * METRIC_VALS + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x < METRIC_VALS) {
        // the mean of an empty selection is NaN, as tf.reduce_mean of an empty tensor
        const double n = tot[4];
        out[threadIdx.x] = threadIdx.x == 4 ? (float)n : (float)(tot[threadIdx.x] / n);
    }
}

__global__ void __launch_bounds__(256) sequence_loss_partial_kernel(const float2 *__restrict__ gt,
                                                                     const unsigned char *__restrict__ valid,
                                                                     const float2 *__restrict__ preds, int64_t pred_stride,
                                                                     int n_pred, int64_t npix, float max_flow, LossWeights lw,
                                                                     double *__restrict__ part) {
#pragma clang fp contract(off)
    double acc[1] = {0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        const float2 g = gt[i];
        // losses.py:14-19 MULTIPLIES by the 0/1 mask (it does not select): a NaN / Inf prediction or ground truth at a
        // masked pixel makes the reference loss NaN (0 * NaN), and so it does here
        const float m = pixel_valid(g, valid[i], max_flow) ? 1.0f : 0.0f;
        double s = 0.0;
        for (int k = 0; k < n_pred; ++k) {
            const float2 p = preds[(int64_t)k * pred_stride + i];
            s += (double)lw.w[k] * ((double)(m * fabsf(p.x - g.x)) + (double)(m * fabsf(p.y - g.y)));   // losses.py:18-19
        }
        acc[0] += s;
    }
    block_reduce_store<1>(acc, part);
}

__global__ void __launch_bounds__(64) sequence_loss_final_kernel(const double *__restrict__ part, int nparts, double inv_count,
                                                                  float *__restrict__ out) {
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < nparts; ++k) s += part[k];
        out[0] = (float)(s * inv_count);
    }
}

int metric_grid(int64_t npix) {
    const int64_t g = raft_ceil_div(npix, 256);
    return (int)(g < METRIC_WGS ? (g < 1 ? 1 : g) : METRIC_WGS);
}

}   // namespace

extern "C" int64_t raft_metrics_workspace_doubles(void) { return (int64_t)METRIC_WGS * METRIC_VALS; }

extern "C" int raft_flow_metrics_f32(const float *flow_gt, const unsigned char *valid, const float *flow_pred, int64_t npix,
                                     float max_flow, float *out5, double *workspace, void *stream) {
    RAFT_REQUIRE_PTR(flow_gt);
    RAFT_REQUIRE_PTR(valid);
    RAFT_REQUIRE_PTR(flow_pred);
    RAFT_REQUIRE_PTR(out5);
    RAFT_REQUIRE_PTR(workspace);
    RAFT_REQUIRE(npix > 0, RAFT_E_SHAPE);
    RAFT_REQUIRE(((uintptr_t)flow_gt & 7) == 0 && ((uintptr_t)flow_pred & 7) == 0 && ((uintptr_t)workspace & 7) == 0, RAFT_E_ALIGN);
    hipStream_t s = (hipStream_t)stream;
    const int grid = metric_grid(npix);
    flow_metrics_partial_kernel<<<grid, 256, 0, s>>>((const float2 *)flow_gt, valid, (const float2 *)flow_pred, npix, max_flow,
                                                     workspace);
    RAFT_TRY(raft_launch_status());
    flow_metrics_final_kernel<<<1, 256, 0, s>>>(workspace, grid, out5);
    return raft_launch_status();
}

extern "C" int raft_sequence_loss_f32(const float *flow_gt, const unsigned char *valid, const float *preds, int64_t pred_stride,
                                      int n_predictions, int64_t npix, double gamma, float max_flow, float *loss_out,
                                      double *workspace, void *stream) {
    RAFT_REQUIRE_PTR(flow_gt);
    RAFT_REQUIRE_PTR(valid);
    RAFT_REQUIRE_PTR(preds);
    RAFT_REQUIRE_PTR(loss_out);
    RAFT_REQUIRE_PTR(workspace);
    RAFT_REQUIRE(npix > 0 && n_predictions > 0 && pred_stride >= npix * 2, RAFT_E_SHAPE);
    RAFT_REQUIRE(n_predictions <= MAX_PRED && (pred_stride & 1) == 0, RAFT_E_UNSUPPORTED);
    RAFT_REQUIRE(((uintptr_t)flow_gt & 7) == 0 && ((uintptr_t)preds & 7) == 0 && ((uintptr_t)workspace & 7) == 0, RAFT_E_ALIGN);
    hipStream_t s = (hipStream_t)stream;
    LossWeights lw = {};
    for (int i = 0; i < n_predictions; ++i) {   // losses.py:17: gamma ** (n - i - 1), a Python float rounded to fp32 by the product
        double w = 1.0;
        for (int k = 0; k < n_predictions - i - 1; ++k) w *= gamma;
        lw.w[i] = (float)w;
    }
    const int grid = metric_grid(npix);
    sequence_loss_partial_kernel<<<grid, 256, 0, s>>>((const float2 *)flow_gt, valid, (const float2 *)preds, pred_stride / 2,
                                                      n_predictions, npix, max_flow, lw, workspace);
    RAFT_TRY(raft_launch_status());
    sequence_loss_final_kernel<<<1, 64, 0, s>>>(workspace, grid, 1.0 / ((double)npix * 2.0), loss_out);
    return raft_launch_status();
}
