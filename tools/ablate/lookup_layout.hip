// Layout study of the correlation pyramid (diagnostics only, not part of the library): the volume build and the stand-alone
// lookup compiled with another tile shape of the per-query maps (-DRAFT_TILE_H_LOG2=a -DRAFT_TILE_W_LOG2=b; the product uses
// 4 x 8 = one 128-byte line per tile).  Random feature maps -> raft_corr_build_f32 -> raft_corr_lookup_f32 at coords = grid +
// N(0, sigma); prints us per launch for both (HIP events, back-to-back launches) and a checksum of the lookup output, which must
// be the same for every tile shape (the layout is invisible in the result).
//   lookup_layout_<h>x<w> [B] [reps] [thrash 0|1] [sigma]
// thrash = 1 streams 512 MB between launches (what the rest of an iteration does to L2 / the Infinity Cache); that loop alone is
// timed and subtracted.  Under rocprofv3 (kernel trace / --pmc) the same binary gives kernel durations and FETCH_SIZE.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../tf_raft_amd/csrc/corr.hip"
#include "../../tf_raft_amd/csrc/host_util.hip"

__global__ void thrash_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) dst[i] = src[i] * 1.0001f;
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4, reps = argc > 2 ? atoi(argv[2]) : 100;
    const int thrash = argc > 3 ? atoi(argv[3]) : 0;
    const float sigma = argc > 4 ? (float)atof(argv[4]) : 6.f;
    const int h = 56, w = 64, levels = 4, C = 256;
    int64_t off[5];
    int lh[4], lw[4];
    if (raft_corr_pyramid_layout(B, h, w, levels, off, lh, lw)) return 1;
    const int64_t nq = (int64_t)B * h * w;
    float *pyr, *coords, *out, *f1, *f2, *ws;
    hipMalloc(&pyr, off[4] * 4);
    hipMalloc(&coords, nq * 2 * 4);
    hipMalloc(&out, nq * 352 * 4);
    hipMalloc(&f1, nq * C * 4);
    hipMalloc(&f2, nq * C * 4);
    hipMalloc(&ws, raft_corr_build_workspace_floats(B, h, w, C, levels) * 4);
    hipMemset(out, 0, nq * 352 * 4);
    {
        std::vector<float> hf(nq * C);
        srand(1);
        for (auto &v : hf) v = rand() / (float)RAND_MAX - 0.5f;
        hipMemcpy(f1, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
        for (auto &v : hf) v = rand() / (float)RAND_MAX - 0.5f;
        hipMemcpy(f2, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> hc(nq * 2);
        for (int64_t q = 0; q < nq; ++q) {
            float n1 = 0, n2 = 0;
            for (int k = 0; k < 12; ++k) { n1 += rand() / (float)RAND_MAX - 0.5f; n2 += rand() / (float)RAND_MAX - 0.5f; }
            hc[2 * q] = (float)(q % w) + sigma * n1;
            hc[2 * q + 1] = (float)((q / w) % h) + sigma * n2;
        }
        hipMemcpy(coords, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    }
    const size_t tn = (size_t)(512u << 20) / 16;
    f32x4 *ta = nullptr, *tb = nullptr;
    if (thrash) { hipMalloc(&ta, tn * 16); hipMalloc(&tb, tn * 16); hipMemset(ta, 0, tn * 16); }
    auto build = [&]() { return raft_corr_build_f32(f1, f2, B, h, w, C, levels, pyr, off, ws, nullptr); };
    auto look = [&]() { return raft_corr_lookup_f32(pyr, off, coords, B, h, w, levels, 4, out, 352, nullptr); };
    auto th = [&]() { if (thrash) thrash_kernel<<<2048, 256>>>(ta, tb, tn); };
    if (build() || look()) { printf("launch failed\n"); return 1; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](int what, int n) {   // 0 = thrash only, 1 = lookup, 2 = build
        for (int i = 0; i < 2; ++i) { th(); if (what == 1) look(); if (what == 2) build(); }
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < n; ++i) { th(); if (what == 1) look(); if (what == 2) build(); }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        return ms * 1e3f / n;
    };
    const float base = thrash ? timed(0, reps) : 0.f;
    const float us_l = timed(1, reps) - base;
    const float us_b = timed(2, reps / 4 + 1) - base;
    std::vector<float> ho(nq * 352);
    hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    double sum = 0, asum = 0;
    for (int64_t q = 0; q < nq; ++q)
        for (int c = 0; c < 324; ++c) { sum += ho[q * 352 + c]; asum += fabs((double)ho[q * 352 + c]); }
    const double bytes = (double)nq * (4 * 100 * 4 + 8 + 324 * 4);
    printf("tile %dx%d (%d B) B=%d thrash=%d sigma=%.1f: lookup %.2f us (%.0f GB/s algorithmic)  build %.1f us  pyramid %.1f MB  checksum %.6e %.6e\n",
           RAFT_TILE_H, RAFT_TILE_W, RAFT_TILE_FLOATS * 4, B, thrash, sigma, us_l, bytes / us_l / 1e3, us_b, off[4] * 4 / 1e6, sum, asum);
    return 0;
}
