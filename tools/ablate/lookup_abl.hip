// Standalone ablation driver for the correlation lookup (diagnostics only, not part of the library).
// Built once per RAFT_LOOKUP_ABL value:  hipcc -DRAFT_LOOKUP_ABL=n ... -o ablate_lookup_n ; prints us per launch.
//   ablate_lookup_n [B] [reps] [staged 0|1] [thrash 0|1]
// RAFT_LOOKUP_ABL bits: 1 = no footprint gathers, 2 = no strip evaluation, 4 = no output stores, 8 = no taps,
//                       16 = empty kernel body (launch + grid ramp only)
// thrash = 1 streams a 512 MB buffer between launches (evicts L2 / Infinity Cache, as the rest of an update
// iteration does); the thrash-only loop is timed separately and subtracted.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../tf_raft_amd/csrc/corr.hip"

__global__ void thrash_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) dst[i] = src[i] * 1.0001f;
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4, reps = argc > 2 ? atoi(argv[2]) : 200;
    const int staged = argc > 3 ? atoi(argv[3]) : 1, thrash = argc > 4 ? atoi(argv[4]) : 0;
    const int h = 56, w = 64, levels = 4;
    int64_t off[5];
    int lh[4], lw[4];
    raft_corr_pyramid_layout(B, h, w, levels, off, lh, lw);
    const int64_t nq = (int64_t)B * h * w;
    float *pyr, *coords, *out;
    hipMalloc(&pyr, off[4] * 4);
    hipMalloc(&coords, nq * 2 * 4);
    hipMalloc(&out, nq * 352 * 4);
    {   // random volume (values do not matter), coords = grid + N(0, 6)
        std::vector<float> hp(1 << 24);
        srand(1);
        for (auto &v : hp) v = rand() / (float)RAND_MAX - 0.5f;
        for (int64_t o = 0; o < off[4]; o += hp.size())
            hipMemcpy(pyr + o, hp.data(), (size_t)std::min<int64_t>(hp.size(), off[4] - o) * 4, hipMemcpyHostToDevice);
        std::vector<float> hc(nq * 2);
        for (int64_t q = 0; q < nq; ++q) {
            float n1 = 0, n2 = 0;
            for (int k = 0; k < 12; ++k) { n1 += rand() / (float)RAND_MAX - 0.5f; n2 += rand() / (float)RAND_MAX - 0.5f; }
            hc[2 * q] = (float)(q % w) + 6.f * n1;
            hc[2 * q + 1] = (float)((q / w) % h) + 6.f * n2;
        }
        hipMemcpy(coords, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    }
    setenv("RAFT_LOOKUP_STAGED", staged ? "1" : "0", 1);
    const size_t tn = (size_t)(512u << 20) / 16;
    f32x4 *ta = nullptr, *tb = nullptr;
    if (thrash) { hipMalloc(&ta, tn * 16); hipMalloc(&tb, tn * 16); hipMemset(ta, 0, tn * 16); }
    auto go = [&]() { raft_corr_lookup_f32(pyr, off, coords, B, h, w, levels, 4, out, 352, nullptr); };
    auto th = [&]() { if (thrash) thrash_kernel<<<2048, 256>>>(ta, tb, tn); };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](bool with_lookup) {
        for (int i = 0; i < 3; ++i) { th(); if (with_lookup) go(); }
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) { th(); if (with_lookup) go(); }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        return ms * 1e3f / reps;
    };
    const float base = thrash ? timed(false) : 0.f;
    const float us = timed(true) - base;
    const double bytes = (double)nq * (4 * 100 * 4 + 8 + 324 * 4);
    printf("lookup abl=%d B=%d staged=%d thrash=%d: %.2f us per launch  (%.0f GB/s algorithmic)%s\n", RAFT_LOOKUP_ABL, B,
           staged, thrash, us, bytes / us / 1e3, thrash ? "  [thrash loop subtracted]" : "");
    return 0;
}
