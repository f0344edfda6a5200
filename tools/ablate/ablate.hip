// Standalone ablation driver for the implicit-GEMM convolution (diagnostics only, not part of the
// library).  Built once per RAFT_ABL value:  hipcc -DRAFT_ABL=n ... -o ablate_n ; prints us per launch.
//   ablate <kh> <kw> <cin> <npad> <tile> [B] [reps]
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../tf_raft_amd/csrc/conv_mfma.h"

template <int KH, int KW>
static void launch(const ConvArgs &a, int tile, int grid, hipStream_t s) {
    switch (tile) {
        case 0: conv_mfma_kernel<KH, KW, 32, 128, 128, 2, 2, EPI_RELU><<<grid, 256, 0, s>>>(a); break;
        case 3: conv_mfma_kernel<KH, KW, 32, 64, 64, 2, 2, EPI_RELU><<<grid, 256, 0, s>>>(a); break;
        case 4: conv_mfma_kernel<KH, KW, 16, 112, 128, 1, 4, EPI_RELU><<<grid, 256, 0, s>>>(a); break;
        case 5: conv_mfma_kernel<KH, KW, 16, 112, 64, 1, 4, EPI_RELU><<<grid, 256, 0, s>>>(a); break;
    }
}

int main(int argc, char **argv) {
    const int kh = atoi(argv[1]), kw = atoi(argv[2]), cin = atoi(argv[3]), npad = atoi(argv[4]), tile = atoi(argv[5]);
    const int B = argc > 6 ? atoi(argv[6]) : 4, reps = argc > 7 ? atoi(argv[7]) : 20;
    const int H = 56, W = 64, M = B * H * W;
    const int bm[6] = {128, 64, 128, 64, 112, 112}, bn[6] = {128, 128, 64, 64, 128, 64};
    float *x, *w, *bias, *out;
    const size_t nx = (size_t)M * cin, nw = (size_t)kh * kw * cin * npad;
    hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&bias, npad * 4); hipMalloc(&out, (size_t)M * npad * 4);
    std::vector<float> hx(nx), hw(nw);
    srand(1);
    const bool zero = getenv("ABL_ZERO") != nullptr;
    for (auto &v : hx) v = zero ? 0.f : (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto &v : hw) v = zero ? 0.f : ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f;
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, npad * 4);
    ConvArgs a = {};
    a.a0 = x; a.lda0 = cin; a.c0 = cin; a.wp = w; a.bias = bias; a.B = B; a.H = H; a.W = W;
    a.npad = npad; a.nvalid = npad; a.scale = 1.f; a.o0 = out; a.ldo0 = npad;
    const int grid = ((M + bm[tile] - 1) / bm[tile]) * (npad / bn[tile]);
    auto go = [&]() {
        if (kh == 1 && kw == 5) launch<1, 5>(a, tile, grid, 0);
        else if (kh == 3 && kw == 3) launch<3, 3>(a, tile, grid, 0);
        else launch<1, 1>(a, tile, grid, 0);
    };
    for (int i = 0; i < 3; ++i) go();
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) go();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, tf = 2.0 * M * kh * kw * cin * npad / (us * 1e-6) / 1e12;
    printf("ABL=%d k=%dx%d cin=%d npad=%d tile=%d B=%d grid=%d: %.1f us  %.1f TF  err=%d\n", RAFT_ABL, kh, kw, cin, npad,
           tile, B, grid, us, tf, (int)hipGetLastError());
    return 0;
}
