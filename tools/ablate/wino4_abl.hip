// Standalone ablation driver for the Winograd F(4x4,3x3) kernel (diagnostics only).  Built once per variant (wino4_abl.sh).
//   ablate_wino4_<name> <cin> <npad> [B] [reps]            (56 x 64 feature maps, relu epilogue)
// RAFT_WINO4_ABL bits: 1 no weight loads in the loop, 2 no stage 1 (LDS patch reads + B^T over rows), 4 no halo staging
// (global loads return nothing), 8 no epilogue stores, 16 no stage 2
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../tf_raft_amd/csrc/conv_wino4.hip"

int main(int argc, char **argv) {
    const int cin = atoi(argv[1]), npad = atoi(argv[2]);
    const int B = argc > 3 ? atoi(argv[3]) : 4, reps = argc > 4 ? atoi(argv[4]) : 50;
    const int H = 56, W = 64, M = B * H * W;
    float *x, *w, *bias, *out;
    const size_t nx = (size_t)M * cin, nw = (size_t)36 * cin * npad;
    hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&bias, npad * 4); hipMalloc(&out, (size_t)M * npad * 4);
    std::vector<float> hx(nx), hw(nw);
    srand(1);
    for (auto &v : hx) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto &v : hw) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f;
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, npad * 4);
    ConvArgs a = {};
    a.a0 = x; a.lda0 = cin; a.c0 = cin; a.wp = w; a.bias = bias; a.B = B; a.H = H; a.W = W;
    a.npad = npad; a.nvalid = npad; a.scale = 1.f; a.o0 = out; a.ldo0 = npad;
    auto go = [&]() { raft_launch_conv_wino4(a, EPI_RELU, 0); };
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
    const double us = ms * 1e3 / reps, flops = 2.0 * M * 9.0 * cin * npad;
    printf("cin=%d npad=%d B=%d: %7.2f us  %6.1f TF algorithmic  %6.1f TF executed\n", cin, npad, B, us, flops / us / 1e6,
           flops / 4.0 / us / 1e6);
    return 0;
}
