// Standalone ablation driver for the 1-D Winograd F(4,5) / F(2,5) GRU kernels (diagnostics only).  Built once per
// RAFT_WINO1D_ABL value (see conv_wino1d.h).
//   ablate_wino1d_n <zr|q> <axis 0|1> <mo 2|4> [B] [reps]       (56 x 64 feature maps, K = 128 + 128)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../tf_raft_amd/csrc/conv_wino1d.hip"

int main(int argc, char **argv) {
    const bool zr = !strcmp(argv[1], "zr");
    const int axis = atoi(argv[2]), mo = atoi(argv[3]);
    const int B = argc > 4 ? atoi(argv[4]) : 4, reps = argc > 5 ? atoi(argv[5]) : 50;
    const int H = 56, W = 64, M = B * H * W, cin = 256, npad = zr ? 256 : 128, hid = 128;
    float *x0, *x1, *w, *bias, *o0, *o1, *h, *z, *ctx;
    const size_t nw = (size_t)(mo + 4) * cin * npad;
    hipMalloc(&x0, (size_t)M * 128 * 4); hipMalloc(&x1, (size_t)M * 256 * 4); hipMalloc(&w, nw * 4); hipMalloc(&bias, npad * 4);
    hipMalloc(&o0, (size_t)M * 128 * 4); hipMalloc(&o1, (size_t)M * 128 * 4); hipMalloc(&h, (size_t)M * 128 * 4);
    hipMalloc(&z, (size_t)M * 128 * 4); hipMalloc(&ctx, (size_t)M * 768 * 4);
    std::vector<float> hx((size_t)M * 256), hw(nw);
    srand(1);
    for (auto &v : hx) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto &v : hw) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f;
    hipMemcpy(x0, hx.data(), (size_t)M * 128 * 4, hipMemcpyHostToDevice);
    hipMemcpy(x1, hx.data(), (size_t)M * 256 * 4, hipMemcpyHostToDevice);
    hipMemcpy(h, hx.data(), (size_t)M * 128 * 4, hipMemcpyHostToDevice);
    hipMemcpy(z, hx.data(), (size_t)M * 128 * 4, hipMemcpyHostToDevice);
    hipMemset(ctx, 0, (size_t)M * 768 * 4);
    hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, npad * 4);
    ConvArgs a = {};
    a.a0 = x0; a.lda0 = 128; a.c0 = 128; a.a1 = x1 + 128; a.lda1 = 256; a.c1 = 128;
    a.wp = w; a.bias = bias; a.B = B; a.H = H; a.W = W; a.npad = npad; a.nvalid = npad; a.scale = 1.f;
    a.o0 = o0; a.ldo0 = 128; a.e0 = h; a.lde0 = 128; a.init = ctx; a.ldi = 768;
    if (zr) { a.hid = hid; a.o1 = o1; a.ldo1 = 128; } else { a.e1 = z; a.lde1 = 128; }
    const int kh = axis ? 5 : 1, kw = axis ? 1 : 5;
    auto go = [&]() { return raft_launch_conv_wino1d(a, kh, kw, zr ? EPI_GRU_ZR : EPI_GRU_Q, 0, mo); };
    for (int i = 0; i < 3; ++i) if (go() != 0) { printf("launch failed\n"); return 1; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) go();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, flops = 2.0 * M * 5.0 * cin * npad, ratio = mo == 4 ? 2.5 : 10.0 / 6.0;
    printf("wino1d abl=%2d %s axis=%d F(%d,5) B=%d: %7.2f us  %6.1f TF algorithmic  %6.1f TF executed\n", RAFT_WINO1D_ABL, argv[1],
           axis, mo, B, us, flops / us / 1e6, flops / ratio / us / 1e6);
    return 0;
}
