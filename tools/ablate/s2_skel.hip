// Skeleton of a tap-batched GEMM main loop (diagnostics only): how close a wave gets to the fp32-MFMA issue rate when the
// K loop is nothing but fragment loads (straight from L2, 16 bytes per lane, 1 KB contiguous per wave) and MFMAs.
//   s2_skel <C> <N> <B> <variant>        56 x 64 maps; variant = taps per wave: 9 (4 waves / WG) or 6 (6 waves / WG)
// wave tile: TP taps x TM row blocks x TN column blocks; V[rbgroup][tap][kk][rb][64 lanes][4], U[tap][kk][cb][64 lanes][4]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TP, int TM, int TN, int NW, int PF>
__global__ void __launch_bounds__(64 * NW, (TP * TM * TN * 4 > 160) ? 1 : 2) skel(const f32x4 *V, const f32x4 *U, float *out, int C, int N, int nrbg) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ncbg = N / (16 * TN);
    const int rbg = blockIdx.x / ncbg, cbg = blockIdx.x % ncbg;
    const int nkk = C / 16;
    f32x4 acc[TP][TM][TN];
#pragma unroll
    for (int t = 0; t < TP; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[t][i][j] = f32x4{0, 0, 0, 0};
    // this wave's taps: w * TP .. w * TP + TP - 1
    const f32x4 *vb = V + ((size_t)rbg * 36 + w * TP) * nkk * TM * 64 + lane;
    const f32x4 *ub = U + ((size_t)(w * TP) * nkk * (N / 16) + cbg * TN) * 64 + lane;
    f32x4 fa[PF + 1][TM], fb[PF + 1][TN];
    auto load = [&](int s, int slot) {   // step s = tap * nkk + kk
        const int t = s / nkk, kk = s % nkk;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][i] = vb[((size_t)t * nkk + kk) * TM * 64 + i * 64];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j] = ub[((size_t)t * nkk + kk) * (N / 16) * 64 + j * 64];
    };
    const int S = TP * nkk;
#pragma unroll
    for (int s = 0; s < PF; ++s) load(s, s);
    for (int t = 0; t < TP; ++t) {
        for (int kk0 = 0; kk0 < nkk; kk0 += (PF + 1)) {
#pragma unroll
            for (int u = 0; u < PF + 1; ++u) {
                const int s = t * nkk + kk0 + u;
                if (s + PF < S) load(s + PF, (u + PF) % (PF + 1));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][i][e], fb[u][j][e], acc[t][i][j], 0, 0, 0);
            }
        }
    }
    f32x4 s = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < TP; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) s += acc[t][i][j];
    if (s[0] == 12345.678f) out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int TP, int TM, int TN, int NW, int PF>
static void run(int C, int N, int B, const char *name) {
    const int nrb = B * 14, nrbg = nrb / TM, nkk = C / 16;
    const size_t nv = (size_t)nrbg * 36 * nkk * TM * 64, nu = (size_t)36 * nkk * (N / 16) * 64;
    f32x4 *V, *U;
    float *out;
    hipMalloc(&V, nv * 16); hipMalloc(&U, nu * 16); hipMalloc(&out, 4096);
    hipMemset(V, 0x3c, nv * 16); hipMemset(U, 0x3c, nu * 16);
    const int grid = nrbg * (N / (16 * TN));
    auto go = [&]() { skel<TP, TM, TN, NW, PF><<<grid, 64 * NW>>>(V, U, out, C, N, nrbg); };
    for (int i = 0; i < 3; ++i) go();
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) go();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, mfmas = (double)nrb * (N / 16) * 36 * (C / 4);
    printf("%-28s C=%d N=%d B=%d grid=%d x %d thr: %7.2f us  %6.1f TF executed (err %s)\n", name, C, N, B, grid, 64 * NW, us,
           mfmas * 2048 / us / 1e6, hipGetErrorString(hipGetLastError()));
    hipFree(V); hipFree(U); hipFree(out);
}

int main(int argc, char **argv) {
    const int C = atoi(argv[1]), N = atoi(argv[2]), B = atoi(argv[3]);
    run<9, 2, 2, 4, 1>(C, N, B, "9 taps 2rb x 2cb pf1");
    run<9, 2, 2, 4, 3>(C, N, B, "9 taps 2rb x 2cb pf3");
    run<9, 1, 2, 4, 3>(C, N, B, "9 taps 1rb x 2cb pf3");
    run<9, 1, 4, 4, 3>(C, N, B, "9 taps 1rb x 4cb pf3");
    run<9, 2, 4, 4, 3>(C, N, B, "9 taps 2rb x 4cb pf3 (1w/simd)");
    run<6, 2, 2, 6, 3>(C, N, B, "6 taps 2rb x 2cb pf3 (6 waves)");
    run<6, 2, 4, 6, 3>(C, N, B, "6 taps 2rb x 4cb pf3 (6 waves)");
    run<12, 2, 2, 3, 3>(C, N, B, "12 taps 2rb x 2cb pf3 (3 waves)");
    return 0;
}
