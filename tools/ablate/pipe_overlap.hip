// Can the fp32 matrix pipe of a gfx950 SIMD run beside other vector work?  (MI355X, ROCm 7.2)
// Every wave runs `iters` trips of ONE kind of segment; the two waves of a SIMD (512-thread workgroups, one per CU: waves w and
// w + 4 share SIMD w & 3) run either the same kind or two different kinds.  Kinds: M = 8 independent v_mfma_f32_16x16x4_f32
// (8 x 32 cycles); V = 64 independent v_pk_fma_f32; S = 64 independent v_fma_f32; L = 16 ds_read_b128; B = 8 v_mfma_f32_16x16x16_bf16.
// For each pair (a, b): time of a alone (one wave per SIMD), b alone, and a beside b.  "beside" ~ max(alone) => the pipes
// overlap; ~ sum => they share an issue port or a datapath.  Also: one wave alternating a- and b-segments (in-order issue).
// hipcc --offload-arch=gfx950 -O2 tools/ablate/pipe_overlap.hip -o tools/ablate/pipe_overlap && tools/ablate/pipe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void segment(f32x4 (&acc)[8], f32x2 (&pk)[16], float (&sc)[16], const float *lds, float a, float b, float &sink) {
    if constexpr (KIND == 'M') {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + j, b - j, acc[j], 0, 0, 0);
    } else if constexpr (KIND == 'P') {      // paced: the wave parks on s_nop (28 cycles) instead of on the busy matrix pipe
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + j, b - j, acc[j], 0, 0, 0);
            asm volatile("s_nop 6" ::: "memory");
        }
    } else if constexpr (KIND == 'Q') {      // paced a little short (24 cycles of nops)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + j, b - j, acc[j], 0, 0, 0);
            asm volatile("s_nop 3" ::: "memory");
        }
    } else if constexpr (KIND == 'I') {      // one wave: every MFMA followed by 4 independent v_pk_fma_f32
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + j, b - j, acc[j], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[(j * 4 + k) & 15]) : "v"(f32x2{a, b}), "v"(f32x2{b, a}));
        }
    } else if constexpr (KIND == 'J') {      // one wave: every MFMA followed by 8 independent v_fma_f32
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + j, b - j, acc[j], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(sc[(j * 8 + k) & 15]) : "v"(a), "v"(b));
        }
    } else if constexpr (KIND == 'K') {      // one wave: every MFMA followed by 2 ds_read_b128, consumed a segment later
        f32x4 t[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + j, b - j, acc[j], 0, 0, 0);
            t[2 * j] = *(const volatile f32x4 *)(lds + ((threadIdx.x * 4 + j * 4096) & 8191));
            t[2 * j + 1] = *(const volatile f32x4 *)(lds + ((threadIdx.x * 4 + j * 4096 + 2048) & 8191));
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) sink += t[j][0];
    } else if constexpr (KIND == 'B') {
        bf16x8 x, y;
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] = (__bf16)(a + k); y[k] = (__bf16)(b - k); }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[j], 0, 0, 0);
    } else if constexpr (KIND == 'V') {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[j]) : "v"(f32x2{a, b}), "v"(f32x2{b, a}));
    } else if constexpr (KIND == 'S') {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(sc[j]) : "v"(a), "v"(b));
    } else if constexpr (KIND == 'L') {
        f32x4 t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) t[j] = *(const volatile f32x4 *)(lds + ((threadIdx.x * 4 + j * 2048) & 8191));
#pragma unroll
        for (int j = 0; j < 16; ++j) sink += t[j][0];
    }
}

// KA for waves 0..3 (and, with ONE = 1, the only waves), KB for waves 4..7; ALT = 1: every wave alternates KA and KB segments
template <int KA, int KB, int ALT>
__global__ void __launch_bounds__(512, 1) probe(float *out, int iters, unsigned *tend = nullptr, int prio_b = 0) {
    const long t_start = wall_clock64();
    __shared__ float lds[8192 + 64];
    for (int i = threadIdx.x; i < 8192 + 64; i += blockDim.x) lds[i] = 0.001f * i;
    __syncthreads();
    const int t = threadIdx.x, w = t >> 6;
    f32x4 acc[8];
    f32x2 pk[16];
    float sc[16];
    float sink = 0.f;
    const float a = 0.5f + 0.0013f * (t & 255), b = 1.25f - 0.0021f * (t & 127);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j) { pk[j] = f32x2{a, b}; sc[j] = a; }
    if (ALT) {
        for (int i = 0; i < iters; ++i) {
            segment<KA>(acc, pk, sc, lds, a, b, sink);
            segment<KB>(acc, pk, sc, lds, a, b, sink);
        }
    } else if (w < 4) {
        for (int i = 0; i < iters; ++i) segment<KA>(acc, pk, sc, lds, a, b, sink);
    } else {
        if (prio_b) __builtin_amdgcn_s_setprio(3);
        for (int i = 0; i < iters; ++i) segment<KB>(acc, pk, sc, lds, a, b, sink);
    }
    if (tend && (t & 63) == 0) tend[blockIdx.x * 8 + w] = (unsigned)(wall_clock64() - t_start);   // 100 MHz ticks
    float s = sink;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][3];
#pragma unroll
    for (int j = 0; j < 16; ++j) s += pk[j][0] + pk[j][1] + sc[j];
    out[blockIdx.x * 512 + t] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int KA, int KB, int ALT>
static float run(float *out, int threads, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KA, KB, ALT><<<256, threads>>>(out, iters);           // warm
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        probe<KA, KB, ALT><<<256, threads>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}

template <int KA, int KB>
static void pair(float *out, const char *na, const char *nb, int iters) {
    const float a1 = run<KA, KA, 0>(out, 256, iters);            // one wave per SIMD, kind a
    const float b1 = run<KB, KB, 0>(out, 256, iters);
    const float a2 = run<KA, KA, 0>(out, 512, iters);            // two waves per SIMD, both kind a
    const float b2 = run<KB, KB, 0>(out, 512, iters);
    const float ab = run<KA, KB, 0>(out, 512, iters);            // a beside b on every SIMD
    const float alt1 = run<KA, KB, 1>(out, 256, iters);          // one wave alternating a / b segments
    const float alt2 = run<KA, KB, 1>(out, 512, iters);          // two such waves per SIMD
    printf("%-22s alone %7.1f us | %-22s alone %7.1f us | 2x%s %7.1f | 2x%s %7.1f | a beside b %7.1f (max %.1f, sum %.1f) | one wave a;b %7.1f | two waves a;b %7.1f (2 x sum %.1f)\n",
           na, a1, nb, b1, na, a2, nb, b2, ab, a1 > b1 ? a1 : b1, a1 + b1, alt1, alt2, 2 * (a1 + b1));
}

// when do the waves of each kind finish (mean over workgroups, us since the workgroup started)?
template <int KA, int KB>
static void ends(float *out, unsigned *tend, const char *na, const char *nb, int iters, int prio_b) {
    static unsigned host[256 * 8];
    probe<KA, KB, 0><<<256, 512>>>(out, iters, tend, prio_b);
    hipDeviceSynchronize();
    probe<KA, KB, 0><<<256, 512>>>(out, iters, tend, prio_b);
    hipDeviceSynchronize();
    hipMemcpy(host, tend, sizeof(host), hipMemcpyDeviceToHost);
    double ea = 0, eb = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? ea : eb) += host[b * 8 + w] * 0.01;
    printf("   waves 0-3 = %-22s finish at %7.1f us, waves 4-7 = %-22s%s finish at %7.1f us\n", na, ea / 1024, nb, prio_b ? " (s_setprio 3)" : "", eb / 1024);
}

int main() {
    float *out;
    unsigned *tend;
    CK(hipMalloc((void **)&out, 256 * 512 * 4));
    CK(hipMalloc((void **)&tend, 256 * 8 * 4));
    const int iters = 2000;
    printf("one wave per SIMD: M %.1f us, P (mfma + s_nop 6 = 7 wait states) %.1f, Q (mfma + s_nop 3 = 4 wait states) %.1f, I (mfma + 4 v_pk_fma) %.1f, J (mfma + 8 v_fma) %.1f, K (mfma + 2 ds_read_b128) %.1f, V %.1f, S %.1f, L %.1f\n",
           run<'M', 'M', 0>(out, 256, iters), run<'P', 'P', 0>(out, 256, iters), run<'Q', 'Q', 0>(out, 256, iters), run<'I', 'I', 0>(out, 256, iters),
           run<'J', 'J', 0>(out, 256, iters), run<'K', 'K', 0>(out, 256, iters), run<'V', 'V', 0>(out, 256, iters), run<'S', 'S', 0>(out, 256, iters), run<'L', 'L', 0>(out, 256, iters));
    printf("two waves per SIMD, both the same: M %.1f us, P %.1f, I %.1f, J %.1f, K %.1f\n",
           run<'M', 'M', 0>(out, 512, iters), run<'P', 'P', 0>(out, 512, iters), run<'I', 'I', 0>(out, 512, iters), run<'J', 'J', 0>(out, 512, iters), run<'K', 'K', 0>(out, 512, iters));
    ends<'P', 'V'>(out, tend, "paced mfma f32", "v_pk_fma_f32", iters, 0);
    ends<'V', 'P'>(out, tend, "v_pk_fma_f32", "paced mfma f32", iters, 0);
    ends<'Q', 'V'>(out, tend, "paced(24) mfma f32", "v_pk_fma_f32", iters, 0);
    ends<'P', 'S'>(out, tend, "paced mfma f32", "v_fma_f32", iters, 0);
    ends<'P', 'L'>(out, tend, "paced mfma f32", "ds_read_b128", iters, 0);
    ends<'P', 'P'>(out, tend, "paced mfma f32", "paced mfma f32", iters, 0);
    ends<'M', 'V'>(out, tend, "mfma f32", "v_pk_fma_f32", iters, 0);
    ends<'V', 'M'>(out, tend, "v_pk_fma_f32", "mfma f32", iters, 0);
    ends<'M', 'V'>(out, tend, "mfma f32", "v_pk_fma_f32", iters, 1);
    ends<'M', 'L'>(out, tend, "mfma f32", "ds_read_b128", iters, 0);
    ends<'L', 'M'>(out, tend, "ds_read_b128", "mfma f32", iters, 0);
    ends<'M', 'L'>(out, tend, "mfma f32", "ds_read_b128", iters, 1);
    ends<'M', 'M'>(out, tend, "mfma f32", "mfma f32", iters, 0);
    ends<'B', 'V'>(out, tend, "mfma bf16", "v_pk_fma_f32", iters, 0);
    ends<'V', 'B'>(out, tend, "v_pk_fma_f32", "mfma bf16", iters, 0);
    pair<'M', 'V'>(out, "8 mfma f32 16x16x4", "64 v_pk_fma_f32", iters);
    pair<'M', 'S'>(out, "8 mfma f32 16x16x4", "64 v_fma_f32", iters);
    pair<'M', 'L'>(out, "8 mfma f32 16x16x4", "16 ds_read_b128", iters);
    pair<'B', 'V'>(out, "8 mfma bf16 16x16x32", "64 v_pk_fma_f32", iters);
    pair<'M', 'B'>(out, "8 mfma f32 16x16x4", "8 mfma bf16 16x16x32", iters);
    return 0;
}
