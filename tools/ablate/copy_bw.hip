// HBM copy-bandwidth variants (diagnostics): which streaming-copy shape gives the best read+write rate on this box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) copy_flat(const f32x4 *__restrict__ s, f32x4 *__restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
template <int U>
__global__ void __launch_bounds__(256) copy_unroll(const f32x4 *__restrict__ s, f32x4 *__restrict__ d, size_t n) {
    size_t i = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = i + u * 256 < n ? s[i + u * 256] : f32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) d[i + u * 256] = v[u];
}
template <int U>
__global__ void __launch_bounds__(256) copy_unroll_nt(const f32x4 *__restrict__ s, f32x4 *__restrict__ d, size_t n) {
    size_t i = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = i + u * 256 < n ? __builtin_nontemporal_load(s + i + u * 256) : f32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) __builtin_nontemporal_store(v[u], d + i + u * 256);
}
__global__ void __launch_bounds__(256) copy_stride(const f32x4 *__restrict__ s, f32x4 *__restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256;
    for (; i < n; i += st) d[i] = s[i];
}
int main() {
    for (size_t mb : {256, 1024}) {
        const size_t n = (mb << 20) / 16;
        f32x4 *a, *b;
        hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
        hipMemset(a, 1, n * 16); hipMemset(b, 0, n * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](const char *name, auto f) {
            for (int i = 0; i < 2; ++i) f();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 10; ++i) f();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%4zu MiB %-18s %7.1f GB/s\n", mb, name, 2.0 * n * 16 * 10 / (ms * 1e-3) / 1e9);
        };
        run("flat", [&] { copy_flat<<<(n + 255) / 256, 256>>>(a, b, n); });
        run("unroll4", [&] { copy_unroll<4><<<(n + 1023) / 1024, 256>>>(a, b, n); });
        run("unroll8", [&] { copy_unroll<8><<<(n + 2047) / 2048, 256>>>(a, b, n); });
        run("unroll4_nt", [&] { copy_unroll_nt<4><<<(n + 1023) / 1024, 256>>>(a, b, n); });
        run("unroll8_nt", [&] { copy_unroll_nt<8><<<(n + 2047) / 2048, 256>>>(a, b, n); });
        run("stride_8192", [&] { copy_stride<<<8192, 256>>>(a, b, n); });
        run("stride_2048", [&] { copy_stride<<<2048, 256>>>(a, b, n); });
        run("hipMemcpyD2D", [&] { hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0); });
        hipFree(a); hipFree(b);
    }
    return 0;
}
