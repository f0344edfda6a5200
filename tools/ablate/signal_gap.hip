// What an ordering operation between two kernels of one stream costs the SECOND kernel (MI355X, ROCm 7.2): nothing in between,
// an event record (a side stream waits for it), or a stream write-value (the side stream waits for the value).
// hipcc --offload-arch=gfx950 -O2 tools/ablate/signal_gap.hip -o /tmp/signal_gap && /tmp/signal_gap
#include <hip/hip_runtime.h>
#ifndef SIG_VARIANT
#define SIG_VARIANT 0
#endif
#include <chrono>
#include <cstdio>

__global__ void spin(long cycles, int *sink) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 12345) *sink = 1;
}

// mode 4: the first kernel announces its own completion -- every wave, after its last store: agent-scope fence + one atomic
// increment of a counter -- and the side stream waits for the count with hipStreamWaitValue32 (>=): NO packet between the two
// kernels of the main stream
__global__ void spin_count(long cycles, unsigned *count, float *out, float tag) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    out[blockIdx.x * 256 + threadIdx.x] = tag;
#if SIG_VARIANT == 0      // every wave: fence + count
    __threadfence();
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#elif SIG_VARIANT == 1    // no fence (timing only: not a correct hand-off), every wave counts
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#elif SIG_VARIANT == 2    // one fence + one count per WORKGROUP, worth 4
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(count, 4u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
#elif SIG_VARIANT == 3    // nothing in the kernel (the side stream would wait forever: it does not wait in this variant)
#endif
}
__global__ void check_out(const float *out, int n, float tag, int *bad) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        if (out[i] < tag) atomicAdd(bad, 1);      // tags grow: a smaller one = the consumer ran early or read a stale line
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0));
    CK(hipStreamCreate(&s1));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    uint32_t *flag = nullptr;
    hipError_t fe = hipExtMallocWithFlags((void **)&flag, 64, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(fe));
    if (fe != hipSuccess) CK(hipMalloc((void **)&flag, 64));
    CK(hipMemset(flag, 0, 64));
    const long cyc = 2000;   // wall_clock64 ticks at 100 MHz: 20 us
    const int N = 300;
    float *outbuf; int *bad; unsigned *count;
    CK(hipMalloc((void **)&outbuf, 224 * 256 * 4)); CK(hipMalloc((void **)&bad, 4)); CK(hipMalloc((void **)&count, 64));
    CK(hipMemset(bad, 0, 4)); CK(hipMemset(count, 0, 64));
    unsigned ticket = 0;
    for (int mode = 0; mode < 6; ++mode) {   // 4 = timing of the counted-out kernel, 5 = the same with the visibility check on the side stream
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (mode >= 4) {
                    // every launch writes a new tag: a consumer that ran early (or read stale lines) sees the previous one.  (The
                    // producer of launch i + 1 may overwrite while the check of launch i still reads -- it then sees tag i + 1,
                    // counted as bad too: the check kernel is short and the main stream's second kernel lies in between.)
                    spin_count<<<224, 256, 0, s0>>>(cyc, count, outbuf, (float)(ticket + 1));
                    ticket += 224 * 4;
                    if (SIG_VARIANT != 3) CK(hipStreamWaitValue32(s1, count, ticket, hipStreamWaitValueGte, 0xffffffffu));
                    if (mode == 5) check_out<<<56, 256, 0, s1>>>(outbuf, 224 * 256, (float)(ticket - 224 * 4 + 1), bad);
                    spin<<<32, 256, 0, s1>>>(cyc / 4, nullptr);
                    spin<<<224, 256, 0, s0>>>(cyc, nullptr);
                    continue;
                }
                spin<<<224, 256, 0, s0>>>(cyc, nullptr);
                if (mode == 1) {
                    CK(hipEventRecord(ev, s0));
                    CK(hipStreamWaitEvent(s1, ev, 0));
                    spin<<<32, 256, 0, s1>>>(cyc / 4, nullptr);
                } else if (mode == 2) {
                    CK(hipStreamWriteValue32(s0, flag, (uint32_t)(rep * N + i + 1), 0));
                    CK(hipStreamWaitValue32(s1, flag, (uint32_t)(rep * N + i + 1), hipStreamWaitValueGte, 0xffffffffu));
                    spin<<<32, 256, 0, s1>>>(cyc / 4, nullptr);
                } else if (mode == 3) {
                    CK(hipEventRecord(ev, s0));      // a record nobody waits for
                }
                spin<<<224, 256, 0, s0>>>(cyc, nullptr);
            }
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (us < best) best = us;
        }
        const char *name[] = {"nothing between the two kernels", "event record + side stream waits for it", "stream write-value + side stream waits for the value",
                              "event record nobody waits for", "first kernel counts its waves out + side stream waits for the count", "the same + visibility check kernel on the side stream"};
        printf("%-55s %7.2f us per pair of 20 us kernels\n", name[mode], best);
    }
    int hbad = -1;
    CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    printf("mode 4 visibility check: %d stale elements seen by the side stream's consumer (must be 0)\n", hbad);
    return 0;
}
