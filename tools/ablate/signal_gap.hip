// What an ordering operation between two kernels of one stream costs the SECOND kernel (MI355X, ROCm 7.2): nothing in between,
// an event record (a side stream waits for it), or a stream write-value (the side stream waits for the value).
// hipcc --offload-arch=gfx950 -O2 tools/ablate/signal_gap.hip -o /tmp/signal_gap && /tmp/signal_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin(long cycles, int *sink) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 12345) *sink = 1;
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
    for (int mode = 0; mode < 4; ++mode) {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
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
                              "event record nobody waits for"};
        printf("%-55s %7.2f us per pair of 20 us kernels\n", name[mode], best);
    }
    return 0;
}
