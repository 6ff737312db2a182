// What one hipEventRecord between two dependent kernels costs on the GPU timeline (gfx950, ROCm 7.2), by event flags.
//   hipcc -O2 --offload-arch=gfx950 tools/event_gap_probe.hip -o build/probe/event_gap_probe && build/probe/event_gap_probe
// A chain of 64 short kernels (~5 us each) on one stream, timed from outside; between every two kernels: nothing, or an
// event record with the given creation flags.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(unsigned* out, int iters) {
    unsigned x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
    if (x == 12345u) out[0] = x;
}

static double run(hipStream_t st, unsigned* buf, int kernels, int mode, unsigned flags, int iters) {
    std::vector<hipEvent_t> ev(kernels);
    if (mode) for (auto& e : ev) (void)hipEventCreateWithFlags(&e, flags);
    double best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < kernels; ++k) {
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, buf, iters);
            if (mode) (void)hipEventRecord(ev[k], st);
        }
        (void)hipStreamSynchronize(st);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < best) best = us;
    }
    if (mode) for (auto& e : ev) (void)hipEventDestroy(e);
    return best / kernels;
}

int main() {
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned* buf;
    (void)hipMalloc(&buf, 4096);
    for (int iters : {2000, 20000}) {
        const double base = run(st, buf, 64, 0, 0, iters);
        printf("kernel of %d iterations: %.2f us per kernel with no events\n", iters, base);
        struct { const char* name; unsigned flags; } cases[] = {
            {"hipEventDefault", hipEventDefault},
            {"hipEventDisableTiming", hipEventDisableTiming},
            {"hipEventDisableSystemFence", hipEventDisableSystemFence},
            {"hipEventDisableTiming|DisableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence},
            {"hipEventReleaseToDevice", hipEventReleaseToDevice},
            {"hipEventDisableTiming|ReleaseToDevice", hipEventDisableTiming | hipEventReleaseToDevice},
        };
        for (auto& c : cases) printf("  + event record (%s): %+.2f us per kernel\n", c.name, run(st, buf, 64, 1, c.flags, iters) - base);
    }
    return 0;
}
