// wake_probe.hip - where do the ~70 us that a 20-step burst loses ~210 us after its first kernel come from?  N dependent kernels of
// ~8 us (FMA work on every CU) are launched back to back after the GPU has idled; every kernel writes its own start and end
// (100 MHz wall clock) - a device-side timeline that involves the host only in how the burst is submitted and awaited:
//   A  eager launches, then the host SLEEPS (no HIP call) until long after the burst has finished
//   B  eager launches, then hipStreamSynchronize at once (what torch.cuda.synchronize does)
//   C  one hipGraph of the N kernels, host sleeps
//   D  one hipGraph, hipStreamSynchronize at once
//   E  as B, with hipEventRecord before and after (bench.py's timed region)
// Output per variant: burst length, sum of the stalls >= 3 us between consecutive kernels and where they happened.
// build: hipcc --offload-arch=gfx950 -O3 tools/wake_probe.hip -o /tmp/wake_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <unistd.h>
#include <time.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void work(unsigned long long *ts, int i, float *sink, int iters) {
    const unsigned long long t0 = wall_clock64();
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    for (int k = 0; k < iters; ++k) {
        x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 0.9999f, 0.25f); x2 = fmaf(x2, 1.0002f, 0.125f); x3 = fmaf(x3, 0.9998f, 0.0625f);
    }
    if (x0 + x1 + x2 + x3 == 12345.678f) sink[0] = x0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { ts[2 * i] = t0; ts[2 * i + 1] = wall_clock64(); }
}

static int report(const char *name, const std::vector<unsigned long long> &h, int n) {
    const double t0 = (double)h[0];
    double stall = 0.0;
    printf("%-58s burst %8.1f us  kernels %.2f us each:", name, (h[2 * n - 1] - t0) * 0.01, (h[1] - h[0]) * 0.01);
    int shown = 0;
    for (int i = 1; i < n; ++i) {
        const double gap = ((double)h[2 * i] - (double)h[2 * i - 1]) * 0.01;
        if (gap >= 3.0) {
            stall += gap;
            if (shown++ < 6) printf("  [%d @%.0f us: %.1f]", i, ((double)h[2 * i - 1] - t0) * 0.01, gap);
        }
    }
    printf("   stalls >= 3 us: %.1f us in total\n", stall);
    return 0;
}

int main(int argc, char **argv) {
    const int n = 100, iters = argc > 1 ? atoi(argv[1]) : 250;
    unsigned long long *ts; float *sink;
    CK(hipMalloc(&ts, sizeof(unsigned long long) * 2 * n));
    CK(hipMalloc(&sink, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned long long> h(2 * n);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s, ts, i, sink, iters);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
    const char *names[5] = {"A eager, host sleeps", "B eager, hipStreamSynchronize at once", "C hipGraph, host sleeps",
                            "D hipGraph, hipStreamSynchronize at once", "E eager between two hipEventRecord, synchronize at once"};
    for (int idle_ms = 0; idle_ms <= 20; idle_ms += 20) {
        printf("--- GPU idle for %d ms (+ the synchronise) before every burst\n", idle_ms);
        for (int rep = 0; rep < 2; ++rep)
            for (int v = 0; v < 5; ++v) {
                CK(hipMemsetAsync(ts, 0, sizeof(unsigned long long) * 2 * n, s));
                CK(hipStreamSynchronize(s));
                usleep(1000 * idle_ms);
                if (v == 4) CK(hipEventRecord(e0, s));
                if (v == 2 || v == 3) CK(hipGraphLaunch(ge, s));
                else {
                    // host view of the same burst: how long does every launch call take, and when does a slow one happen?
                    struct timespec a, b, first;
                    clock_gettime(CLOCK_MONOTONIC, &first);
                    double slow_sum = 0.0; int slow_n = 0; char where[256] = ""; int wl = 0;
                    for (int i = 0; i < n; ++i) {
                        clock_gettime(CLOCK_MONOTONIC, &a);
                        hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s, ts, i, sink, iters);
                        clock_gettime(CLOCK_MONOTONIC, &b);
                        const double us = (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3;
                        if (us >= 15.0) {
                            slow_sum += us; ++slow_n;
                            if (wl < 200) wl += snprintf(where + wl, sizeof(where) - wl, " [%d @%.0f us: %.0f]", i,
                                                         (a.tv_sec - first.tv_sec) * 1e6 + (a.tv_nsec - first.tv_nsec) * 1e-3, us);
                        }
                    }
                    clock_gettime(CLOCK_MONOTONIC, &b);
                    printf("    host: %d launches in %.0f us; launch calls >= 15 us: %d (%.0f us)%s\n", n,
                           (b.tv_sec - first.tv_sec) * 1e6 + (b.tv_nsec - first.tv_nsec) * 1e-3, slow_n, slow_sum, where);
                }
                if (v == 4) CK(hipEventRecord(e1, s));
                if (v == 0 || v == 2) usleep(20000);
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(h.data(), ts, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost));
                report(names[v], h, n);
            }
    }
    return 0;
}
