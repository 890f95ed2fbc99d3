// icache_probe.hip - what does COLD straight-line code cost on MI355X?  Every kernel of the training step starts with a cold
// instruction cache (the launches alternate between different kernels) and most of their code runs once per wavefront.
// A kernel of NI straight-line VALU instructions (8-byte encodings, four independent dependency chains) is executed twice
// inside the same wavefront: pass 1 fetches its code cold, pass 2 hot; between launches another kernel with ~48 KB of code runs.
// build: hipcc --offload-arch=gfx950 -O3 tools/icache_probe.hip -o tools/icache_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NI>
__global__ __launch_bounds__(256) void k_straight(float *o, unsigned long long *tl) {
    float x0 = threadIdx.x * 0.5f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    const float y = 1.0001f;
    unsigned long long t0 = wall_clock64(), t1 = 0;
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int i = 0; i < NI / 4; ++i) {
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x0) : "v"(y));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x1) : "v"(y));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x2) : "v"(y));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x3) : "v"(y));
        }
        if (r == 0) t1 = wall_clock64();
    }
    unsigned long long t2 = wall_clock64();
    o[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        tl[2 * w] = t1 - t0; tl[2 * w + 1] = t2 - t1;
    }
}
// "another kernel": ~48 KB of different code, executed by every wavefront
__global__ __launch_bounds__(256) void k_other(float *o) {
    float x0 = threadIdx.x * 0.25f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    const float y = 0.9999f;
#pragma unroll
    for (int i = 0; i < 1536; ++i) {
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x0) : "v"(y));
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x1) : "v"(y));
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x2) : "v"(y));
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x3) : "v"(y));
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <int NI>
int probe(hipStream_t s, float *o, unsigned long long *tl, int blocks, bool with_other) {
    const int waves = blocks * 4;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int r = 0; r < 20; ++r) {
        if (with_other) hipLaunchKernelGGL(k_other, dim3(256), dim3(256), 0, s, o);
        hipLaunchKernelGGL(k_straight<NI>, dim3(blocks), dim3(256), 0, s, o, tl);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(2 * waves);
    CK(hipMemcpy(h.data(), tl, 2 * waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> c(waves), w(waves);
    for (int i = 0; i < waves; ++i) { c[i] = h[2 * i] * 10.0; w[i] = h[2 * i + 1] * 10.0; }   // 100 MHz clock -> ns
    std::sort(c.begin(), c.end()); std::sort(w.begin(), w.end());
    printf("NI %5d (%6d B of code) x %4d workgroups, %s: first pass p50 %7.0f ns p90 %7.0f ns | second pass p50 %7.0f ns | "
           "cold cost p50 %6.0f ns = %.2f ns per instruction, %.0f ns per 64-byte line\n", NI, NI * 8, blocks,
           with_other ? "another kernel in between" : "same kernel back to back   ", c[waves / 2], c[waves * 9 / 10], w[waves / 2],
           c[waves / 2] - w[waves / 2], (c[waves / 2] - w[waves / 2]) / NI, (c[waves / 2] - w[waves / 2]) / (NI * 8 / 64.0));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return 0;
}

int main() {
    float *o; unsigned long long *tl;
    CK(hipMalloc(&o, 1024 * 256 * sizeof(float))); CK(hipMalloc(&tl, 2 * 4096 * sizeof(unsigned long long)));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int other = 1; other >= 0; --other) {
        probe<256>(s, o, tl, 256, other); probe<1024>(s, o, tl, 256, other); probe<4096>(s, o, tl, 256, other);
        probe<1024>(s, o, tl, 1000, other);
    }
    return 0;
}
