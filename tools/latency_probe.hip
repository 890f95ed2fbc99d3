// latency_probe.hip - calibrate per-kernel overhead and dependent-load latency on MI355X.
// build: hipcc --offload-arch=gfx950 -O3 tools/latency_probe.hip -o gpurun_out/latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_null() {}
__global__ void k_store(float *o) { o[blockIdx.x * blockDim.x + threadIdx.x] = 1.f; }
// depth dependent loads per thread through an index array (coalesced chase)
template <int DEPTH>
__global__ void k_chase(const int *idx, float *o, int n) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) p = idx[p];
    o[blockIdx.x * blockDim.x + threadIdx.x] = (float)p;
}
struct Big { float *p[24]; int v[16]; };
__global__ void k_bigarg(Big b) { b.p[0][blockIdx.x * blockDim.x + threadIdx.x] = (float)b.v[3]; }
__global__ void k_rmw(float *a) { a[blockIdx.x * blockDim.x + threadIdx.x] += 1.f; }
__global__ void k_atomic(float *a) { atomicAdd(&a[blockIdx.x * blockDim.x + threadIdx.x], 1.f); }

int main() {
    const int n = 256 * 1000;   // 1000 blocks of 256
    int *idx; float *o, *a;
    CK(hipMalloc(&idx, n * sizeof(int))); CK(hipMalloc(&o, n * sizeof(float))); CK(hipMalloc(&a, n * sizeof(float)));
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = (i * 7919 + 12345) % n;   // scattered permutation-ish
    CK(hipMemcpy(idx, h.data(), n * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemset(a, 0, n * sizeof(float)));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Big big{}; big.p[0] = o; big.v[3] = 5;
    const int REP = 200;
    auto run = [&](const char *name, auto launch) -> int {
        for (int r = 0; r < 20; ++r) launch();
        CK(hipStreamSynchronize(s));
        // capture into a graph like the bench does
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int r = 0; r < REP; ++r) launch();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %8.3f us per kernel (graph of %d)\n", name, 1e3 * ms / REP, REP);
        return 0;
    };
    run("null  1x64", [&] { hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, s); });
    run("null  1000x256", [&] { hipLaunchKernelGGL(k_null, dim3(1000), dim3(256), 0, s); });
    run("store 1000x256", [&] { hipLaunchKernelGGL(k_store, dim3(1000), dim3(256), 0, s, o); });
    run("bigarg 1000x256", [&] { hipLaunchKernelGGL(k_bigarg, dim3(1000), dim3(256), 0, s, big); });
    run("chase1 1000x256", [&] { hipLaunchKernelGGL(k_chase<1>, dim3(1000), dim3(256), 0, s, idx, o, n); });
    run("chase2 1000x256", [&] { hipLaunchKernelGGL(k_chase<2>, dim3(1000), dim3(256), 0, s, idx, o, n); });
    run("chase4 1000x256", [&] { hipLaunchKernelGGL(k_chase<4>, dim3(1000), dim3(256), 0, s, idx, o, n); });
    run("chase8 1000x256", [&] { hipLaunchKernelGGL(k_chase<8>, dim3(1000), dim3(256), 0, s, idx, o, n); });
    run("chase4 250x256", [&] { hipLaunchKernelGGL(k_chase<4>, dim3(250), dim3(256), 0, s, idx, o, n); });
    run("rmw 1000x256", [&] { hipLaunchKernelGGL(k_rmw, dim3(1000), dim3(256), 0, s, a); });
    run("atomic 1000x256", [&] { hipLaunchKernelGGL(k_atomic, dim3(1000), dim3(256), 0, s, a); });
    run("rmw 16x256", [&] { hipLaunchKernelGGL(k_rmw, dim3(16), dim3(256), 0, s, a); });
    return 0;
}
