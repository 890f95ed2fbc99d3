// gridbar_probe.hip - cost of an in-kernel grid-wide barrier on MI355X (8 XCDs, L2 per XCD): the number a
// "cooperative strict step" (phase hand-offs inside ONE persistent kernel instead of 5 launches) would pay per phase.
// build: hipcc --offload-arch=gfx950 -O3 tools/gridbar_probe.hip -o tools/gridbar_probe.bin
// The spin is BOUNDED (err flag, sticky) - the probe cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned *cnt, unsigned nblk, unsigned &phase, int *err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                              // release: this block's writes
        const unsigned target = (phase + 1) * nblk;
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 400000) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __threadfence();                                              // acquire: the other blocks' writes
    }
    ++phase;
    __syncthreads();
}

// NB phases; MODE 0: barrier only; MODE 1: every phase each block writes one 1 KB row and, after the barrier,
// reads the row ANOTHER block (other XCD: +1) wrote in that phase and checks it
template <int MODE>
__global__ __launch_bounds__(256) void k_phases(unsigned *cnt, int *err, float *buf, int nb, int *bad) {
    unsigned phase = 0;
    const unsigned nblk = gridDim.x;
    float acc = 0.f;
    for (int p = 0; p < nb; ++p) {
        if (MODE == 1) buf[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)(p * 1000 + blockIdx.x);
        grid_barrier(cnt, nblk, phase, err);
        if (MODE == 1) {
            const unsigned src = (blockIdx.x + 1) % nblk;
            const float v = __builtin_nontemporal_load(&buf[(size_t)src * 256 + threadIdx.x]);
            if (v != (float)(p * 1000 + src)) atomicAdd(bad, 1);
            acc += v;
            grid_barrier(cnt, nblk, phase, err);                      // before the rows are overwritten
        }
    }
    if (MODE == 1 && acc == -1.f) buf[0] = acc;
}

int main() {
    unsigned *cnt; int *err, *bad; float *buf;
    CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&buf, 4096 * 256 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, int mode, int blocks, int nb) -> int {
        float best = 1e30f; int herr = 0, hbad = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemsetAsync(cnt, 0, 4, s)); CK(hipMemsetAsync(err, 0, 4, s)); CK(hipMemsetAsync(bad, 0, 4, s));
            CK(hipEventRecord(e0, s));
            if (mode == 0) hipLaunchKernelGGL(k_phases<0>, dim3(blocks), dim3(256), 0, s, cnt, err, buf, nb, bad);
            else hipLaunchKernelGGL(k_phases<1>, dim3(blocks), dim3(256), 0, s, cnt, err, buf, nb, bad);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
            int e_, b_; CK(hipMemcpy(&e_, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b_, bad, 4, hipMemcpyDeviceToHost));
            herr |= e_; hbad += b_;
        }
        printf("%-44s blocks %4d  phases %4d  %9.3f us total  %7.3f us per phase  timeout %d  stale reads %d\n",
               name, blocks, nb, 1e3 * best, nb ? 1e3 * best / nb : 0.f, herr, hbad);
        return 0;
    };
    run("launch only (0 phases)", 0, 256, 0);
    for (int blocks : {256, 512, 1024}) run("barrier only", 0, blocks, 200);
    for (int blocks : {256, 512, 1024}) run("write row | barrier | read neighbour | barrier", 1, blocks, 100);
    return 0;
}
