// mfma_valu_probe.hip - does v_mfma_f32_16x16x4_f32 run beside VALU / transcendental work on one SIMD of MI355X?
// 256 workgroups x 1024 threads (four wavefronts per SIMD).  Per iteration and wavefront: 4 MFMAs and / or 16 v_sqrt + 8 v_pk_add.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_probe.hip -o tools/mfma_valu_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define PIN(v) do { asm volatile("" : "+v"(v)); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ v4f sq4(v4f v) {
    return (v4f){__builtin_amdgcn_sqrtf(__builtin_fabsf(v.x)), __builtin_amdgcn_sqrtf(__builtin_fabsf(v.y)),
                 __builtin_amdgcn_sqrtf(__builtin_fabsf(v.z)), __builtin_amdgcn_sqrtf(__builtin_fabsf(v.w))};
}
// MODE bit 0: MFMAs, bit 1: VALU work; SPEC: wavefront-specialised (wave groups alternate MFMA-only / VALU-only at twice the count);
// KIND 0: f32 16x16x4, 1: bf16 16x16x32, 2: VALU work = plain fma instead of sqrt
template <int MODE, bool SPEC, int KIND>
__global__ __launch_bounds__(1024) void k(float *o, int iters) {
    const int wave = threadIdx.x >> 6;
    v4f acc[4], m[4], x[4];
    for (int t = 0; t < 4; ++t) { acc[t] = (v4f){0, 0, 0, 0}; m[t] = acc[t]; x[t] = (v4f){1.f + threadIdx.x, 2.f, 3.f, 4.f}; }
    const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    bf8 ab, bb;
    for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)a; bb[e] = (__bf16)b; }
    bool doM = MODE & 1, doV = MODE & 2;
    int n = iters;
    if (SPEC) { doM = ((wave >> 2) & 1) == 0; doV = !doM; n = 2 * iters; }
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (doV) {
                if (KIND == 2) { acc[t] = __builtin_elementwise_fma(x[t], x[t], acc[t]); acc[(t + 1) & 3] = __builtin_elementwise_fma(x[t], acc[t], acc[(t + 1) & 3]); }
                else { x[t] = sq4(x[t]); acc[t & 1] += x[t]; }
                PIN(acc[t & 1]);
            }
            if (doM) {
                if (KIND == 1) m[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, m[t], 0, 0, 0);
                else m[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m[t], 0, 0, 0);
                PIN(m[t]);
            }
        }
    }
    v4f s = acc[0] + acc[1] + acc[2] + acc[3] + m[0] + m[1] + m[2] + m[3];
    o[blockIdx.x * 1024 + threadIdx.x] = s.x + s.y + s.z + s.w;
}
template <int MODE, bool SPEC, int KIND> int run(const char *name, float *o) {
    const int iters = 2000, reps = 5;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, SPEC, KIND>), dim3(256), dim3(1024), 0, 0, o, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<MODE, SPEC, KIND>), dim3(256), dim3(1024), 0, 0, o, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    // per SIMD and iteration: 4 wavefronts x (4 MFMAs, 24 VALU)
    printf("%-58s %9.1f us/kernel  %7.1f ns per (4 wavefronts x iteration)\n", name, us, us * 1e3 / iters);
    return 0;
}
int main() {
    float *o; CK(hipMalloc(&o, 256 * 1024 * 4));
    run<1, false, 0>("f32 16x16x4 MFMA only (4 per iteration)", o);
    run<2, false, 0>("VALU only (16 v_sqrt + 8 v_pk_add per iteration)", o);
    run<3, false, 0>("both, interleaved in every wavefront", o);
    run<3, true, 0>("both, wavefront-specialised (same totals)", o);
    run<2, false, 2>("VALU only (8 v_pk_fma x2 per iteration, no transcendental)", o);
    run<3, false, 2>("f32 MFMA + plain packed fma, interleaved", o);
    run<1, false, 1>("bf16 16x16x32 MFMA only (4 per iteration)", o);
    run<3, false, 1>("bf16 MFMA + sqrt work, interleaved", o);
    run<3, true, 1>("bf16 MFMA + sqrt work, wavefront-specialised", o);
    return 0;
}
