// mfma_probe.hip - calibrate fp32 MFMA issue rate and operand-load patterns on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float *o, int iters) {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = (f32x4){0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    o[blockIdx.x * 256 + threadIdx.x] = s;
}
// MFMA-operand-shaped loads: lane (m=l&15,q=l>>4) reads row (row0+m), 16 B at k*16+q*4 floats
__global__ __launch_bounds__(256) void k_ld_frag(const float *A, float *o, int D, int nrows) {
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float *p = A + (size_t)((wave * 16 + m) % nrows) * D + q * 4;
    float4 s = make_float4(0, 0, 0, 0);
#pragma unroll 5
    for (int k = 0; k < D / 16; ++k) { float4 v = *(const float4 *)(p + k * 16); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    o[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}
// same bytes, row-contiguous: wave reads 16 rows, each row as 16-byte pieces across 64 lanes
__global__ __launch_bounds__(256) void k_ld_row(const float *A, float *o, int D, int nrows) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    float4 s = make_float4(0, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const float *p = A + (size_t)((wave * 16 + r) % nrows) * D;
        for (int c = lane; c < D / 4; c += 64) { float4 v = *(const float4 *)(p + c * 4); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
    o[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}
// row gather like the row-wise kernels: one wave per row, rows picked by an index array
__global__ __launch_bounds__(256) void k_gather(const float *T, const long *idx, float *out, int D, int n, int nsrc) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= n) return;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int s = 0; s < nsrc; ++s) {
        const float *p = T + idx[(size_t)s * n + wave] * (size_t)D;
        for (int c = lane; c < D / 4; c += 64) { float4 v = *(const float4 *)(p + c * 4); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    float *q = out + (size_t)wave * D;
    for (int c = lane; c < D / 4; c += 64) *(float4 *)(q + c * 4) = acc;
}

int main() {
    const int D = 400, NR = 2000;
    float *A, *o, *T, *out; long *idx;
    CK(hipMalloc(&A, (size_t)NR * D * 4)); CK(hipMalloc(&o, 1 << 22)); CK(hipMemset(A, 0, (size_t)NR * D * 4));
    const int NENT = 14951, NW = 4000;
    CK(hipMalloc(&T, (size_t)NENT * D * 4)); CK(hipMemset(T, 0, (size_t)NENT * D * 4));
    CK(hipMalloc(&out, (size_t)NW * D * 4)); CK(hipMalloc(&idx, (size_t)NW * 4 * 8));
    std::vector<long> h(NW * 4); for (size_t i = 0; i < h.size(); ++i) h[i] = (i * 7919 + 13) % NENT;
    CK(hipMemcpy(idx, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int REP = 100;
    auto run = [&](const char *name, auto launch) -> int {
        for (int r = 0; r < 10; ++r) launch();
        CK(hipStreamSynchronize(s));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int r = 0; r < REP; ++r) launch();
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.3f us\n", name, 1e3 * ms / REP);
        return 0;
    };
    run("mfma 16x16x4 x200, 2 acc, 256 blk (1 wave/SIMD)", [&] { hipLaunchKernelGGL(k_mfma<2>, dim3(256), dim3(256), 0, s, o, 100); });
    run("mfma 16x16x4 x200, 4 acc, 256 blk", [&] { hipLaunchKernelGGL(k_mfma<4>, dim3(256), dim3(256), 0, s, o, 50); });
    run("mfma 16x16x4 x2000, 4 acc, 256 blk", [&] { hipLaunchKernelGGL(k_mfma<4>, dim3(256), dim3(256), 0, s, o, 500); });
    run("mfma x200 4acc, 212 blk (845 waves)", [&] { hipLaunchKernelGGL(k_mfma<4>, dim3(212), dim3(256), 0, s, o, 50); });
    run("ld_frag 212 blk x 4 waves (845 tiles), 2 opnds", [&] { hipLaunchKernelGGL(k_ld_frag, dim3(424), dim3(256), 0, s, A, o, D, NR); });
    run("ld_row  same bytes", [&] { hipLaunchKernelGGL(k_ld_row, dim3(424), dim3(256), 0, s, A, o, D, NR); });
    run("gather 1000 waves x 3 rows -> 1 row", [&] { hipLaunchKernelGGL(k_gather, dim3(250), dim3(256), 0, s, T, idx, out, D, 1000, 3); });
    run("gather 2000 waves x 2 rows -> 1 row", [&] { hipLaunchKernelGGL(k_gather, dim3(500), dim3(256), 0, s, T, idx, out, D, 2000, 2); });
    run("gather 4000 waves x 1 row -> 1 row", [&] { hipLaunchKernelGGL(k_gather, dim3(1000), dim3(256), 0, s, T, idx, out, D, 4000, 1); });
    return 0;
}
