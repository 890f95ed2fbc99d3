// pingpong_probe.hip - how fast can kernel B read what kernel A just wrote (different CUs/XCDs)?
// Models the step's inter-kernel traffic: row-granular (1.6 KB) producer -> consumer hand-offs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// one wave per row: out[idx_out[w]] = f(in[idx_in[w]]) (+ optional second input row)
__global__ __launch_bounds__(256) void k_rows(const float *in, const long *idx_in, float *out, const long *idx_out,
                                             int D, int n, int nsrc) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n) return;
    float4 acc[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    for (int s = 0; s < nsrc; ++s) {
        const float *p = in + idx_in[(size_t)s * n + w] * (size_t)D;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = lane + 64 * k;
            if (c < D / 4) { float4 v = *(const float4 *)(p + c * 4); acc[k].x += v.x; acc[k].y += v.y; acc[k].z += v.z; acc[k].w += v.w; }
        }
    }
    float *q = out + idx_out[w] * (size_t)D;
#pragma unroll
    for (int k = 0; k < 2; ++k) { const int c = lane + 64 * k; if (c < D / 4) *(float4 *)(q + c * 4) = acc[k]; }
}

int main() {
    const int D = 400, NROW = 16000, NW = 4000;
    float *bufA, *bufB; long *idxR, *idxW, *idxSeq;
    CK(hipMalloc(&bufA, (size_t)NROW * D * 4)); CK(hipMalloc(&bufB, (size_t)NROW * D * 4));
    CK(hipMemset(bufA, 0, (size_t)NROW * D * 4)); CK(hipMemset(bufB, 0, (size_t)NROW * D * 4));
    CK(hipMalloc(&idxR, (size_t)NW * 4 * 8)); CK(hipMalloc(&idxW, (size_t)NW * 8)); CK(hipMalloc(&idxSeq, (size_t)NW * 4 * 8));
    std::vector<long> hr(NW * 4), hw(NW), hs(NW * 4);
    for (int i = 0; i < NW * 4; ++i) { hr[i] = ((long)i * 7919 + 13) % NW; hs[i] = i % NW; }   // reads hit the rows kernel A wrote
    for (int i = 0; i < NW; ++i) hw[i] = i;                                                       // A writes rows 0..NW-1 (dense)
    CK(hipMemcpy(idxR, hr.data(), hr.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(idxW, hw.data(), hw.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(idxSeq, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int REP = 100;
    auto run = [&](const char *name, auto launch) -> int {
        for (int r = 0; r < 5; ++r) launch();
        CK(hipStreamSynchronize(s));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int r = 0; r < REP; ++r) launch();
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-64s %8.3f us per iteration\n", name, 1e3 * ms / REP);
        return 0;
    };
    const int nb = NW / 4;
    // 1. read-only gather (input never rewritten): 1 src row -> 1 out row, out rows distinct from inputs
    run("gather 4000x1 row, input read-only (L2 can keep it)", [&] { hipLaunchKernelGGL(k_rows, dim3(nb), dim3(256), 0, s, bufA, idxR, bufB, idxW, D, NW, 1); });
    // 2. ping-pong: A: bufA->bufB ; B: bufB->bufA  (every kernel reads what the previous one wrote)
    run("ping-pong pair (each kernel reads the other's output), 1 src", [&] {
        hipLaunchKernelGGL(k_rows, dim3(nb), dim3(256), 0, s, bufA, idxR, bufB, idxW, D, NW, 1);
        hipLaunchKernelGGL(k_rows, dim3(nb), dim3(256), 0, s, bufB, idxR, bufA, idxW, D, NW, 1); });
    run("ping-pong pair, 3 src rows per wave", [&] {
        hipLaunchKernelGGL(k_rows, dim3(nb), dim3(256), 0, s, bufA, idxR, bufB, idxW, D, NW, 3);
        hipLaunchKernelGGL(k_rows, dim3(nb), dim3(256), 0, s, bufB, idxR, bufA, idxW, D, NW, 3); });
    run("ping-pong pair, same-wave mapping (reader wave w reads row w)", [&] {
        hipLaunchKernelGGL(k_rows, dim3(nb), dim3(256), 0, s, bufA, idxSeq, bufB, idxW, D, NW, 1);
        hipLaunchKernelGGL(k_rows, dim3(nb), dim3(256), 0, s, bufB, idxSeq, bufA, idxW, D, NW, 1); });
    run("in-place RMW of 4000 rows (read row w, write row w)", [&] { hipLaunchKernelGGL(k_rows, dim3(nb), dim3(256), 0, s, bufA, idxSeq, bufA, idxW, D, NW, 1); });
    return 0;
}
