// frag_shape_probe.hip - does the SHAPE of a gathered 1-KB wavefront load set its cost?  (round 5)
// The forward tiles' B operand is gathered in the matrix core's fragment shape: one dwordx4 load per k-step = 16 rows x 64 bytes
// (16 half cache lines).  profiles/r04_prefetch_depth_and_reg3.txt reads the loop as bound by the per-CU cost of such requests.
// This probe runs the same amount of data per wavefront (16 gathered rows x 384 floats = 24 loads of 1 KB, 4 MFMAs per load,
// 260 workgroups x 4 wavefronts, rows drawn at random from a 15 k x 384 table, other rows every launch) with the 1 KB arranged as
//   FRAG  16 rows x  64 B      (today's fragment loads)
//   ROW8   8 rows x 128 B      (one full line per row)
//   ROW4   4 rows x 256 B
//   ROW1   1 row  x 1024 B     (plain coalesced)
// at 2 / 4 loads in flight, and prints the time per launch (graph of 100 launches, boundary included).
// build: hipcc --offload-arch=gfx950 -O3 tools/frag_shape_probe.hip -o tools/run/frag_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int D = 384, NL = 24;

template <int SHAPE>
__device__ __forceinline__ const float *addr(const float *tab, const int *rows, int j, int lane) {
    int r, off;
    if (SHAPE == 0) { r = lane & 15; off = j * 16 + (lane >> 4) * 4; }
    else if (SHAPE == 1) { r = 8 * (j & 1) + (lane >> 3); off = (j >> 1) * 32 + (lane & 7) * 4; }
    else if (SHAPE == 2) { r = 4 * (j & 3) + (lane >> 4); off = (j >> 2) * 64 + (lane & 15) * 4; }
    else { const int f = j * 256 + lane * 4; r = f / D; off = f % D; }
    return tab + (size_t)rows[r] * D + off;
}

template <int SHAPE, int FU>
__global__ __launch_bounds__(256) void k_probe(const float *tab, const int *ids, int id_off, float *out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // XCDAWARE (id_off < 0 encodes it): block ids go round-robin over the 8 XCDs - logical workgroup = (bid % 8) * (grid / 8) + bid / 8,
    // so that the 52 workgroups sharing a chunk's 200 rows sit on one or two XCDs (what xcd_remap does in kge_neg_gemm.hip)
    int bid = blockIdx.x;
    if (id_off < 0) { id_off = -id_off - 1; const int per = (gridDim.x + 7) / 8; bid = min((bid % 8) * per + bid / 8, (int)gridDim.x - 1); }
    const int w = bid * 4 + wv;
    __shared__ int rows_s[4][16];
    if (lane < 16) rows_s[wv][lane] = ids[id_off + w * 16 + lane];      // the id round
    __syncthreads();
    const int *rows = rows_s[wv];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float4 buf[FU];
#pragma unroll
    for (int u = 0; u < FU; ++u) buf[u] = *reinterpret_cast<const float4 *>(addr<SHAPE>(tab, rows, u, lane));
    const float av = (float)lane;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const float4 b = buf[j % FU];
        if (j + FU < NL) buf[j % FU] = *reinterpret_cast<const float4 *>(addr<SHAPE>(tab, rows, j + FU, lane));
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.w, acc1, 0, 0, 0);
    }
    float4 o;
    o.x = acc0[0] + acc1[0]; o.y = acc0[1] + acc1[1]; o.z = acc0[2] + acc1[2]; o.w = acc0[3] + acc1[3];
    *reinterpret_cast<float4 *>(out + ((size_t)w * 64 + lane) * 4) = o;
}

// ROW8 loads staged through a wavefront-private LDS slab (16 rows x 32 floats, row stride 36 floats) and read back in the fragment
// shape: what a forward tile would do to keep the matrix core's operand layout - no barrier (the slab belongs to one wavefront),
// registers hold slab s+1 while slab s is in LDS, the LDS round trip sits behind the 8 MFMAs of the slab before
template <int MODE, int RD>     // 0: staged ROW8; 1: no loads at all (id round + MFMAs + store: the floor of this launch)
__global__ __launch_bounds__(256) void k_staged(const float *tab, const int *ids, int id_off, float *out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int bid = blockIdx.x;
    if (id_off < 0) { id_off = -id_off - 1; const int per = (gridDim.x + 7) / 8; bid = min((bid % 8) * per + bid / 8, (int)gridDim.x - 1); }
    const int w = bid * 4 + wv;
    __shared__ int rows_s[4][16];
    __shared__ float slab[4][16 * 36];
    if (lane < 16) rows_s[wv][lane] = ids[id_off + w * 16 + lane];
    __syncthreads();
    const int *rows = rows_s[wv];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float av = (float)lane;
    constexpr int NS = D / 32;
    const int r8 = lane >> 3, c8 = lane & 7, m = lane & 15, q = lane >> 4;
    const float *p0 = tab + (size_t)rows[r8] * D + c8 * 4, *p1 = tab + (size_t)rows[8 + r8] * D + c8 * 4;
    float *w0 = &slab[wv][r8 * 36 + c8 * 4], *w1 = &slab[wv][(8 + r8) * 36 + c8 * 4];
    const float *rd = &slab[wv][m * 36 + q * 4];
    // LDS operations of one wavefront execute in program order: "read slab s+1, then overwrite it with slab s+2" needs no wait in
    // between; RD slabs in flight in registers (ring), fragments of slab s+1 read while the MFMAs of slab s run
    float4 ra[RD], rb[RD];
#pragma unroll
    for (int u = 0; u < RD; ++u) {
        if (MODE == 0) { ra[u] = *reinterpret_cast<const float4 *>(p0 + min(u, NS - 1) * 32); rb[u] = *reinterpret_cast<const float4 *>(p1 + min(u, NS - 1) * 32); }
        else { ra[u] = make_float4(1.f, 2.f, 3.f, 4.f); rb[u] = ra[u]; }
    }
    *reinterpret_cast<float4 *>(w0) = ra[0]; *reinterpret_cast<float4 *>(w1) = rb[0];           // slab 0 -> LDS
    if (MODE == 0) { ra[0] = *reinterpret_cast<const float4 *>(p0 + min(RD, NS - 1) * 32); rb[0] = *reinterpret_cast<const float4 *>(p1 + min(RD, NS - 1) * 32); }
    float4 f0 = *reinterpret_cast<const float4 *>(rd), f1 = *reinterpret_cast<const float4 *>(rd + 16);
#pragma unroll
    for (int sI = 0; sI < NS; ++sI) {
        float4 g0 = f0, g1 = f1;
        if (sI + 1 < NS) {
            const int u = (sI + 1) % RD;                              // the ring slot that holds slab sI + 1
            *reinterpret_cast<float4 *>(w0) = ra[u]; *reinterpret_cast<float4 *>(w1) = rb[u];
            if (MODE == 0 && sI + 1 + RD < NS) {
                ra[u] = *reinterpret_cast<const float4 *>(p0 + (sI + 1 + RD) * 32); rb[u] = *reinterpret_cast<const float4 *>(p1 + (sI + 1 + RD) * 32);
            }
            g0 = *reinterpret_cast<const float4 *>(rd); g1 = *reinterpret_cast<const float4 *>(rd + 16);
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f0.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f0.w, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f1.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f1.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f1.w, acc1, 0, 0, 0);
        f0 = g0; f1 = g1;
    }
    float4 o;
    o.x = acc0[0] + acc1[0]; o.y = acc0[1] + acc1[1]; o.z = acc0[2] + acc1[2]; o.w = acc0[3] + acc1[3];
    *reinterpret_cast<float4 *>(out + ((size_t)w * 64 + lane) * 4) = o;
}

int main() {
    const int NROW = 15000, NWG = 260, REP = 100;
    float *tab, *out; int *ids;
    CK(hipMalloc(&tab, (size_t)NROW * D * 4)); CK(hipMalloc(&out, (size_t)NWG * 4 * 64 * 16));
    const int nid = NWG * 4 * 16 * REP;
    std::vector<int> h(nid);
    srand(1);
    // like the step: a chunk's 200 negative rows are shared by 13 row strips - here every group of 13 wavefronts x 16 rows draws from
    // the same 200 rows of the launch (fresh rows every launch)
    for (int rep = 0; rep < REP; ++rep) {
        std::vector<int> pool(5 * 200);
        for (auto &p : pool) p = rand() % NROW;
        for (int w = 0; w < NWG * 4; ++w)
            for (int l = 0; l < 16; ++l) h[(size_t)rep * NWG * 64 + w * 16 + l] = pool[((w / 4) / 52) * 200 + ((w % 13) * 16 + l) % 200];
    }
    CK(hipMalloc(&ids, (size_t)nid * 4));
    CK(hipMemcpy(ids, h.data(), (size_t)nid * 4, hipMemcpyHostToDevice));
    CK(hipMemset(tab, 0, (size_t)NROW * D * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) -> int {
        for (int r = 0; r < 5; ++r) launch(r);
        CK(hipStreamSynchronize(s));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int r = 0; r < REP; ++r) launch(r);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        float best = 1e9f;
        for (int t = 0; t < 5; ++t) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-22s %7.3f us per launch (best of 5 graphs of %d)\n", name, 1e3 * best / REP, REP);
        return 0;
    };
    int xa = 0;
#define RUN(S, F, NAME) run(NAME, [&](int r) { const int o = r * NWG * 64; hipLaunchKernelGGL((k_probe<S, F>), dim3(NWG), dim3(256), 0, s, tab, ids, xa ? -o - 1 : o, out); })
    for (int pass = 0; pass < 2; ++pass) {
        xa = pass;
        printf("---- workgroups of a chunk %s ----\n", xa ? "on one or two XCDs" : "spread over the eight XCDs");
        RUN(0, 2, "FRAG 16x64B  FU=2"); RUN(1, 2, "ROW8  8x128B FU=2"); RUN(2, 2, "ROW4  4x256B FU=2"); RUN(3, 2, "ROW1  1x1KB  FU=2");
        RUN(0, 4, "FRAG 16x64B  FU=4"); RUN(1, 4, "ROW8  8x128B FU=4"); RUN(2, 4, "ROW4  4x256B FU=4"); RUN(3, 4, "ROW1  1x1KB  FU=4");
        RUN(0, 8, "FRAG 16x64B  FU=8"); RUN(2, 8, "ROW4  4x256B FU=8"); RUN(3, 8, "ROW1  1x1KB  FU=8");
#define RUNS(M, R_, NAME) run(NAME, [&](int r) { const int o = r * NWG * 64; hipLaunchKernelGGL((k_staged<M, R_>), dim3(NWG), dim3(256), 0, s, tab, ids, xa ? -o - 1 : o, out); })
        RUNS(0, 1, "ROW8 via LDS, 1 slab"); RUNS(0, 2, "ROW8 via LDS, 2 slabs"); RUNS(0, 3, "ROW8 via LDS, 3 slabs"); RUNS(1, 1, "no loads (floor)");
    }
    return 0;
}
