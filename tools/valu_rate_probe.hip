// valu_rate_probe.hip - issue cost of the VALU instruction kinds the pairwise kernels (kge_neg_bcast.hip: TransE_l1, RotatE) are
// made of, on MI355X: plain fp32, packed fp32 (v_pk_*), the quarter-rate unit (v_sqrt / v_rsq / v_exp) and the instruction MIX of
// the RotatE forward / backward inner loops.  Every wavefront runs REPS x 64 independent instructions of one kind (8 dependency
// chains) between two s_memtime reads; 1 / 2 / 4 wavefronts per SIMD (one workgroup per CU, 256 CUs).
// Output: shader cycles per wavefront-instruction as seen by ONE wavefront, and per SIMD (= the former / wavefronts per SIMD):
// the second is the issue cost that bounds a VALU-bound kernel.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o tools/valu_rate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum { K_FMA, K_ADDABS, K_PKFMA, K_PKADD, K_PKMUL, K_SQRT, K_RSQ, K_EXP, K_ROT_FWD, K_ROT_FWD_SCALAR, K_ROT_BWD, K_ROT_FWD_LDS, K_L1_FWD, K_N };
static const char *KNAME[K_N] = {"v_fma_f32", "v_add_f32 |abs|", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_sqrt_f32", "v_rsq_f32",
                                 "v_exp_f32", "RotatE fwd mix (per 2 complex: 3 pk_add 1 pk_mul 1 pk_fma 2 sqrt)",
                                 "RotatE fwd, unpacked (per complex: 2 sub 1 mul 1 fma 1 sqrt 1 add)",
                                 "RotatE bwd mix (per 2 complex: 2 pk_add 1 pk_mul 6 pk_fma 1 mov_dpp 2 rsq)",
                                 "RotatE fwd mix + 1 ds_read_b128 (broadcast) per 2 complex",
                                 "TransE_l1 fwd (per 2 elements: 1 pk_add 2 add|abs|)"};
// instructions per unrolled block (for the per-instruction figures) and "units" per block (complex elements / elements)
static const int KINSTR[K_N] = {64, 64, 64, 64, 64, 64, 64, 64, 8 * 7, 8 * 6, 8 * 12, 8 * 8, 16 * 3};
static const int KUNITS[K_N] = {64, 64, 128, 128, 128, 64, 64, 64, 16, 8, 16, 16, 32};

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(1024) void k_rate(float *o, unsigned long long *tl, int reps) {
    __shared__ v4f lds[256];
    const float s = threadIdx.x * 0.001f + 1.f;
    float x[8];
    v2f p[8], q[8], acc[8];
    for (int i = 0; i < 8; ++i) { x[i] = s + i; p[i] = v2f{s + i, s - i}; q[i] = v2f{0.5f * s + i, 0.25f * s}; acc[i] = v2f{0.f, 0.f}; }
    if (threadIdx.x < 256) lds[threadIdx.x] = v4f{s, s + 1, s + 2, s + 3};
    __syncthreads();
    const float y = 1.0001f;
    const v2f y2 = v2f{1.0001f, 0.9999f};
    const v4f *lp = lds + (blockIdx.x & 3);          // the same address in every lane: a broadcast read
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (KIND == K_FMA) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(y));
                REP8(X)
#undef X
            }
        } else if (KIND == K_ADDABS) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#define X(i) asm volatile("v_add_f32_e64 %0, %0, |%1|" : "+v"(x[i]) : "v"(y));
                REP8(X)
#undef X
            }
        } else if (KIND == K_PKFMA) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(y2));
                REP8(X)
#undef X
            }
        } else if (KIND == K_PKADD) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(y2));
                REP8(X)
#undef X
            }
        } else if (KIND == K_PKMUL) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(y2));
                REP8(X)
#undef X
            }
        } else if (KIND == K_SQRT) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
                REP8(X)
#undef X
            }
        } else if (KIND == K_RSQ) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#define X(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
                REP8(X)
#undef X
            }
        } else if (KIND == K_EXP) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                REP8(X)
#undef X
            }
        } else if (KIND == K_ROT_FWD || KIND == K_ROT_FWD_LDS) {
            // per chain and pass TWO complex elements, the stages of the eight chains interleaved (as the compiler schedules the
            // real loop): d0 = a0 - b0, d1 = a1 - b1 (2 pk_add), n = d.re^2 (pk_mul), n += d.im^2 (pk_fma), 2 sqrt, acc += n (pk_add).
            // Register roles only - the values are irrelevant.
            v2f d0[8], d1[8], n[8];
            if (KIND == K_ROT_FWD_LDS) {
#define X(i) { v4f t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)lp)); p[i] = v2f{t.x, t.y}; q[i] = v2f{t.z, t.w}; }
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#define X(i) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d0[i]) : "v"(p[i]), "v"(q[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d1[i]) : "v"(q[i]), "v"(y2));
            REP8(X)
#undef X
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(n[i]) : "v"(d0[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(n[i]) : "v"(d1[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(n[i].x));
            REP8(X)
#undef X
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(n[i].y));
            REP8(X)
#undef X
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(n[i]));
            REP8(X)
#undef X
        } else if (KIND == K_ROT_FWD_SCALAR) {
            float dr[8], di[8], n[8];
#define X(i) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dr[i]) : "v"(p[i].x), "v"(q[i].x));
            REP8(X)
#undef X
#define X(i) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(di[i]) : "v"(p[i].y), "v"(q[i].y));
            REP8(X)
#undef X
#define X(i) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(n[i]) : "v"(dr[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(n[i]) : "v"(di[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(n[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(n[i]));
            REP8(X)
#undef X
        } else if (KIND == K_ROT_BWD) {
            // the instruction counts of neg_bwd_lc_kernel<RotatE>'s inner loop per complex element (one per lane: re, im packed):
            // 1 pk_add, 3 pk_fma, 1/2 pk_mul, 1/2 v_mov dpp, 1 rsq - here per chain and pass TWO complex elements
            v2f d0[8], d1[8], n0[8], n1[8];
            float w[8];
#define X(i) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d0[i]) : "v"(p[i]), "v"(q[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d1[i]) : "v"(q[i]), "v"(y2));
            REP8(X)
#undef X
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(n0[i]) : "v"(d0[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(n1[i]) : "v"(d1[i]), "v"(n0[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(w[i]) : "v"(x[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(n1[i].x));
            REP8(X)
#undef X
#define X(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(n1[i].y));
            REP8(X)
#undef X
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(d0[i]), "v"(n1[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[i]) : "v"(d1[i]), "v"(n1[i]));
            REP8(X)
#undef X
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(n0[i]) : "v"(d0[i]), "v"(y2)); x[i] = w[i];
            REP8(X)
#undef X
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(n0[i]), "v"(y2));
            REP8(X)
#undef X
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q[i]) : "v"(n1[i]), "v"(y2));
            REP8(X)
#undef X
        } else if (KIND == K_L1_FWD) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                v2f d[8];
#define X(i) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d[i]) : "v"(p[i]), "v"(q[i]));
                REP8(X)
#undef X
#define X(i) asm volatile("v_add_f32_e64 %0, %0, |%1|" : "+v"(x[i]) : "v"(d[i].x));
                REP8(X)
#undef X
#define X(i) asm volatile("v_add_f32_e64 %0, %0, |%1|" : "+v"(x[i]) : "v"(d[i].y));
                REP8(X)
#undef X
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float r_ = 0.f;
    for (int i = 0; i < 8; ++i) r_ += x[i] + p[i].x + p[i].y + q[i].x + q[i].y + acc[i].x + acc[i].y;
    o[blockIdx.x * blockDim.x + threadIdx.x] = r_;
    if ((threadIdx.x & 63) == 0) tl[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kfn)(float *, unsigned long long *, int);

int main() {
    float *o; unsigned long long *tl;
    const int blocks = 256;
    CK(hipMalloc(&o, sizeof(float) * blocks * 1024));
    CK(hipMalloc(&tl, sizeof(unsigned long long) * blocks * 16));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kfn K[K_N] = {k_rate<K_FMA>, k_rate<K_ADDABS>, k_rate<K_PKFMA>, k_rate<K_PKADD>, k_rate<K_PKMUL>, k_rate<K_SQRT>, k_rate<K_RSQ>, k_rate<K_EXP>,
                  k_rate<K_ROT_FWD>, k_rate<K_ROT_FWD_SCALAR>, k_rate<K_ROT_BWD>, k_rate<K_ROT_FWD_LDS>, k_rate<K_L1_FWD>};
    const int reps = 400;
    printf("valu_rate_probe: %d workgroups (one per CU), %d repetitions of the unrolled block per wavefront; cycles = s_memtime shader cycles\n", blocks, reps);
    printf("%-96s %5s %12s %12s %12s %10s\n", "instruction kind", "w/SIMD", "cyc/instr/wave", "cyc/instr/SIMD", "cyc/unit/SIMD", "kernel us");
    for (int k = 0; k < K_N; ++k) {
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 256 * wps;
            const int waves = blocks * threads / 64;
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(K[k], dim3(blocks), dim3(threads), 0, s, o, tl, reps);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(K[k], dim3(blocks), dim3(threads), 0, s, o, tl, reps);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(waves);
            CK(hipMemcpy(h.data(), tl, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double med = (double)h[waves / 2];
            const double per_instr = med / ((double)reps * KINSTR[k]);
            printf("%-96s %5d %12.2f %12.2f %12.2f %10.1f\n", KNAME[k], wps, per_instr, per_instr / wps, med / ((double)reps * KUNITS[k]) / wps, ms * 1e3);
        }
    }
    printf("units: element pairs for the packed kinds, complex elements for the RotatE mixes, elements for TransE_l1.\n");
    return 0;
}
