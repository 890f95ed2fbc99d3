// kge_eval.hip - ranking evaluation (KEModel.forward_test, models/general_models.py:436-485):
// the scores of every candidate corruption come from the SAME negative-score kernels as training
// (kge_neg_gemm.hip / kge_neg_pair.hip, one chunk = the batch of test triples, N = the candidates);
// this file only counts, per test triple, the candidates that score at least as high as the true
// triple and removes the filtered ones (neg_g.edata['bias'] == -1 in the reference).
#include "kge_common.hpp"

using namespace kge;

// one workgroup per test triple i:  rank_i = 1 + #{j : S[i,j] >= p_i} - #{j in filt_i : S[i,j] >= p_i}
// filt_i = filt_ids[filt_ptr[2i] .. filt_ptr[2i+1])  (ranges, so that triples with the same (h,r) / (r,t) share a list)
__global__ __launch_bounds__(KGE_BLOCK) void rank_count_kernel(const float *__restrict__ S, const float *__restrict__ P,
                                                               int64_t N, const int64_t *__restrict__ filt_ptr,
                                                               const int64_t *__restrict__ filt_ids, int64_t e0,
                                                               int32_t *__restrict__ ranks) {
    const int i = blockIdx.x;
    const float p = P[i];
    const float *row = S + (int64_t)i * N;
    int cnt = 0;
    for (int64_t j = threadIdx.x; j < N; j += KGE_BLOCK) cnt += row[j] >= p ? 1 : 0;
    if (filt_ptr) {
        const int64_t f0 = filt_ptr[2 * (e0 + i)], f1 = filt_ptr[2 * (e0 + i) + 1];
        for (int64_t k = f0 + threadIdx.x; k < f1; k += KGE_BLOCK) {
            const int64_t col = filt_ids[k];
            if (col >= 0 && col < N) cnt -= row[col] >= p ? 1 : 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    __shared__ int part[KGE_WAVES_PER_BLOCK];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < KGE_WAVES_PER_BLOCK; ++w) tot += part[w];
        ranks[e0 + i] = 1 + tot;
    }
}

int launch_rank_count(const float *S, const float *P, int rows, int64_t N, const int64_t *filt_ptr,
                      const int64_t *filt_ids, int64_t e0, int32_t *ranks, hipStream_t s) {
    if (rows <= 0) return KGE_OK;
    hipLaunchKernelGGL(rank_count_kernel, dim3(rows), dim3(KGE_BLOCK), 0, s, S, P, N, filt_ptr, filt_ids, e0, ranks);
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}
