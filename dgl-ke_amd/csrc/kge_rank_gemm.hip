// kge_rank_gemm.hip - ranking evaluation of the matrix-form models (KEModel.forward_test, models/general_models.py:436-485;
// candidates = every entity, dataloader/sampler.py:514-597) as ONE LDS-tiled fp32-MFMA GEMM per batch of test triples whose
// epilogue keeps a single BIT per (test triple, candidate): score >= the true triple's score.
//
// Round 6.  The evaluation used the training step's forward tiles (kge_neg_gemm.hip: one wavefront per 16 x 16 tile, operands
// straight from L2 in fragment layout - built for a 200 x 200 x 400 product that is latency) on a 1024 x 14 951 x 400 product
// and wrote the [rows, candidates] score block for a counting kernel: 241 us per batch = 51 TFLOP/s = 32 % of the fp32-MFMA
// peak, 92 % of a validation's device time (tools/eval_timing.py).  A product of this size is a throughput problem:
//   * workgroup tile 128 test rows x 128 candidates, four wavefronts of 64 x 64 (4 x 4 MFMA tiles: one fragment read feeds four
//     MFMAs), k in stages of 32 through LDS (row stride 36 dwords = 4 x odd: the fragment reads - 16 rows x one float4 - are
//     conflict-free), 128-byte global segments, the next stage's global loads in flight under the stage's 128 MFMAs per wavefront;
//   * epilogue: the score exactly as the forward tiles form it (TransE_l2: gamma - sqrt(max(|a|^2 + |b|^2 - 2 a.b, 1e-30)) with the
//     precomputed norms; SimplE: clamp), compared with the row's positive score, four ballots per 16 x 64 strip -> one 64-bit
//     word per row and wavefront: 1.9 MB of mask per batch instead of 61 MB of scores written and read back;
//   * rank_mask_kernel: rank = 1 + popcount(row) - bits at the row's filtered columns - the SAME comparison result serves both
//     terms, so the true triple's own column (always in the filter list) cancels whatever its rounding.
// The pairwise models (TransE_l1, RotatE) and TransR keep the score block + rank_count_kernel (kge_eval.hip).
#include "kge_common.hpp"

using namespace kge;

#define RG_BM 128
#define RG_BN 128
#define RG_BK 32                                  // (the stage loop below is written for two 16-k halves)
#define RG_LD (RG_BK + 4)                         // dwords per staged row
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct RankGemmArgs {
    const float *A; int rows;                     // [rows, D] pos-side vectors of this batch (dense)
    const float *nbase; const int64_t *nidx;      // candidate j: nbase + (nidx ? nidx[j] : j) * D
    int64_t N; int D;
    int l2; float gamma, clampv;
    const float *asq, *bsq, *P;                   // [rows], [N] squared norms (TransE_l2); [rows] positive scores
    unsigned long long *mask; int64_t words;      // out [rows, words] bits: score >= P
};

static inline int check_launch_g() { return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH; }

__global__ __launch_bounds__(256) void rank_gemm_kernel(RankGemmArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2][(RG_BM + RG_BN) * RG_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbn = (int)((a.N + RG_BN - 1) / RG_BN);
    // consecutive workgroups walk the candidates of one row block: the 128 pos-side rows stay in L2 while the table streams
    const int bm = (int)blockIdx.x / nbn, bn = (int)blockIdx.x % nbn;
    const int D = a.D;
    const int nst = (D + RG_BK - 1) / RG_BK;
    const int c4t = (tid & 7) * 4;
    // ---- staging: float4 f = tid + 256 i of the stage's 2 x 1024: row f >> 3 (< 128: pos-side, else candidate), piece f & 7 ------
    const float *gp[8];
    int lo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = tid + 256 * i, row = f >> 3, c4 = (f & 7) * 4;
        if (row < RG_BM) gp[i] = a.A + (int64_t)min(bm * RG_BM + row, a.rows - 1) * D + c4;
        else gp[i] = row_ptr(a.nbase, a.nidx, min((int64_t)bn * RG_BN + row - RG_BM, a.N - 1), D) + c4;
        lo[i] = row * RG_LD + c4;
    }
    f32x4 g[8];
    // (D % 4 == 0: this thread's float4 of a stage is whole or beyond the row.  Only the LAST stage can reach beyond: its loads are
    //  clamped in-bounds re-reads like every prefetch here - unconditional, no exec-masked branch around a load - and stored as zeros)
    const bool tail = (D % RG_BK) != 0 && (nst - 1) * RG_BK + c4t >= D;
    auto gload = [&](int s) {
        const int off = min(s * RG_BK, D - 4 - c4t);
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = *reinterpret_cast<const f32x4 *>(gp[i] + off);
    };
    auto lstore = [&](int bf, bool last) {
        if (last && (D % RG_BK) != 0) {                            // (wave-uniform)
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = tail ? (f32x4){0.f, 0.f, 0.f, 0.f} : g[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4 *>(&lds[bf][lo[i]]) = g[i];
    };
    // ---- compute: wavefront (wr, wc) owns rows [64 wr, +64) x candidates [64 wc, +64) of the tile ---------------------------
    const int wr = wave >> 1, wc = wave & 1;
    const int m = lane & 15, q = lane >> 4;
    const int aoff = (wr * 64 + m) * RG_LD + 4 * q, boff = (RG_BM + wc * 64 + m) * RG_LD + 4 * q;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gload(0);
    lstore(0, nst == 1);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        const int bf = s & 1;
        if (s + 1 < nst) gload(s + 1);
        // both 16-k halves' fragments are requested up front; the next stage goes to the other LDS buffer BETWEEN the two halves'
        // MFMAs (its global loads were issued a half-stage = 64 MFMAs ago), so that the stores sit under the second half's MFMAs
        f32x4 af[2][4], bfr[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[kb][i] = *reinterpret_cast<const f32x4 *>(&lds[bf][aoff + i * 16 * RG_LD + kb * 16]);
                bfr[kb][i] = *reinterpret_cast<const f32x4 *>(&lds[bf][boff + i * 16 * RG_LD + kb * 16]);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = MFMA16(af[0][i][e], bfr[0][j][e], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nst) lstore(bf ^ 1, s + 2 == nst);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = MFMA16(af[1][i][e], bfr[1][j][e], acc[i][j]);
        __syncthreads();
    }
    // ---- epilogue: one bit per (row, candidate) ---------------------------------------------------------------------------------
    const int64_t j0 = (int64_t)bn * RG_BN + wc * 64;
    float bs[4];
    bool jok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t col = j0 + 16 * j + m;
        jok[j] = col < a.N;
        bs[j] = (a.l2 && jok[j]) ? a.bsq[col] : 0.f;
    }
    const int myrow = bm * RG_BM + wr * 64 + lane;                 // the row whose word this lane writes
    unsigned long long word = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned long long bal[4][4];                               // [r][j]: bit 16 q' + m' = (row 16 i + 4 q' + r, column 16 j + m')
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = min(bm * RG_BM + wr * 64 + 16 * i + 4 * q + r, a.rows - 1);
            const float p = a.P[row];
            const float as = a.l2 ? a.asq[row] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = acc[i][j][r];
                if (a.l2) x = a.gamma - sqrtf(fmaxf(fmaf(-2.f, x, as + bs[j]), 1e-30f));
                else if (a.clampv > 0.f) x = fminf(fmaxf(x, -a.clampv), a.clampv);
                bal[r][j] = __ballot(jok[j] && x >= p);
            }
        }
        if ((lane >> 4) == i) {                                     // lanes 16 i .. 16 i + 15 own rows 16 i + (lane & 15) of the strip
            const int qq = (lane & 15) >> 2, rr = lane & 3;
            unsigned long long w = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r == rr) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) w |= ((bal[r][j] >> (16 * qq)) & 0xffffull) << (16 * j);
                }
            word = w;
        }
    }
    if (myrow < a.rows) a.mask[(int64_t)myrow * a.words + (j0 >> 6)] = word;     // (columns beyond N: zero bits)
}

// rank_i = 1 + #{j : bit(i, j)} - #{j in filt_i : bit(i, j)};  filt_i = filt_ids[filt_ptr[2 (e0 + i)] .. filt_ptr[2 (e0 + i) + 1])
__global__ __launch_bounds__(KGE_BLOCK) void rank_mask_kernel(const unsigned long long *__restrict__ mask, int64_t words, int64_t N,
                                                             int rows, const int64_t *__restrict__ filt_ptr,
                                                             const int64_t *__restrict__ filt_ids, int64_t e0,
                                                             int32_t *__restrict__ ranks) {
    const int i = (int)blockIdx.x * KGE_WAVES_PER_BLOCK + (int)(threadIdx.x >> 6);      // one wavefront per test triple
    if (i >= rows) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long *row = mask + (int64_t)i * words;
    int cnt = 0;
    for (int64_t w = lane; w < words; w += 64) cnt += __popcll(row[w]);
    if (filt_ptr) {
        const int64_t f0 = filt_ptr[2 * (e0 + i)], f1 = filt_ptr[2 * (e0 + i) + 1];
        for (int64_t k = f0 + lane; k < f1; k += 64) {
            const int64_t col = filt_ids[k];
            if (col >= 0 && col < N) cnt -= (int)((row[col >> 6] >> (col & 63)) & 1ull);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) ranks[e0 + i] = 1 + cnt;
}

bool rank_gemm_supported(int model, int d_e) {
    return (model == KGE_TRANSE_L2 || model == KGE_DISTMULT || model == KGE_COMPLEX || model == KGE_SIMPLE || model == KGE_RESCAL) &&
           d_e >= 4 && d_e % 4 == 0;
}
size_t rank_gemm_mask_bytes(int rows, int64_t N) { return (size_t)rows * (size_t)((N + 127) / 128 * 2) * 8; }

// A, asq (TransE_l2), P: this batch's pos-side vectors / norms / positive scores (edge_fwd); bsq: the candidates' norms
int launch_rank_gemm(int model, const float *A, int rows, const float *nbase, const int64_t *nidx, int64_t N, int D, float gamma,
                     float clampv, const float *asq, const float *bsq, const float *P, void *mask, const int64_t *filt_ptr,
                     const int64_t *filt_ids, int64_t e0, int32_t *ranks, hipStream_t s) {
    if (rows <= 0) return KGE_OK;
    RankGemmArgs a{};
    a.A = A; a.rows = rows; a.nbase = nbase; a.nidx = nidx; a.N = N; a.D = D;
    a.l2 = model == KGE_TRANSE_L2 ? 1 : 0; a.gamma = gamma; a.clampv = clampv;
    a.asq = asq; a.bsq = bsq; a.P = P;
    a.mask = reinterpret_cast<unsigned long long *>(mask);
    a.words = (N + 127) / 128 * 2;                                // whole 128-candidate tiles: every word of a row is written
    const int64_t nb = (int64_t)((rows + RG_BM - 1) / RG_BM) * ((N + RG_BN - 1) / RG_BN);
    hipLaunchKernelGGL(rank_gemm_kernel, dim3((unsigned)nb), dim3(256), 0, s, a);
    if (int rc = check_launch_g()) return rc;
    hipLaunchKernelGGL(rank_mask_kernel, dim3((unsigned)((rows + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK)), dim3(KGE_BLOCK), 0, s,
                       a.mask, a.words, N, rows, filt_ptr, filt_ids, e0, ranks);
    return check_launch_g();
}
