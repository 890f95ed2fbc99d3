// kge_neg_mfma.hip - chunked negative scoring as fp32 matrix-core GEMMs (gfx950).
//
// Reference: the create_neg closures that are a batched [chunk x D]·[D x N] product
// (models/pytorch/score_fun.py:26-34 batched_l2_dist/baddbmm for TransE_l2, :275,284 DistMult bmm,
// :359,375 ComplEx bmm) and their autograd.
//
// Per chunk c:   S   = A_c · Bn_c^T          A_c [chunk,D] pos-side vectors, Bn_c [N,D] negatives
//                GA  = W_c · Bn_c            W_c [chunk,N] = dL/dS (TransE_l2: dL/dn / dist)
//                GN  = W_c^T · A_c
// TransE_l2 wraps these in rank-1 terms:  n = gamma - sqrt(max(|a|^2 + |b|^2 - 2 S, 1e-30)),
//                GA_i = -a_i * rowsum_i(W) + (W·Bn)_i ,   GN_j = (W^T·A)_j - b_j * colsum_j(W).
//
// The problem is small (cfg: 5 chunks of 200x200x400) and LATENCY-bound, not bandwidth- or
// flop-bound: the whole operand set (3.2 MB) sits in L2/Infinity Cache and the matrix work is
// ~1 us of the chip.  So the decomposition favours MANY independent wavefronts over big tiles -
// one wavefront per 16x16 (forward) or 16x64 (backward) output tile, at most ~1 wavefront per SIMD -
// and each wavefront hides memory latency by itself: operands are streamed straight from L2 into
// registers in the MFMA operand layout with 16-byte loads (no LDS round trip), software-pipelined
// in register double buffers so that 16-20 loads are in flight while the previous group's
// v_mfma_f32_16x16x4_f32 instructions issue (exact fp32 FMA chain, so scores match an fp32
// reference to rounding).  Fragment layout of the 16x16x4 f32 MFMA (wave64):
//   A operand: lane l holds A[m = l&15][k = l>>4];  B operand: lane l holds B[k = l>>4][n = l&15]
//   C/D: lane l, reg r holds D[m = 4*(l>>4) + r][n = l&15].
// A lane loads 4 consecutive k (one float4) and feeds element e to MFMA step e; as long as A and B
// use the same lane->k assignment the sum over k is complete.
// Workgroups are remapped so that the blocks that land on one XCD (block b -> XCD b%8) work on
// consecutive tiles, i.e. on the same chunk's operands, which then stay in that XCD's L2.
#include "kge_common.hpp"

using namespace kge;

static inline int check_launch_m() {
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

bool neg_mfma_supported(int model, int d_e, int N) {
    (void)N;
    if (model != KGE_TRANSE_L2 && model != KGE_DISTMULT && model != KGE_COMPLEX) return false;
    return d_e % 4 == 0;   // 16-byte aligned rows
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// bijective XCD-aware remap: hardware block b runs on XCD b%8; give the blocks of one XCD
// consecutive logical ids.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

__device__ __forceinline__ float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ------------------------------------------------------------------------------------------
// forward: one wavefront per 16x16 tile of S
// ------------------------------------------------------------------------------------------
#ifndef FU
#define FU 8    // k-steps (of 16) per register buffer
#endif

template <bool L2>
__global__ __launch_bounds__(KGE_BLOCK) void neg_fwd_mfma_kernel(NegArgs a, int ti, int tj) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t ntiles = (int64_t)a.C * ti * tj;
    if (tile >= ntiles) return;
    const int jt = (int)(tile % tj);
    const int it = (int)((tile / tj) % ti);
    const int c = (int)(tile / ((int64_t)tj * ti));
    const int D = a.d_e;
    const int m = lane & 15, q = lane >> 4;
    // operand rows of this lane (clamped so that loads stay in bounds; masked at the store)
    const int ia = min(it * 16 + m, a.chunk - 1);
    const int jb = min(jt * 16 + m, a.N - 1);
    const float *Ap = a.A + ((int64_t)c * a.chunk + ia) * D + q * 4;
    const float *Bp = a.nbase + ((int64_t)c * a.N + jb) * D + q * 4;
    const int kq = q * 4;
    const int ksteps = (D + 15) / 16;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float4 a0[FU], b0[FU], a1[FU], b1[FU];

#define FWD_LOAD(AV, BV, KS0)                                                    \
    _Pragma("unroll") for (int u = 0; u < FU; ++u) {                             \
        const int k = ((KS0) + u) * 16;                                          \
        if (k + kq < D) { AV[u] = ldg4(Ap + k); BV[u] = ldg4(Bp + k); }          \
        else { AV[u] = zero4(); BV[u] = zero4(); }                               \
    }
#define FWD_MMA(AV, BV, KS0)                                                     \
    _Pragma("unroll") for (int u = 0; u < FU; ++u) {                             \
        if ((KS0) + u < ksteps) {                                                \
            acc0 = MFMA16(AV[u].x, BV[u].x, acc0);                               \
            acc1 = MFMA16(AV[u].y, BV[u].y, acc1);                               \
            acc0 = MFMA16(AV[u].z, BV[u].z, acc0);                               \
            acc1 = MFMA16(AV[u].w, BV[u].w, acc1);                               \
        }                                                                        \
    }

    FWD_LOAD(a0, b0, 0);
    for (int g = 0; g < ksteps; g += 2 * FU) {
        FWD_LOAD(a1, b1, g + FU);
        FWD_MMA(a0, b0, g);
        FWD_LOAD(a0, b0, g + 2 * FU);
        FWD_MMA(a1, b1, g + FU);
    }
#undef FWD_LOAD
#undef FWD_MMA

    const int j = jt * 16 + m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = it * 16 + q * 4 + r;
        if (i < a.chunk && j < a.N) {
            float v = acc0[r] + acc1[r];
            if (L2) {
                const float sq = a.asq[(int64_t)c * a.chunk + i] + a.bsq[(int64_t)c * a.N + j] - 2.f * v;
                v = a.gamma - sqrtf(fmaxf(sq, 1e-30f));
            }
            a.S[((int64_t)c * a.chunk + i) * a.N + j] = v;
        }
    }
}

int launch_neg_fwd_mfma(const NegArgs &a, hipStream_t s) {
    if (a.nidx) return KGE_ERR_ARG;   // the GEMM kernels want dense negative rows
    const int ti = (a.chunk + 15) / 16, tj = (a.N + 15) / 16;
    const int64_t ntiles = (int64_t)a.C * ti * tj;
    if (ntiles == 0) return KGE_OK;
    const int nb = (int)((ntiles + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK);
    if (a.model == KGE_TRANSE_L2)
        hipLaunchKernelGGL(neg_fwd_mfma_kernel<true>, dim3(nb), dim3(KGE_BLOCK), 0, s, a, ti, tj);
    else
        hipLaunchKernelGGL(neg_fwd_mfma_kernel<false>, dim3(nb), dim3(KGE_BLOCK), 0, s, a, ti, tj);
    return check_launch_m();
}

// ------------------------------------------------------------------------------------------
// backward: one wavefront per 16 x 64 tile of GA (rows = positives) or GN (rows = negatives).
// The B operand is loaded as one float4 along d per lane, element s feeding accumulator s, so
// accumulator s holds output columns d0 + 4*n + s and the epilogue stores one float4 per row.
// One "macro step" = 16 values of the reduction index = 4 MFMA k-steps = 16 MFMAs.
// ------------------------------------------------------------------------------------------
#ifndef BU
#define BU 4    // macro steps per register buffer
#endif

struct BwdStage { float w[4]; float4 r[4]; };

template <bool L2>
__global__ __launch_bounds__(KGE_BLOCK) void neg_bwd_mfma_kernel(NegArgs a, int ti, int tj, int td) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t nGA = (int64_t)a.C * ti * td, nGN = (int64_t)a.C * tj * td;
    if (tile >= nGA + nGN) return;
    // interleave the two products chunk by chunk so that one XCD works on one chunk's operands
    const int64_t per_chunk = (int64_t)(ti + tj) * td;
    const int c = (int)(tile / per_chunk);
    const int64_t tc = tile % per_chunk;
    const bool isGA = tc < (int64_t)ti * td;
    const int64_t tl = isGA ? tc : tc - (int64_t)ti * td;
    const int dt = (int)(tl % td);
    const int rt = (int)(tl / td);
    const int D = a.d_e, N = a.N, chunk = a.chunk;
    const int m = lane & 15, q = lane >> 4;
    const int d = dt * 64 + m * 4;              // this lane's 4 output columns
    const bool dok = d < D;                     // D % 4 == 0, so the float4 is all-or-nothing
    const int dc = dok ? d : 0;
    const float *Wc = a.W + (int64_t)c * chunk * N;
    const float *Ac = a.A + (int64_t)c * chunk * D;
    const float *Bc = a.nbase + (int64_t)c * N * D;
    f32x4 acc[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) acc[s_] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f;                           // partial row (GA) / column (GN) sum of W
    BwdStage s0[BU], s1[BU];

    // K = reduction length, rows of the streamed operand (Bn for GA, A for GN)
    const int K = isGA ? N : chunk;
    const int msteps = (K + 15) / 16;
    const float *Xc = isGA ? Bc : Ac;           // streamed operand rows [K, D]
    // W access: GA: W[i = row][k], contiguous in k ; GN: W[k][j = row], strided by N
    const int row = rt * 16 + m;
    const int R = isGA ? chunk : N;
    const bool rok = row < R;
    const int rowc = min(row, R - 1);
    const bool vecW = isGA && (N % 4 == 0);
    const float *Wrow = isGA ? Wc + (int64_t)rowc * N : Wc + rowc;
    const int64_t wstride = isGA ? 1 : N;

#define BWD_LOAD(ST, MS0)                                                                      \
    _Pragma("unroll") for (int u = 0; u < BU; ++u) {                                           \
        const int kk = ((MS0) + u) * 16 + q * 4;                                               \
        if (vecW && kk + 3 < K) {                                                              \
            const float4 t4 = ldg4(Wrow + kk);                                                 \
            ST[u].w[0] = t4.x; ST[u].w[1] = t4.y; ST[u].w[2] = t4.z; ST[u].w[3] = t4.w;        \
        } else {                                                                               \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                      \
                ST[u].w[e] = (kk + e < K) ? Wrow[(int64_t)(kk + e) * wstride] : 0.f;           \
        }                                                                                      \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                        \
            if (kk + e < K) ST[u].r[e] = ldg4(Xc + (int64_t)(kk + e) * D + dc);                \
            else ST[u].r[e] = zero4();                                                         \
        }                                                                                      \
    }
#define BWD_MMA(ST, MS0)                                                                       \
    _Pragma("unroll") for (int u = 0; u < BU; ++u) {                                           \
        if ((MS0) + u < msteps) {                                                              \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                    \
                const float wgt = rok ? ST[u].w[e] : 0.f;                                      \
                wsum += wgt;                                                                   \
                acc[0] = MFMA16(wgt, ST[u].r[e].x, acc[0]);                                    \
                acc[1] = MFMA16(wgt, ST[u].r[e].y, acc[1]);                                    \
                acc[2] = MFMA16(wgt, ST[u].r[e].z, acc[2]);                                    \
                acc[3] = MFMA16(wgt, ST[u].r[e].w, acc[3]);                                    \
            }                                                                                  \
        }                                                                                      \
    }

    BWD_LOAD(s0, 0);
    for (int g = 0; g < msteps; g += 2 * BU) {
        BWD_LOAD(s1, g + BU);
        BWD_MMA(s0, g);
        BWD_LOAD(s0, g + 2 * BU);
        BWD_MMA(s1, g + BU);
    }
#undef BWD_LOAD
#undef BWD_MMA

    // lanes with equal (lane&15) hold partial sums of the same W row/column: combine the 4 groups
    wsum += __shfl_xor(wsum, 16, 64);
    wsum += __shfl_xor(wsum, 32, 64);
    const bool reg = (!isGA) && a.reg_coef > 0.f && a.reg_norm > 0;
    float *O = isGA ? a.GA : a.GN;
    const float *Self = isGA ? Ac : Bc;         // the row's own vector (rank-1 term / regulariser)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ro = rt * 16 + q * 4 + r;
        const float rs = __shfl(wsum, q * 4 + r, 64);   // W row/column sum of output row ro
        if (ro < R && dok) {
            float4 o = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
            if (L2 || reg) {
                const float4 sv = ldg4(Self + (int64_t)ro * D + d);
                if (L2) { o.x -= sv.x * rs; o.y -= sv.y * rs; o.z -= sv.z * rs; o.w -= sv.w * rs; }
                if (reg) {
                    o.x += reg_grad(sv.x, a.reg_coef, a.reg_norm);
                    o.y += reg_grad(sv.y, a.reg_coef, a.reg_norm);
                    o.z += reg_grad(sv.z, a.reg_coef, a.reg_norm);
                    o.w += reg_grad(sv.w, a.reg_coef, a.reg_norm);
                }
            }
            *reinterpret_cast<float4 *>(O + ((int64_t)c * R + ro) * D + d) = o;
        }
    }
}

int launch_neg_bwd_mfma(const NegArgs &a, hipStream_t s) {
    if (a.nidx) return KGE_ERR_ARG;
    const int ti = (a.chunk + 15) / 16, tj = (a.N + 15) / 16, td = (a.d_e + 63) / 64;
    const int64_t ntiles = (int64_t)a.C * (ti + tj) * td;
    if (ntiles == 0) return KGE_OK;
    const int nb = (int)((ntiles + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK);
    if (a.model == KGE_TRANSE_L2)
        hipLaunchKernelGGL(neg_bwd_mfma_kernel<true>, dim3(nb), dim3(KGE_BLOCK), 0, s, a, ti, tj, td);
    else
        hipLaunchKernelGGL(neg_bwd_mfma_kernel<false>, dim3(nb), dim3(KGE_BLOCK), 0, s, a, ti, tj, td);
    return check_launch_m();
}
