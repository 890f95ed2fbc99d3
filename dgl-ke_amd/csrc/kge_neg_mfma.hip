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
// The problem is small (cfg: 5 chunks of 200x200x400) and latency-bound, so the decomposition
// favours MANY independent wavefronts over big tiles: one wavefront per 16x16 (forward) or 16x64
// (backward) output tile, operands streamed straight from L2 with 16-byte loads in the MFMA
// operand layout (no LDS round trip), v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, so scores
// match an fp32 reference to rounding).  Fragment layout of the 16x16x4 f32 MFMA (wave64):
//   A operand: lane l holds A[m = l&15][k = l>>4];  B operand: lane l holds B[k = l>>4][n = l&15]
//   C/D: lane l, reg r holds D[m = 4*(l>>4) + r][n = l&15].
// A lane loads 4 consecutive k (one float4) and feeds element e to MFMA step e; as long as A and B
// use the same lane->k assignment the sum over k is complete.
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

// ------------------------------------------------------------------------------------------
// forward: one wavefront per 16x16 tile of S
// ------------------------------------------------------------------------------------------
template <bool L2>
__global__ __launch_bounds__(KGE_BLOCK) void neg_fwd_mfma_kernel(NegArgs a, int ti, int tj) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
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
    const float *Ar = a.A + ((int64_t)c * a.chunk + ia) * D;
    const float *Br = row_ptr(a.nbase, a.nidx, (int64_t)c * a.N + jb, D);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int ksteps = (D + 15) / 16;
#pragma unroll 4
    for (int ks = 0; ks < ksteps; ++ks) {
        const int k = ks * 16 + q * 4;
        float4 av = make_float4(0.f, 0.f, 0.f, 0.f), bv = av;
        if (k < D) {
            av = *reinterpret_cast<const float4 *>(Ar + k);
            bv = *reinterpret_cast<const float4 *>(Br + k);
        }
        acc0 = MFMA16(av.x, bv.x, acc0);
        acc1 = MFMA16(av.y, bv.y, acc1);
        acc0 = MFMA16(av.z, bv.z, acc0);
        acc1 = MFMA16(av.w, bv.w, acc1);
    }
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
// ------------------------------------------------------------------------------------------
template <bool L2>
__global__ __launch_bounds__(KGE_BLOCK) void neg_bwd_mfma_kernel(NegArgs a, int ti, int tj, int td) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t nGA = (int64_t)a.C * ti * td, nGN = (int64_t)a.C * tj * td;
    if (tile >= nGA + nGN) return;
    const bool isGA = tile < nGA;
    const int64_t tl = isGA ? tile : tile - nGA;
    const int tr = isGA ? ti : tj;              // row tiles of this product
    const int dt = (int)(tl % td);
    const int rt = (int)((tl / td) % tr);
    const int c = (int)(tl / ((int64_t)td * tr));
    const int D = a.d_e, N = a.N, chunk = a.chunk;
    const int m = lane & 15, q = lane >> 4;
    const int d = dt * 64 + m * 4;              // this lane's 4 output columns
    const bool dok = d < D;                     // D % 4 == 0, so the float4 is all-or-nothing
    const float *Wc = a.W + (int64_t)c * chunk * N;
    f32x4 acc[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) acc[s_] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f;                           // partial row (GA) / column (GN) sum of W

    if (isGA) {
        // GA[i,d] = sum_j W[i,j] * Bn[j,d]     M = i, K = j
        const int i = rt * 16 + m;
        const bool iok = i < chunk;
        const float *Wrow = Wc + (int64_t)min(i, chunk - 1) * N;
        const bool vecW = (N % 4) == 0;
        for (int j0 = 0; j0 < N; j0 += 16) {
            const int jj = j0 + q * 4;
            float wv[4];
            if (vecW && jj + 3 < N) {
                const float4 t4 = *reinterpret_cast<const float4 *>(Wrow + jj);
                wv[0] = t4.x; wv[1] = t4.y; wv[2] = t4.z; wv[3] = t4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) wv[e] = (jj + e < N) ? Wrow[jj + e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float wgt = iok ? wv[e] : 0.f;   // also zero for jj+e >= N (loaded as 0)
                wsum += wgt;
                const int j = min(jj + e, N - 1);
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (dok) bv = *reinterpret_cast<const float4 *>(row_ptr(a.nbase, a.nidx, (int64_t)c * N + j, D) + d);
                acc[0] = MFMA16(wgt, bv.x, acc[0]);
                acc[1] = MFMA16(wgt, bv.y, acc[1]);
                acc[2] = MFMA16(wgt, bv.z, acc[2]);
                acc[3] = MFMA16(wgt, bv.w, acc[3]);
            }
        }
        // lanes with equal (lane&15) hold partial sums of the same row: combine the 4 groups
        wsum += __shfl_xor(wsum, 16, 64);
        wsum += __shfl_xor(wsum, 32, 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int io = rt * 16 + q * 4 + r;
            const float rs = __shfl(wsum, q * 4 + r, 64);   // row sum of output row io
            if (io < chunk && dok) {
                float4 o = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
                const int64_t grow = (int64_t)c * chunk + io;
                if (L2) {
                    const float4 av = *reinterpret_cast<const float4 *>(a.A + grow * D + d);
                    o.x -= av.x * rs; o.y -= av.y * rs; o.z -= av.z * rs; o.w -= av.w * rs;
                }
                *reinterpret_cast<float4 *>(a.GA + grow * D + d) = o;
            }
        }
    } else {
        // GN[j,d] = sum_i W[i,j] * A[i,d]      M = j, K = i
        const int j = rt * 16 + m;
        const bool jok = j < N;
        const int jc = min(j, N - 1);
        const float *Ac = a.A + (int64_t)c * chunk * D;
        for (int i0 = 0; i0 < chunk; i0 += 16) {
            const int ii = i0 + q * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = ii + e;
                const bool iok = i < chunk;
                const int ic = min(i, chunk - 1);
                const float wgt = (iok && jok) ? Wc[(int64_t)ic * N + jc] : 0.f;
                wsum += wgt;
                float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
                if (dok) av = *reinterpret_cast<const float4 *>(Ac + (int64_t)ic * D + d);
                acc[0] = MFMA16(wgt, av.x, acc[0]);
                acc[1] = MFMA16(wgt, av.y, acc[1]);
                acc[2] = MFMA16(wgt, av.z, acc[2]);
                acc[3] = MFMA16(wgt, av.w, acc[3]);
            }
        }
        wsum += __shfl_xor(wsum, 16, 64);
        wsum += __shfl_xor(wsum, 32, 64);
        const bool reg = a.reg_coef > 0.f && a.reg_norm > 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jo = rt * 16 + q * 4 + r;
            const float cs = __shfl(wsum, q * 4 + r, 64);
            if (jo < N && dok) {
                float4 o = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
                const int64_t grow = (int64_t)c * N + jo;
                if (L2 || reg) {
                    const float4 bv = *reinterpret_cast<const float4 *>(row_ptr(a.nbase, a.nidx, grow, D) + d);
                    if (L2) { o.x -= bv.x * cs; o.y -= bv.y * cs; o.z -= bv.z * cs; o.w -= bv.w * cs; }
                    if (reg) {
                        o.x += reg_grad(bv.x, a.reg_coef, a.reg_norm);
                        o.y += reg_grad(bv.y, a.reg_coef, a.reg_norm);
                        o.z += reg_grad(bv.z, a.reg_coef, a.reg_norm);
                        o.w += reg_grad(bv.w, a.reg_coef, a.reg_norm);
                    }
                }
                *reinterpret_cast<float4 *>(a.GN + grow * D + d) = o;
            }
        }
    }
}

int launch_neg_bwd_mfma(const NegArgs &a, hipStream_t s) {
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
