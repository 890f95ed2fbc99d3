// kge_neg_gemm.hip - chunked negative scoring as fp32 matrix-core GEMMs (gfx950), with the loss
// gradient fused into the backward kernel.
//
// Reference: the create_neg closures that are a batched [chunk x D]·[D x N] product
// (models/pytorch/score_fun.py:26-34 baddbmm for TransE_l2, :275,284 DistMult bmm, :359,375 ComplEx
// bmm), LossGenerator.get_total_loss (models/pytorch/loss.py:69-98) and their autograd.
//
// Per chunk c:   S   = A_c · Bn_c^T          A_c [chunk,D] pos-side vectors, Bn_c [N,D] negatives
//                W   = dL/dS                  (row softmax of the adversarial weighting needs full rows)
//                GA  = W_c · Bn_c ,  GN = W_c^T · A_c
// TransE_l2:     n = gamma - sqrt(max(|a|^2 + |b|^2 - 2 S, 1e-30)),  W := dL/dn / dist,
//                GA_i = -a_i*rowsum_i(W) + (W·Bn)_i ,  GN_j = (W^T·A)_j - b_j*colsum_j(W).
//
// Design (MI355X).  The problem is small (cfg: 5 chunks of 200x200x400 = 0.48 GFLOP) and
// latency-bound: operands (3.2 MB) sit in L2 / Infinity Cache and the matrix work is ~3 us of the
// chip.  Measured on the hardware (profiles/r01_*): many independent wavefronts that each stream
// their operands straight from L2 into registers in the MFMA operand layout beat LDS-staged
// workgroup tiles (whose load -> barrier -> compute phases cannot overlap at <= 1 workgroup per
// CU).  So: one wavefront per 16x16 (forward) / 16x64 (backward) output tile, <= 1 wavefront per
// SIMD, register double buffering so that loads of the next group fly under the MFMAs of the
// current one, v_mfma_f32_16x16x4_f32 (fp32 in / fp32 accumulate = exact FMA chain).  There are
// only TWO grid-wide phases: the forward tiles also emit, per 16 columns, the partial max /
// sum-exp of every score row; the backward wavefronts combine those partials into the softmax
// statistics and apply the loss gradient ON THE FLY to the score fragment they load as MFMA
// operand - the separate loss kernel, its launch boundary and the W round trip disappear.
// Negative rows are gathered straight from the entity table through neg_ids (no dense copy).
// Fragment layout of the 16x16x4 f32 MFMA (wave64): A: lane l = A[m=l&15][k=l>>4],
// B: lane l = B[k=l>>4][n=l&15], C/D: lane l, reg r = D[4*(l>>4)+r][l&15].  A lane loads 4
// consecutive k (one float4) and feeds element e to MFMA step e.
#include "kge_common.hpp"

using namespace kge;
KGE_TL_DEFINE(gemm)

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

bool neg_mfma_supported(int model, int d_e, int N) {
    (void)N;
    if (model != KGE_TRANSE_L2 && model != KGE_DISTMULT && model != KGE_COMPLEX && model != KGE_SIMPLE &&
        model != KGE_RESCAL) return false;
    return d_e % 4 == 0;   // 16-byte aligned rows
}

static inline int check_launch_g() { return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH; }
__device__ __forceinline__ float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float sq4(const float4 &v) { return v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }

// =============================================================================================
// forward: one wavefront per 16x16 tile of S
// =============================================================================================
#ifndef FU
#define FU 2    // k-steps (of 16) per register buffer (small on purpose: code size)
#endif

template <bool L2, bool STATS>
__global__ __launch_bounds__(KGE_BLOCK) void neg_fwd_gemm_kernel(GemmArgs a, int ti, int tj) {
    KGE_TL(1);
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t ntiles = (int64_t)a.C * ti * tj;
    if (tile >= ntiles) return;
    const int jt = (int)(tile % tj);
    const int it = (int)((tile / tj) % ti);
    const int c = (int)(tile / ((int64_t)tj * ti));
    const int D = a.D;
    const int m = lane & 15, q = lane >> 4;
    // operand rows of this lane (clamped so that loads stay in bounds; masked at the store)
    const int ia = min(it * 16 + m, a.chunk - 1);
    const int jb = min(jt * 16 + m, a.N - 1);
    const float *Ap = a.A + ((int64_t)c * a.chunk + ia) * D + q * 4;
    const float *Bp = row_ptr(a.nbase, a.nidx, (int64_t)c * a.N + jb, D) + q * 4;
    const int kq = q * 4;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float4 a0[FU], b0[FU], a1[FU], b1[FU];
    // epilogue operands are requested NOW (they do not depend on the loop): one dependent ~1.5 us round less
    // at the end of every wavefront
    float bsq_pre = 0.f, asq_pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (L2) {
        bsq_pre = a.bsq[(int64_t)c * a.N + jb];
#pragma unroll
        for (int r = 0; r < 4; ++r) asq_pre[r] = a.asq[(int64_t)c * a.chunk + min(it * 16 + q * 4 + r, a.chunk - 1)];
    }

    // main loop: only FULL k-steps (all 64 lanes in range) - no per-lane predicates, no exec-mask
    // branches between the MFMAs; the (KS0)+u < kfull tests are wave-uniform scalar branches
    const int kfull = D >> 4;
#define FWD_LOAD(AV, BV, KS0)                                                    \
    _Pragma("unroll") for (int u = 0; u < FU; ++u) {                             \
        if ((KS0) + u < kfull) { AV[u] = ldg4(Ap + ((KS0) + u) * 16); BV[u] = ldg4(Bp + ((KS0) + u) * 16); } \
    }
#define FWD_MMA(AV, BV, KS0)                                                     \
    _Pragma("unroll") for (int u = 0; u < FU; ++u) {                             \
        if ((KS0) + u < kfull) {                                                 \
            acc0 = MFMA16(AV[u].x, BV[u].x, acc0);                               \
            acc1 = MFMA16(AV[u].y, BV[u].y, acc1);                               \
            acc0 = MFMA16(AV[u].z, BV[u].z, acc0);                               \
            acc1 = MFMA16(AV[u].w, BV[u].w, acc1);                               \
        }                                                                        \
    }

    FWD_LOAD(a0, b0, 0);
    for (int g = 0; g < kfull; g += 2 * FU) {
        FWD_LOAD(a1, b1, g + FU);
        FWD_MMA(a0, b0, g);
        FWD_LOAD(a0, b0, g + 2 * FU);
        FWD_MMA(a1, b1, g + FU);
    }
#undef FWD_LOAD
#undef FWD_MMA
    if (D & 15) {   // tail k-step: lanes whose 4 floats lie beyond D contribute zeros
        float4 av = zero4(), bv = zero4();
        if (kfull * 16 + kq < D) { av = ldg4(Ap + kfull * 16); bv = ldg4(Bp + kfull * 16); }
        acc0 = MFMA16(av.x, bv.x, acc0);
        acc1 = MFMA16(av.y, bv.y, acc1);
        acc0 = MFMA16(av.z, bv.z, acc0);
        acc1 = MFMA16(av.w, bv.w, acc1);
    }

    // NOTE: the MFMA stream above is kept free of VALU work on purpose - accumulating |a|^2,|b|^2
    // from the loaded fragments inside the loop cost +3.5 us (45 %) on MI355X; they come precomputed.
    const int j = jt * 16 + m;
    const bool jok = j < a.N;
    const float bsq = bsq_pre;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = it * 16 + q * 4 + r;
        float v = acc0[r] + acc1[r];
        if (L2) {
            const float ar = asq_pre[r];
            v = a.gamma - sqrtf(fmaxf(ar + bsq - 2.f * v, 1e-30f));
        } else if (a.clampv > 0.f) {
            v = fminf(fmaxf(v, -a.clampv), a.clampv);            // SimplE: th.clamp(tmp, -20, 20)
        }
        const bool ok = jok && i < a.chunk;
        if (ok) a.S[((int64_t)c * a.chunk + i) * a.N + j] = v;
        if (STATS) {
            // max and sum-exp of T*n over this tile's 16 columns (the 16 lanes with the same q)
            const float tn = ok ? v * a.adv_temp : -INFINITY;
            float mx = tn;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            float e = ok ? __expf(tn - mx) : 0.f;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
            if (m == 0 && i < a.chunk) {
                const int64_t o = ((int64_t)c * a.chunk + i) * tj + jt;
                a.PM[o] = mx;
                a.PS[o] = e;
            }
        }
    }
}

int launch_neg_fwd_gemm(const GemmArgs &a, hipStream_t s) {
    const int ti = (a.chunk + 15) / 16, tj = (a.N + 15) / 16;
    const int64_t ntiles = (int64_t)a.C * ti * tj;
    if (ntiles == 0) return KGE_OK;
    const int nb = (int)((ntiles + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK);
    const bool l2 = a.model == KGE_TRANSE_L2, st = a.PM != nullptr;
    const dim3 g(nb), b(KGE_BLOCK);
    if (l2 && st) hipLaunchKernelGGL((neg_fwd_gemm_kernel<true, true>), g, b, 0, s, a, ti, tj);
    else if (l2) hipLaunchKernelGGL((neg_fwd_gemm_kernel<true, false>), g, b, 0, s, a, ti, tj);
    else if (st) hipLaunchKernelGGL((neg_fwd_gemm_kernel<false, true>), g, b, 0, s, a, ti, tj);
    else hipLaunchKernelGGL((neg_fwd_gemm_kernel<false, false>), g, b, 0, s, a, ti, tj);
    return check_launch_g();
}

// =============================================================================================
// backward: one wavefront per 16 x 64 tile of GA (rows = positives) or GN (rows = negatives).
// A workgroup holds 4 tiles of the same (chunk, product) so that it can share, through LDS, the
// operand row pointers (GA: negative rows gathered by id) and the per-row softmax statistics
// (GN: every positive row of the chunk).  The B operand is one float4 along d per lane, element s
// feeding accumulator s (accumulator s = output columns d0 + 4*n + s -> float4 stores).
// One "macro step" = 16 values of the reduction index = 4 MFMA k-steps = 16 MFMAs.
// =============================================================================================
#ifndef BU
#define BU 1    // macro steps per register buffer (small on purpose: code size)
#endif
#define GB_MAXK 2048                   // rows of a chunk operand whose indices / statistics fit in LDS (32 KB)

struct BwdStage { float w[4]; float4 r[4]; };

// fast logistic: 1/(1+e^-x) with the hardware exp (|rel err| ~1e-6)
__device__ __forceinline__ float fsigmoid(float x) { return __frcp_rn(1.f + __expf(-x)); }

// d criterion / d score only (fast-math version of kge::criterion)
__device__ __forceinline__ float crit_grad(int genre, float s, float label, float margin) {
    if (genre == KGE_LOSS_HINGE) return (margin - label * s) < 0.f ? 0.f : -label;
    if (genre == KGE_LOSS_BCE) return fsigmoid(s) - label;
    return -label * fsigmoid(-label * s);
}

// d loss / d n_ij of a pointwise loss (loss.py:82-94) given the row's softmax statistics:
//   M = max_j T*n_ij,  coef = w_i / (2B) * (adv ? 1/sum_j exp(T*n_ij - M) : 1/N)
template <bool L2>
__device__ __forceinline__ float wgrad(const LossParams &lp, float gamma, float n, float M, float coef) {
    const float label = lp.genre == KGE_LOSS_BCE ? 0.f : -1.f;
    float g = crit_grad(lp.genre, n, label, lp.margin) * coef;
    if (lp.adv) g *= __expf(n * lp.adv_temp - M);            // detached softmax weight, loss.py:88
    if (L2) { const float d = gamma - n; g = d > 1e-15f ? g * __frcp_rn(d) : 0.f; }
    return g;
}

// combine the per-16-column partials of one score row.  All partial loads are issued before the
// first use (independent registers) so that they overlap instead of forming a latency chain.
#define GB_TJ 16
__device__ __forceinline__ void row_stats(const GemmArgs &a, int64_t gi, int tj, float &M, float &coef) {
    const float w = a.w ? a.w[gi] : 1.f;
    const float base = w * 0.5f / (float)a.B;
    if (!a.lp.adv) { M = 0.f; coef = base / (float)a.N; return; }
    const float *pm = a.PM + gi * tj, *ps = a.PS + gi * tj;
    float mx = -INFINITY, z = 0.f;
    for (int k0 = 0; k0 < tj; k0 += GB_TJ) {       // one pass for N <= 256
        float vm[GB_TJ], vs[GB_TJ];
#pragma unroll
        for (int k = 0; k < GB_TJ; ++k) {
            const bool ok = k0 + k < tj;
            vm[k] = ok ? pm[k0 + k] : -INFINITY;
            vs[k] = ok ? ps[k0 + k] : 0.f;
        }
        float m2 = mx;
#pragma unroll
        for (int k = 0; k < GB_TJ; ++k) m2 = fmaxf(m2, vm[k]);
        z *= __expf(mx - m2);                       // rescale the running sum (exp(-inf) = 0 on the first pass)
#pragma unroll
        for (int k = 0; k < GB_TJ; ++k) z += vs[k] * __expf(vm[k] - m2);
        mx = m2;
    }
    M = mx; coef = base / z;
}

template <bool L2, bool OTF>
__global__ __launch_bounds__(KGE_BLOCK) void neg_bwd_gemm_kernel(GemmArgs a, int ti, int tj, int td,
                                                                 int bpA, int bpN) {
    KGE_TL(3);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // workgroup -> (chunk, product, 4 consecutive tiles)
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int c = blk / (bpA + bpN);
    const int bc = blk % (bpA + bpN);
    const bool isGA = bc < bpA;
    const int tl = (isGA ? bc : bc - bpA) * KGE_WAVES_PER_BLOCK + wv;
    const int tr = isGA ? ti : tj;                       // row tiles of this product
    const bool tile_ok = tl < tr * td;
    const int dt = tl % td, rt = tl / td;
    const int D = a.D, N = a.N, chunk = a.chunk;
    const int m = lane & 15, q = lane >> 4;
    constexpr bool otf = OTF;                            // loss gradient applied on the fly to S
    const int K = isGA ? N : chunk;                      // reduction length
    const int R = isGA ? chunk : N;                      // output rows per chunk

    // ---- workgroup-shared tables in LDS ----
    // rix[k]: row index of reduction element k in the streamed operand (GA: negative row, gathered
    // through neg_ids or dense; GN: positive row of the chunk).  INDICES, not pointers: a pointer read
    // back from LDS turns the row loads into flat loads, which count on lgkmcnt and serialise behind
    // every LDS wait (measured: 48 % of the wavefront time parked in s_waitcnt).
    int64_t *rix = reinterpret_cast<int64_t *>(smem);                     // [K]
    float2 *st = reinterpret_cast<float2 *>(smem + 2 * GB_MAXK);          // [K] (GN, fused loss only)
    for (int k = threadIdx.x; k < K; k += KGE_BLOCK)
        rix[k] = isGA ? (a.nidx ? a.nidx[(int64_t)c * N + k] : (int64_t)c * N + k) : (int64_t)c * chunk + k;
    if (!isGA && otf) {
        for (int k = threadIdx.x; k < K; k += KGE_BLOCK) {
            float M, coef;
            row_stats(a, (int64_t)c * chunk + k, tj, M, coef);
            st[k] = make_float2(M, coef);
        }
    }
    __syncthreads();
    if (!tile_ok) return;

    const int d = dt * 64 + m * 4;                       // this lane's 4 output columns
    const bool dok = d < D;                              // D % 4 == 0: all-or-nothing
    const int dc = dok ? d : 0;
    const float *Wc = (otf ? a.Sc : a.W) + (int64_t)c * chunk * N;
    const float *Ac = a.A + (int64_t)c * chunk * D;
    const int row = rt * 16 + m;
    const bool rok = row < R;
    const int rowc = min(row, R - 1);
    const bool vecW = isGA && (N % 4 == 0);
    const float *Wrow = isGA ? Wc + (int64_t)rowc * N : Wc + rowc;
    const int64_t wstride = isGA ? 1 : N;
    // GA: this lane's row statistics; loss terms of the row are summed by the d-tile-0 wavefronts
    float rM = 0.f, rcoef = 0.f;
    if (otf && isGA) row_stats(a, (int64_t)c * chunk + rowc, tj, rM, rcoef);
    const bool do_loss = otf && isGA && dt == 0 && (a.row_neg || a.acc);
    float lsum = 0.f;

    f32x4 acc[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) acc[s_] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the output rows' own vectors (rank-1 term of the L2 expansion / regulariser) are requested NOW: they do
    // not depend on the loop, and at the end they would be one more dependent round (index -> row)
    const bool need_self = L2 || ((!isGA) && a.reg_coef > 0.f && a.reg_norm > 0);
    float4 selfv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        selfv[r] = zero4();
        if (need_self) {
            const int roc = min(rt * 16 + q * 4 + r, R - 1);
            const float *self = isGA ? Ac + (int64_t)roc * D : row_ptr(a.nbase, a.nidx, (int64_t)c * N + roc, D);
            selfv[r] = ldg4(self + dc);
        }
    }
    float wsum = 0.f;                                    // partial row (GA) / column (GN) sum of W
    BwdStage s0[BU], s1[BU];

    // Main loop over FULL macro steps: no per-lane predicates (rows / columns are clamped so every
    // load is in bounds; garbage in clamped rows only reaches outputs that are never stored), row
    // addresses = kernel-argument base + LDS index (global loads, vmcnt only).  One predicated tail
    // step handles K % 16.
    const int msfull = K >> 4;
    const float *Xb = (isGA ? a.nbase : a.A) + dc;                 // operand base (global address space)
    const float *Wq = Wrow + (int64_t)(q * 4) * wstride;

#define BWD_LOAD(ST, MS0)                                                                      \
    _Pragma("unroll") for (int u = 0; u < BU; ++u) {                                           \
        if ((MS0) + u < msfull) {                                                              \
            const int ms = (MS0) + u;                                                          \
            int64_t ri[4];                                                                     \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) ri[e] = rix[ms * 16 + q * 4 + e];    \
            if (vecW) {                                                                        \
                const float4 t4 = ldg4(Wq + ms * 16);                                          \
                ST[u].w[0] = t4.x; ST[u].w[1] = t4.y; ST[u].w[2] = t4.z; ST[u].w[3] = t4.w;    \
            } else {                                                                           \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                  \
                    ST[u].w[e] = Wq[(int64_t)(ms * 16 + e) * wstride];                         \
            }                                                                                  \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) ST[u].r[e] = ldg4(Xb + ri[e] * D);   \
        }                                                                                      \
    }
#define BWD_XFORM(ST, MS0)                                                                     \
    if (otf) {                                                                                 \
        _Pragma("unroll") for (int u = 0; u < BU; ++u) {                                       \
            if ((MS0) + u < msfull) {                                                          \
                const int kk = ((MS0) + u) * 16 + q * 4;                                       \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                \
                    const float n_ = ST[u].w[e];                                               \
                    float M_ = rM, cf_ = rcoef;                                                \
                    if (!isGA) { const float2 s2 = st[kk + e]; M_ = s2.x; cf_ = s2.y; }        \
                    if (do_loss) {                                                             \
                        float nl, dnl;                                                         \
                        criterion(a.lp.genre, n_, a.lp.genre == KGE_LOSS_BCE ? 0.f : -1.f, a.lp.margin, nl, dnl); \
                        lsum += nl * cf_ * (a.lp.adv ? __expf(n_ * a.lp.adv_temp - M_) : 1.f); \
                    }                                                                          \
                    ST[u].w[e] = wgrad<L2>(a.lp, a.gamma, n_, M_, cf_);                        \
                }                                                                              \
            }                                                                                  \
        }                                                                                      \
    }
#define BWD_MMA(ST, MS0)                                                                       \
    _Pragma("unroll") for (int u = 0; u < BU; ++u) {                                           \
        if ((MS0) + u < msfull) {                                                              \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                    \
                const float wgt = ST[u].w[e];                                                  \
                wsum += wgt;                                                                   \
                acc[0] = MFMA16(wgt, ST[u].r[e].x, acc[0]);                                    \
                acc[1] = MFMA16(wgt, ST[u].r[e].y, acc[1]);                                    \
                acc[2] = MFMA16(wgt, ST[u].r[e].z, acc[2]);                                    \
                acc[3] = MFMA16(wgt, ST[u].r[e].w, acc[3]);                                    \
            }                                                                                  \
        }                                                                                      \
    }

    BWD_LOAD(s0, 0);
    for (int g = 0; g < msfull; g += 2 * BU) {
        BWD_LOAD(s1, g + BU);
        BWD_XFORM(s0, g);
        BWD_MMA(s0, g);
        BWD_LOAD(s0, g + 2 * BU);
        BWD_XFORM(s1, g + BU);
        BWD_MMA(s1, g + BU);
    }
#undef BWD_LOAD
#undef BWD_XFORM
#undef BWD_MMA
    if (K & 15) {   // tail macro step: reduction indices beyond K get zero weight
        const int kk = msfull * 16 + q * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool kok = kk + e < K;
            const int kc = min(kk + e, K - 1);
            float wgt = Wrow[(int64_t)kc * wstride];
            const float4 xv = ldg4(Xb + rix[kc] * D);
            if (otf) {
                float M_ = rM, cf_ = rcoef;
                if (!isGA) { const float2 s2 = st[kc]; M_ = s2.x; cf_ = s2.y; }
                if (do_loss && kok) {
                    float nl, dnl;
                    criterion(a.lp.genre, wgt, a.lp.genre == KGE_LOSS_BCE ? 0.f : -1.f, a.lp.margin, nl, dnl);
                    lsum += nl * cf_ * (a.lp.adv ? __expf(wgt * a.lp.adv_temp - M_) : 1.f);
                }
                wgt = wgrad<L2>(a.lp, a.gamma, wgt, M_, cf_);
            }
            wgt = kok ? wgt : 0.f;
            wsum += wgt;
            acc[0] = MFMA16(wgt, xv.x, acc[0]);
            acc[1] = MFMA16(wgt, xv.y, acc[1]);
            acc[2] = MFMA16(wgt, xv.z, acc[2]);
            acc[3] = MFMA16(wgt, xv.w, acc[3]);
        }
    }

    // lanes with equal (lane&15) hold partial sums of the same W row/column: combine the 4 groups
    wsum += __shfl_xor(wsum, 16, 64);
    wsum += __shfl_xor(wsum, 32, 64);
    if (do_loss) {
        // negative loss term of the row: sum_j A_ij * nl_ij * w_i / B = 2 * sum_j nl * coef * e_ij
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        if (q == 0 && rok) {
            const int64_t gi = (int64_t)c * chunk + row;
            const float v = 2.f * lsum;
            if (a.row_neg) a.row_neg[gi] = v;
            if (a.acc) {
                const bool uq = a.B <= KGE_ACC_SLOTS;
                const int slot = (int)(gi & (KGE_ACC_SLOTS - 1));
                float *p1 = a.acc + 1 * KGE_ACC_SLOTS + slot, *p2 = a.acc + 2 * KGE_ACC_SLOTS + slot;
                if (uq) { *p1 += v; *p2 += 0.5f * v; } else { atomicAdd(p1, v); atomicAdd(p2, 0.5f * v); }
            }
        }
    }
    const bool reg = (!isGA) && a.reg_coef > 0.f && a.reg_norm > 0;
    float *O = isGA ? a.GA : a.GN;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ro = rt * 16 + q * 4 + r;
        const float rs = __shfl(wsum, q * 4 + r, 64);    // W row/column sum of output row ro
        if (ro < R && dok) {
            float4 o = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
            if (L2 || reg) {
                // the row's own vector (rank-1 term of the L2 expansion / regulariser)
                const float4 sv = selfv[r];
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

int launch_neg_bwd_gemm(const GemmArgs &a, hipStream_t s) {
    if (a.C == 0) return KGE_OK;
    if (a.W == nullptr && a.lp.pairwise) return KGE_ERR_ARG;     // on-the-fly gradient: pointwise losses
    const int maxK = a.chunk > a.N ? a.chunk : a.N;
    if (maxK > GB_MAXK) return KGE_ERR_ARG;
    const int ti = (a.chunk + 15) / 16, tj = (a.N + 15) / 16, td = (a.D + 63) / 64;
    const int bpA = (ti * td + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK;   // workgroups per chunk, GA
    const int bpN = (tj * td + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK;
    const int nb = a.C * (bpA + bpN);
    const size_t sm = (size_t)GB_MAXK * 16;      // row indices [GB_MAXK] int64 + row statistics [GB_MAXK] float2
    const bool l2 = a.model == KGE_TRANSE_L2, otf = a.W == nullptr;
    const dim3 g(nb), b(KGE_BLOCK);
    if (l2 && otf) hipLaunchKernelGGL((neg_bwd_gemm_kernel<true, true>), g, b, sm, s, a, ti, tj, td, bpA, bpN);
    else if (l2) hipLaunchKernelGGL((neg_bwd_gemm_kernel<true, false>), g, b, sm, s, a, ti, tj, td, bpA, bpN);
    else if (otf) hipLaunchKernelGGL((neg_bwd_gemm_kernel<false, true>), g, b, sm, s, a, ti, tj, td, bpA, bpN);
    else hipLaunchKernelGGL((neg_bwd_gemm_kernel<false, false>), g, b, sm, s, a, ti, tj, td, bpA, bpN);
    return check_launch_g();
}
