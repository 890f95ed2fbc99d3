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
// current one, v_mfma_f32_16x16x4_f32 (fp32 in / fp32 accumulate = exact FMA chain).
//
// Fused loss (pointwise criteria; round 2): there are only TWO grid-wide phases and no loss kernel.
// The loss gradient factorises per (row i, 16-column tile t):
//     dL/dn_ij = crit'(n_ij) * softmax_j(T n_ij) * w_i / 2B
//              = [ crit'(n_ij) * exp(T n_ij - m_it) ]  *  [ exp(m_it - M_i) * w_i / (2B Z_i) ]
//              =              u_ij                     *            f_it
// with m_it the maximum of T*n over the tile's 16 columns, M_i = max_t m_it, Z_i = sum_t s_it exp(m_it - M_i),
// s_it = sum_{j in t} exp(T n_ij - m_it).  The forward tile knows everything in u_ij (it also folds the
// 1/dist of the TransE_l2 chain rule in) and emits u instead of the scores, plus the partials (m, s, and
// l_it = sum_j exp(.) * criterion(n_ij) for the loss value); the backward wavefronts combine the <= tj partials
// of a row (a handful of exps per wavefront) and multiply the u fragment they load as MFMA operand by ONE
// factor per (row, macro step) - no transcendental work between the MFMAs (the first fused version applied the
// whole gradient there and was 2.5x slower), no atomics, no completion counters, bit-reproducible.
// Negative rows are gathered straight from the entity table through neg_ids (no dense copy).
// Fragment layout of the 16x16x4 f32 MFMA (wave64): A: lane l = A[m=l&15][k=l>>4],
// B: lane l = B[k=l>>4][n=l&15], C/D: lane l, reg r = D[4*(l>>4)+r][l&15].  A lane loads 4
// consecutive k (one float4) and feeds element e to MFMA step e.
#include "kge_common.hpp"
#include "kge_update_body.hpp"
#include "kge_edge_fwd_body.hpp"
#include "kge_sampler_tail.hpp"
#include "kge_loss_body.hpp"

using namespace kge;
KGE_TL_DEFINE(gemm)

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// program order is kept across this point: loads by the compiler-level memory barrier, everything else by the scheduling barrier
// MFMAs are pure values without a place in program order: the accumulators go through an empty volatile asm (AGPR operands, no
// instruction) so that the MFMAs producing them stay BEFORE this point
#define KGE_PIN_ACC(x, y) do { asm volatile("" : "+a"(x), "+a"(y) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define KGE_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

bool neg_mfma_supported(int model, int d_e, int N) {
    (void)N;
    if (model != KGE_TRANSE_L2 && model != KGE_DISTMULT && model != KGE_COMPLEX && model != KGE_SIMPLE &&
        model != KGE_RESCAL) return false;
    return d_e % 4 == 0;   // 16-byte aligned rows
}

// the fused loss keeps a [chunk][tiles] factor table in LDS and combines <= 16 partials per row in registers
bool neg_gemm_fused_loss_supported(int chunk, int N) { return N <= 256 && chunk <= 768; }

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

// `bid` / `nblk`: this workgroup's index and the number of workgroups doing forward-GEMM work (the body is also one half of
// the horizontally fused "forward GEMM of step s + update of step s-1" launch of the --async_update pipeline)
// AM: where the pos-side fragments come from - 0: the dense A buffer; 1 / 2 (merged launch, launch_neg_fwd_gemm_with_edge): built on
// the fly from the gathered table rows, a = x + asign * r (TransE) / a = x * r (DistMult) - then L2 means "emit the raw products"
// (|a|^2, |b|^2 are being written by the other half of the launch; the loss kernel applies the distance transform)
template <bool L2, bool FUSE, int AM = 0>     // FUSE: emit the factorised loss gradient + per-tile partials instead of the scores
__device__ __forceinline__ void neg_fwd_gemm_body(const GemmArgs &a, int ti, int tj, int bid, int nblk) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)xcd_remap(bid, nblk) * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
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
    // (AM: the three id loads - x row, r row, negative row - are one round; the rows follow in the loop)
    const float *Ap = AM ? row_ptr(a.xbase, a.xidx, (int64_t)c * a.chunk + ia, D) + q * 4
                         : a.A + ((int64_t)c * a.chunk + ia) * D + q * 4;
    const float *Rp = AM ? row_ptr(a.rbase, a.ridx, (int64_t)c * a.chunk + ia, D) + q * 4 : Ap;
    const float *Bp = row_ptr(a.nbase, a.nidx, (int64_t)c * a.N + jb, D) + q * 4;
    const int kq = q * 4;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float4 a0[FU], b0[FU], a1[FU], b1[FU];
    float4 r0[AM ? FU : 1], r1[AM ? FU : 1];
    const float asg = a.asign;
    // epilogue operands are requested NOW (they do not depend on the loop): one dependent ~1.5 us round less
    // at the end of every wavefront
    float bsq_pre = 0.f, asq_pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (L2 && !AM) {
        bsq_pre = a.bsq[(int64_t)c * a.N + jb];
#pragma unroll
        for (int r = 0; r < 4; ++r) asq_pre[r] = a.asq[(int64_t)c * a.chunk + min(it * 16 + q * 4 + r, a.chunk - 1)];
    }

    // main loop: only FULL k-steps (all 64 lanes in range) - no per-lane predicates, no exec-mask branches between the
    // MFMAs.  The loads are UNCONDITIONAL (k-step index clamped to the last full one, a redundant in-bounds load at the end):
    // a load under a branch - even a wave-uniform scalar one - makes the compiler's s_waitcnt pass assume at the join that it
    // was NOT issued, so the wait before the MFMAs of the older buffer became vmcnt(0) and also waited for the buffer just
    // requested: the double buffering was there in the source and absent in the ISA (profiles/r02_waitcnt_fix.txt).
#ifdef FWD_PROBE_NOLOOP            // tuning probe (wrong results): wavefront time without the main loop
    const int kfull = 0;
#else
    const int kfull = D >> 4;
#endif
    // (the k-step index goes through an empty volatile asm: these loads have no other tie to program order - read-only
    //  kernel-argument pointers - and were otherwise hoisted above the MFMAs that still read the buffer they refill)
#define FWD_LOAD(AV, RV, BV, KS0)                                                \
    { int k0_ = (KS0); asm volatile("" : "+s"(k0_));                             \
      _Pragma("unroll") for (int u = 0; u < FU; ++u) {                           \
        const int ks_ = min(k0_ + u, kfull - 1);                                 \
        AV[u] = ldg4(Ap + ks_ * 16); BV[u] = ldg4(Bp + ks_ * 16);                \
        if (AM) RV[u] = ldg4(Rp + ks_ * 16);                                     \
    } }
    // pos-side fragment from the gathered rows: four VALU operations per k-step, issued under the previous MFMAs
#define FWD_AFRAG(AV, RV, u)                                                     \
    { if (AM == 1) { AV[u].x = fmaf(asg, RV[u].x, AV[u].x); AV[u].y = fmaf(asg, RV[u].y, AV[u].y);    \
                     AV[u].z = fmaf(asg, RV[u].z, AV[u].z); AV[u].w = fmaf(asg, RV[u].w, AV[u].w); }  \
      else if (AM == 2) { AV[u].x *= RV[u].x; AV[u].y *= RV[u].y; AV[u].z *= RV[u].z; AV[u].w *= RV[u].w; } }
#define FWD_MMA1(AV, RV, BV, u)                                                  \
    { FWD_AFRAG(AV, RV, u)                                                       \
      acc0 = MFMA16(AV[u].x, BV[u].x, acc0);                                     \
      acc1 = MFMA16(AV[u].y, BV[u].y, acc1);                                     \
      acc0 = MFMA16(AV[u].z, BV[u].z, acc0);                                     \
      acc1 = MFMA16(AV[u].w, BV[u].w, acc1); }
#define FWD_MMA(AV, RV, BV) _Pragma("unroll") for (int u = 0; u < FU; ++u) FWD_MMA1(AV, RV, BV, u)
#define FWD_MMA_G(AV, RV, BV, KS0) _Pragma("unroll") for (int u = 0; u < FU; ++u) { if ((KS0) + u < kfull) FWD_MMA1(AV, RV, BV, u) }
    if (kfull > 0) {
        FWD_LOAD(a0, r0, b0, 0);
        int g = 0;
        for (; g + 2 * FU <= kfull; g += 2 * FU) {
            FWD_LOAD(a1, r1, b1, g + FU);
            KGE_ORDER();                                // the requests go out BEFORE the MFMAs on the other buffer, as written
            FWD_MMA(a0, r0, b0);
            KGE_PIN_ACC(acc0, acc1);                    // ... and a buffer is refilled only AFTER its MFMAs were issued
            FWD_LOAD(a0, r0, b0, g + 2 * FU);
            KGE_ORDER();
            FWD_MMA(a1, r1, b1);
            KGE_PIN_ACC(acc0, acc1);
        }
        if (g < kfull) {                     // fewer than 2 * FU k-steps left; a0 / b0 hold the first FU of them
            FWD_LOAD(a1, r1, b1, g + FU);
            FWD_MMA_G(a0, r0, b0, g);
            FWD_MMA_G(a1, r1, b1, g + FU);
        }
    }
#undef FWD_LOAD
#undef FWD_MMA1
#undef FWD_MMA
#undef FWD_MMA_G
    if (D & 15) {   // tail k-step: lanes whose 4 floats lie beyond D contribute zeros
        float4 av = zero4(), bv = zero4();
        if (kfull * 16 + kq < D) {
            av = ldg4(Ap + kfull * 16); bv = ldg4(Bp + kfull * 16);
            if (AM) {
                float4 rv[1]; rv[0] = ldg4(Rp + kfull * 16);
                float4 av1[1]; av1[0] = av;
                FWD_AFRAG(av1, rv, 0)
                av = av1[0];
            }
        }
        acc0 = MFMA16(av.x, bv.x, acc0);
        acc1 = MFMA16(av.y, bv.y, acc1);
        acc0 = MFMA16(av.z, bv.z, acc0);
        acc1 = MFMA16(av.w, bv.w, acc1);
    }
#undef FWD_AFRAG

    // NOTE: the MFMA stream above is kept free of VALU work on purpose - accumulating |a|^2,|b|^2
    // from the loaded fragments inside the loop cost +3.5 us (45 %) on MI355X; they come precomputed.
    const int j = jt * 16 + m;
    const bool jok = j < a.N;
    const float bsq = bsq_pre;
    const float neg_label = a.lp.genre == KGE_LOSS_BCE ? 0.f : -1.f;
    // The four rows of a lane are independent: every stage below runs over all four before the next stage starts,
    // so that the dependent chains (DPP reductions, exp / log / rcp) of the rows interleave - one wavefront per
    // SIMD has nothing else to hide a dependent VALU instruction's latency with.
    float v[4]; bool ok[4]; int64_t so[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = it * 16 + q * 4 + r;
        float x = acc0[r] + acc1[r];
        if (L2 && !AM) x = a.gamma - sqrtf(fmaxf(fmaf(-2.f, x, asq_pre[r] + bsq), 1e-30f));   // (explicit fma: same bits as the loss kernel's RAW transform)
        else if (!L2 && a.clampv > 0.f) x = fminf(fmaxf(x, -a.clampv), a.clampv);    // SimplE: th.clamp(tmp, -20, 20)
        v[r] = x;
        ok[r] = jok && i < a.chunk;
        so[r] = ((int64_t)c * a.chunk + i) * a.N + j;
    }
    if (!FUSE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (ok[r]) a.S[so[r]] = v[r];
        return;
    }
    // loss.py:82-94 on this tile's 16 columns of each row (the 16 lanes with the same q are one DPP row)
    if (a.Sraw) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (ok[r]) a.Sraw[so[r]] = v[r];
    }
    float mx[4], ex[4], nl[4], u[4], ps[4], pl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = (a.lp.adv && ok[r]) ? v[r] * a.adv_temp : (a.lp.adv ? -INFINITY : 0.f);
    if (a.lp.adv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], KGE_DPP_F(mx[r], KGE_DPP_QUAD_X1));
#pragma unroll
        for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], KGE_DPP_F(mx[r], KGE_DPP_QUAD_X2));
#pragma unroll
        for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], KGE_DPP_F(mx[r], KGE_DPP_HALF_MIRROR));
#pragma unroll
        for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], KGE_DPP_F(mx[r], KGE_DPP_MIRROR));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ex[r] = ok[r] ? (a.lp.adv ? __expf(v[r] * a.adv_temp - mx[r]) : 1.f) : 0.f;
        float dnl;
        criterion_fast(a.lp.genre, v[r], neg_label, a.lp.margin, nl[r], dnl);
        float t = dnl * ex[r];
        if (L2) { const float d = a.gamma - v[r]; t = d > 1e-15f ? t * __frcp_rn(d) : 0.f; }   // dn/d(a-b) = -(a-b)/dist
        else if (a.clampv > 0.f && fabsf(v[r]) >= a.clampv) t = 0.f;                            // saturated clamp: no gradient
        u[r] = t;
        ps[r] = ex[r]; pl[r] = ex[r] * nl[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) if (ok[r]) a.S[so[r]] = u[r];
#define KGE_ROWSTEP(CTRL) _Pragma("unroll") for (int r = 0; r < 4; ++r) { ps[r] += KGE_DPP_F(ps[r], CTRL); pl[r] += KGE_DPP_F(pl[r], CTRL); }
    KGE_ROWSTEP(KGE_DPP_QUAD_X1) KGE_ROWSTEP(KGE_DPP_QUAD_X2) KGE_ROWSTEP(KGE_DPP_HALF_MIRROR) KGE_ROWSTEP(KGE_DPP_MIRROR)
#undef KGE_ROWSTEP
    if (m == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + q * 4 + r;
            if (i < a.chunk) {
                const int64_t o = ((int64_t)c * a.chunk + i) * tj + jt;
                a.PM[o] = mx[r]; a.PS[o] = ps[r]; a.PL[o] = pl[r];
            }
        }
    }
}

template <bool L2, bool FUSE>
__global__ __launch_bounds__(KGE_BLOCK) void neg_fwd_gemm_kernel(GemmArgs a, int ti, int tj) {
    KGE_TL(1);
    neg_fwd_gemm_body<L2, FUSE>(a, ti, tj, (int)blockIdx.x, (int)gridDim.x);
}

// --async_update pipeline, horizontal fusion #1: forward GEMM tiles of step s (first nbG workgroups) + Adagrad update of
// step s-1 (the rest).  See neg_bwd_prep_kernel below for the other half of the schedule.
template <bool L2, int NIT, int LEAN>
__global__ __launch_bounds__(KGE_BLOCK) void neg_fwd_update_kernel(GemmArgs a, int ti, int tj, int nbG, UpdateArgs u, int nb_ent) {
    if ((int)blockIdx.x < nbG) {
        KGE_TL(1);
        neg_fwd_gemm_body<L2, false>(a, ti, tj, (int)blockIdx.x, nbG);
    } else {
        KGE_TL(4);
        update_reg_body<NIT, false, LEAN>(u, nb_ent, (int)blockIdx.x - nbG, (int)gridDim.x - nbG);
    }
}

// ---- wide forward tiles for the merged launch: one wavefront = 16 positives x (NB x 16) negatives.  Building the pos-side
// fragment on the fly costs a second gathered load per k-step (x and r rows); with ONE negative fragment per wavefront that is
// 3 fragment loads per 4 MFMAs and the tiles - bound by the texture addresser (a fragment load touches 16 rows x 64 B) - lived
// 9.3 us instead of 5.6 (profiles/r03_merged_fwd.txt).  NB = 2 shares the (x, r) pair between two negative fragments: 4 loads
// per 8 MFMAs, the ratio of the dense-A kernel, in half as many wavefronts.  Raw products out (see neg_fwd_edge_kernel).
template <int AM, int NB>
__device__ __forceinline__ void neg_fwd_gemm_wide_body(const GemmArgs &a, int ti, int tjg, int bid, int nblk) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)xcd_remap(bid, nblk) * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t ntiles = (int64_t)a.C * ti * tjg;
    if (tile >= ntiles) return;
    const int jg = (int)(tile % tjg);
    const int it = (int)((tile / tjg) % ti);
    const int c = (int)(tile / ((int64_t)tjg * ti));
    const int D = a.D;
    const int m = lane & 15, q = lane >> 4;
    const int ia = min(it * 16 + m, a.chunk - 1);
    const float *Ap = AM ? row_ptr(a.xbase, a.xidx, (int64_t)c * a.chunk + ia, D) + q * 4
                         : a.A + ((int64_t)c * a.chunk + ia) * D + q * 4;
    const float *Rp = AM ? row_ptr(a.rbase, a.ridx, (int64_t)c * a.chunk + ia, D) + q * 4 : Ap;
    const float *Bp[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
        Bp[n] = row_ptr(a.nbase, a.nidx, (int64_t)c * a.N + min((jg * NB + n) * 16 + m, a.N - 1), D) + q * 4;
    f32x4 acc[NB][2];
#pragma unroll
    for (int n = 0; n < NB; ++n) { acc[n][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[n][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    float4 a0[FU], a1[FU], r0[AM ? FU : 1], r1[AM ? FU : 1], b0[NB][FU], b1[NB][FU];
    const float asg = a.asign;
    const int kfull = D >> 4;
#define WF_LOAD(AV, RV, BV, KS0)                                                 \
    { int k0_ = (KS0); asm volatile("" : "+s"(k0_));                             \
      _Pragma("unroll") for (int u = 0; u < FU; ++u) {                           \
        const int ks_ = min(k0_ + u, kfull - 1);                                 \
        AV[u] = ldg4(Ap + ks_ * 16);                                             \
        if (AM) RV[u] = ldg4(Rp + ks_ * 16);                                     \
        _Pragma("unroll") for (int n = 0; n < NB; ++n) BV[n][u] = ldg4(Bp[n] + ks_ * 16); \
    } }
#define WF_AFRAG(AV, RV, u)                                                      \
    { if (AM == 1) { AV[u].x = fmaf(asg, RV[u].x, AV[u].x); AV[u].y = fmaf(asg, RV[u].y, AV[u].y);    \
                     AV[u].z = fmaf(asg, RV[u].z, AV[u].z); AV[u].w = fmaf(asg, RV[u].w, AV[u].w); }  \
      else if (AM == 2) { AV[u].x *= RV[u].x; AV[u].y *= RV[u].y; AV[u].z *= RV[u].z; AV[u].w *= RV[u].w; } }
#define WF_MMA1(AV, RV, BV, u)                                                   \
    { WF_AFRAG(AV, RV, u)                                                        \
      _Pragma("unroll") for (int n = 0; n < NB; ++n) {                           \
        acc[n][0] = MFMA16(AV[u].x, BV[n][u].x, acc[n][0]);                      \
        acc[n][1] = MFMA16(AV[u].y, BV[n][u].y, acc[n][1]); }                    \
      _Pragma("unroll") for (int n = 0; n < NB; ++n) {                           \
        acc[n][0] = MFMA16(AV[u].z, BV[n][u].z, acc[n][0]);                      \
        acc[n][1] = MFMA16(AV[u].w, BV[n][u].w, acc[n][1]); } }
#define WF_MMA(AV, RV, BV) _Pragma("unroll") for (int u = 0; u < FU; ++u) WF_MMA1(AV, RV, BV, u)
#define WF_MMA_G(AV, RV, BV, KS0) _Pragma("unroll") for (int u = 0; u < FU; ++u) { if ((KS0) + u < kfull) WF_MMA1(AV, RV, BV, u) }
#define WF_PIN() do { _Pragma("unroll") for (int n = 0; n < NB; ++n) asm volatile("" : "+a"(acc[n][0]), "+a"(acc[n][1]) :: "memory"); \
                      __builtin_amdgcn_sched_barrier(0); } while (0)
    if (kfull > 0) {
        WF_LOAD(a0, r0, b0, 0);
        int g = 0;
        for (; g + 2 * FU <= kfull; g += 2 * FU) {
            WF_LOAD(a1, r1, b1, g + FU);
            KGE_ORDER();
            WF_MMA(a0, r0, b0);
            WF_PIN();
            WF_LOAD(a0, r0, b0, g + 2 * FU);
            KGE_ORDER();
            WF_MMA(a1, r1, b1);
            WF_PIN();
        }
        if (g < kfull) {
            WF_LOAD(a1, r1, b1, g + FU);
            WF_MMA_G(a0, r0, b0, g);
            WF_MMA_G(a1, r1, b1, g + FU);
        }
    }
    if (D & 15) {   // tail k-step: lanes whose 4 floats lie beyond D contribute zeros
        float4 av[1], rv[1], bv[NB][1];
        av[0] = zero4(); rv[0] = zero4();
#pragma unroll
        for (int n = 0; n < NB; ++n) bv[n][0] = zero4();
        if (kfull * 16 + q * 4 < D) {
            av[0] = ldg4(Ap + kfull * 16);
            if (AM) rv[0] = ldg4(Rp + kfull * 16);
#pragma unroll
            for (int n = 0; n < NB; ++n) bv[n][0] = ldg4(Bp[n] + kfull * 16);
        }
        WF_MMA1(av, rv, bv, 0)
    }
#undef WF_LOAD
#undef WF_AFRAG
#undef WF_MMA1
#undef WF_MMA
#undef WF_MMA_G
#undef WF_PIN
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int j = (jg * NB + n) * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + q * 4 + r;
            if (j < a.N && i < a.chunk) a.S[((int64_t)c * a.chunk + i) * a.N + j] = acc[n][0][r] + acc[n][1][r];
        }
    }
}

// ---- merged launch, pos-side tile through LDS: a workgroup = one row tile (16 positives) x 4 consecutive column tiles.  The 16
// pos-side vectors a_i = x_i + asign * r_i (TransE) / x_i * r_i (DistMult) are built ONCE per workgroup from coalesced row loads
// (16 threads per row, 256 contiguous bytes per pass) and parked in LDS; the wavefronts take their A fragments from there and only
// the B fragments (gathered negative rows) are direct global loads - one gathered fragment load per k-step instead of three
// (x, r, b: the tiles of the first merged version lived 9.3 us, ~0.1 us per fragment load; profiles/r03_merged_fwd.txt).
// Row stride of the tile: whole k-steps + 4 floats (the 16-byte fragment reads of a quarter wavefront fall on different banks).
static inline int fwd_lds_stride(int D) { return ((D + 15) & ~15) + 4; }
static inline size_t fwd_lds_bytes(int D) { return (size_t)16 * fwd_lds_stride(D) * sizeof(float); }
// loss-fold instance (round 4): + |a_i|^2 of the 16 rows + the "this workgroup arrived last" word, in the SAME shared array
static inline size_t fwd_lds_bytes_lf(int D) { return fwd_lds_bytes(D) + 32 * sizeof(float); }

// LF (round 4, the strict step's 3-launch form): the loss rows of a 16-row strip run INSIDE this launch.  The tiles store final
// scores (TransE_l2: |a_i|^2 from the LDS tile, |b_j|^2 accumulated from the B fragments under the MFMAs) write-through, every
// workgroup of the strip draws an arrival ticket, and the workgroup that arrives last runs LossGenerator on the strip's rows
// (kge_loss_body.hpp - the stand-alone kernel's code) with L1-bypassing loads.  Placement-independent hand-off
// (MI355X_MICROARCH.md, Workgroup dispatch: sc1 stores -> vmcnt(0) -> barrier -> relaxed agent-scope fetch_add; reader: sc1 loads);
// no workgroup ever waits for another one.  The ticket word is returned to 0 by the last arriver (kge_step_out.tickets).
template <int AM, bool LF = false, bool LLEAN = false>
__device__ __forceinline__ void neg_fwd_gemm_ldsa_body(const GemmArgs &a, int ti, int tj, int bid, int nblk, float *As,
                                                       const LossArgs *lap = nullptr, int *tickets = nullptr) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tjg = (tj + 3) >> 2;
    // workgroup -> (chunk, strip, column group).  Round 4: when the last column group of a strip is partial (N = 200: 13 tiles =
    // 4 + 4 + 4 + 1) the FULL groups take the first block ids and the partial ones the last: at cfg-T 260 workgroups meet 256
    // CUs, and the four CUs that get a second workgroup should get a 1-tile one, not a second set of four loops (the doubled
    // CUs set the end of the launch: tile wavefronts p90 7.0 us, max 8.2 us, profiles/r04_loss_fold.txt).  Same tiles, same
    // arithmetic; only where they run changes.
    int jg, sidx;
#ifndef KGE_FWD_PLAIN_ORDER
    const int nfull = (tj & 3) ? tjg - 1 : tjg, nheavy = a.C * ti * nfull;
    if (bid < nheavy) { const int L = xcd_remap(bid, nheavy); jg = L % nfull; sidx = L / nfull; }
    else { sidx = xcd_remap(bid - nheavy, nblk - nheavy); jg = tjg - 1; }
#else
    { const int L = xcd_remap(bid, nblk); jg = L % tjg; sidx = L / tjg; }
#endif
    const int it = sidx % ti, c = sidx / ti;
    const int jt = min(jg * 4 + wv, tj - 1);             // (a wavefront beyond the last column tile helps building A, then leaves)
    const bool tile_ok = jg * 4 + wv < tj;
    const int D = a.D, KP = ((D + 15) & ~15) + 4, n4 = D >> 2;
    const int m = lane & 15, q = lane >> 4;
    // ids first (one round): this thread's row of the A tile (x and r) and this lane's negative row
    const int arow = threadIdx.x >> 4, ac = threadIdx.x & 15;
    const int64_t ie = (int64_t)c * a.chunk + min(it * 16 + arow, a.chunk - 1);
    const int64_t jl = (int64_t)c * a.N + min(jt * 16 + m, a.N - 1);
    const int64_t xid = a.xidx[ie], rid = a.ridx[ie];
    const int64_t jraw = *(a.nidx ? a.nidx + jl : a.xidx);            // (unconditional load: no branch join in front of the waits)
    const float *Xs = a.xbase + xid * D, *Rs = a.rbase + rid * D;
    const float *Bp = a.nbase + (a.nidx ? jraw : jl) * D + q * 4;
    float *Adst = As + arow * KP;
    const float *Ap = As + m * KP + q * 4;               // this lane's fragment of k-step ks: Ap + ks * 16 (LDS)
    const float asg = a.asign;
    const int kfull = D >> 4;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float4 a0[FU], b0[FU], a1[FU], b1[FU];
    // the A tile: 8 passes of 16 x 16 bytes per row and source (512 floats per row) requested together, then the first B fragments
    float4 xv[8], rv[8];
    float asq_t = 0.f, bs0 = 0.f, bs1 = 0.f;             // LF, TransE_l2: partial |a_i|^2 (A build) / |b_j|^2 (main loop)
    if constexpr (AM == 3) {
        // ComplEx: rows are [re | im] halves; a = x o r (tail-corrupted step) / x o conj(r) (head-corrupted, x = tail):
        //   a_re = x_re c -/+ x_im s,  a_im = +/- x_re s + x_im c   (score_fun.py:347-371; same fmaf forms as edge_fwd_body)
        // a thread takes the SAME 16-byte piece of both halves: 4 passes x (x_re, x_im, r_re, r_im)
        const int hd = D >> 1, h4 = hd >> 2;
        for (int c0 = 0; c0 < h4; c0 += 16 * 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = min(c0 + j * 16 + ac, h4 - 1) * 4;
                xv[j] = ldg4(Xs + o); xv[4 + j] = ldg4(Xs + hd + o); rv[j] = ldg4(Rs + o); rv[4 + j] = ldg4(Rs + hd + o);
            }
            if (c0 == 0) {
#pragma unroll
                for (int u = 0; u < FU; ++u) b0[u] = ldg4(Bp + (kfull > 0 ? min(u, kfull - 1) * 16 : -(q * 4)));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c4 = c0 + j * 16 + ac;
                if (c4 < h4) {
                    const float4 xr = xv[j], xi = xv[4 + j], cc = rv[j], ss = rv[4 + j];
                    float4 re, im;
#define LA_CX(E) re.E = fmaf(-asg * xi.E, ss.E, xr.E * cc.E); im.E = fmaf(asg * xr.E, ss.E, xi.E * cc.E);
                    LA_CX(x) LA_CX(y) LA_CX(z) LA_CX(w)
#undef LA_CX
                    *reinterpret_cast<float4 *>(Adst + c4 * 4) = re;
                    *reinterpret_cast<float4 *>(Adst + hd + c4 * 4) = im;
                }
            }
        }
    } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int o = min(j * 16 + ac, n4 - 1) * 4; xv[j] = ldg4(Xs + o); rv[j] = ldg4(Rs + o); }
#pragma unroll
    for (int u = 0; u < FU; ++u) b0[u] = ldg4(Bp + (kfull > 0 ? min(u, kfull - 1) * 16 : -(q * 4)));
#define LA_COMB(X, R) (AM == 1 ? make_float4(fmaf(asg, R.x, X.x), fmaf(asg, R.y, X.y), fmaf(asg, R.z, X.z), fmaf(asg, R.w, X.w)) \
                               : make_float4(X.x * R.x, X.y * R.y, X.z * R.z, X.w * R.w))
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c4 = j * 16 + ac;
        if (c4 < n4) {
            const float4 av = LA_COMB(xv[j], rv[j]);
            *reinterpret_cast<float4 *>(Adst + c4 * 4) = av;
            if constexpr (LF && AM == 1) asq_t = fmaf(av.x, av.x, fmaf(av.y, av.y, fmaf(av.z, av.z, fmaf(av.w, av.w, asq_t))));
        }
    }
    for (int c0 = 16 * 8; c0 < n4; c0 += 16 * 8) {       // rows longer than 512 floats
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int o = min(c0 + j * 16 + ac, n4 - 1) * 4; xv[j] = ldg4(Xs + o); rv[j] = ldg4(Rs + o); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c4 = c0 + j * 16 + ac;
            if (c4 < n4) {
                const float4 av = LA_COMB(xv[j], rv[j]);
                *reinterpret_cast<float4 *>(Adst + c4 * 4) = av;
                if constexpr (LF && AM == 1) asq_t = fmaf(av.x, av.x, fmaf(av.y, av.y, fmaf(av.z, av.z, fmaf(av.w, av.w, asq_t))));
            }
        }
    }
#undef LA_COMB
    }
    float *Aq = As + 16 * KP;                            // LF: |a_i|^2 of the tile's 16 rows, then the last-arriver word
    if constexpr (LF && AM == 1) {
        asq_t = kge::row_sum16(asq_t);                   // the 16 threads of a row are one DPP row
        if (ac == 0) Aq[arow] = asq_t;
    }
    if (ac < (KP - D) / 4) *reinterpret_cast<float4 *>(Adst + D + ac * 4) = zero4();   // columns D .. KP-1 (D % 4 == 0): zeros
    __syncthreads();
    if (!LF && !tile_ok) return;
    if (tile_ok) {
#define LA_LOAD(AV, BV, KS0)                                                     \
    { int k0_ = (KS0); asm volatile("" : "+s"(k0_));                             \
      _Pragma("unroll") for (int u = 0; u < FU; ++u) {                           \
        const int ks_ = min(k0_ + u, kfull - 1);                                 \
        AV[u] = *reinterpret_cast<const float4 *>(Ap + ks_ * 16); BV[u] = ldg4(Bp + ks_ * 16); \
    } }
    // (LF, TransE_l2: four VALU fmas per k-step on the B fragment just fed to the matrix core - |b_j|^2 of this lane's quarter row)
#define LA_MMA1(AV, BV, u)                                                       \
    { acc0 = MFMA16(AV[u].x, BV[u].x, acc0);                                     \
      acc1 = MFMA16(AV[u].y, BV[u].y, acc1);                                     \
      acc0 = MFMA16(AV[u].z, BV[u].z, acc0);                                     \
      acc1 = MFMA16(AV[u].w, BV[u].w, acc1);                                     \
      if constexpr (LF && AM == 1) { bs0 = fmaf(BV[u].x, BV[u].x, bs0); bs1 = fmaf(BV[u].y, BV[u].y, bs1);   \
                                     bs0 = fmaf(BV[u].z, BV[u].z, bs0); bs1 = fmaf(BV[u].w, BV[u].w, bs1); } }
#define LA_MMA(AV, BV) _Pragma("unroll") for (int u = 0; u < FU; ++u) LA_MMA1(AV, BV, u)
#define LA_MMA_G(AV, BV, KS0) _Pragma("unroll") for (int u = 0; u < FU; ++u) { if ((KS0) + u < kfull) LA_MMA1(AV, BV, u) }
    if (kfull > 0) {
#pragma unroll
        for (int u = 0; u < FU; ++u) a0[u] = *reinterpret_cast<const float4 *>(Ap + min(u, kfull - 1) * 16);
        int g = 0;
        for (; g + 2 * FU <= kfull; g += 2 * FU) {
            LA_LOAD(a1, b1, g + FU);
            KGE_ORDER();
            LA_MMA(a0, b0);
            KGE_PIN_ACC(acc0, acc1);
            LA_LOAD(a0, b0, g + 2 * FU);
            KGE_ORDER();
            LA_MMA(a1, b1);
            KGE_PIN_ACC(acc0, acc1);
        }
        if (g < kfull) {
            LA_LOAD(a1, b1, g + FU);
            LA_MMA_G(a0, b0, g);
            LA_MMA_G(a1, b1, g + FU);
        }
    }
#undef LA_LOAD
#undef LA_MMA1
#undef LA_MMA
#undef LA_MMA_G
    if (D & 15) {   // tail k-step: the tile is zero-padded in LDS; B lanes beyond D contribute zeros
        float4 bv = zero4();
        const float4 av = *reinterpret_cast<const float4 *>(Ap + kfull * 16);
        if (kfull * 16 + q * 4 < D) bv = ldg4(Bp + kfull * 16);
        acc0 = MFMA16(av.x, bv.x, acc0);
        acc1 = MFMA16(av.y, bv.y, acc1);
        acc0 = MFMA16(av.z, bv.z, acc0);
        acc1 = MFMA16(av.w, bv.w, acc1);
        if constexpr (LF && AM == 1) { bs0 = fmaf(bv.x, bv.x, bs0); bs1 = fmaf(bv.y, bv.y, bs1);
                                       bs0 = fmaf(bv.z, bv.z, bs0); bs1 = fmaf(bv.w, bv.w, bs1); }
    }
    const int j = jt * 16 + m;
    if constexpr (!LF) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + q * 4 + r;
            if (j < a.N && i < a.chunk) a.S[((int64_t)c * a.chunk + i) * a.N + j] = acc0[r] + acc1[r];
        }
    } else {
        // final scores, written through to the memory side (sc1): whichever workgroup of the strip arrives last reads them
        float bsq = 0.f;
        if constexpr (AM == 1) {             // the four quarter rows of column j sit in lanes m, m + 16, m + 32, m + 48
            bsq = bs0 + bs1;
            bsq += __shfl_xor(bsq, 16);
            bsq += __shfl_xor(bsq, 32);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + q * 4 + r;
            float x = acc0[r] + acc1[r];
            if constexpr (AM == 1) x = a.gamma - sqrtf(fmaxf(fmaf(-2.f, x, Aq[q * 4 + r] + bsq), 1e-30f));   // score_fun.py:26-34
            if (j < a.N && i < a.chunk)
                __hip_atomic_store(a.S + ((int64_t)c * a.chunk + i) * a.N + j, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    }   // tile_ok
    if constexpr (LF) {
        // ---- hand-off: every workgroup of the strip (16 positives of chunk c) draws a ticket; the last one runs the loss rows ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wavefront's score stores have been acknowledged
        __syncthreads();
        int *lastw = reinterpret_cast<int *>(Aq + 16);
        if (threadIdx.x == 0) {
            int *tk = tickets + (c * ti + it);
            const int old = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == tjg - 1;
            if (last) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next step
            else if (old < 0 || old >= tjg) __builtin_trap();         // the caller did not hand over zeroed tickets
            *lastw = last;
        }
        __syncthreads();
        if (!*lastw) return;
        LossArgs la = *lap;
        if constexpr (LLEAN) loss_args_lean(la);
        constexpr int NPER = 4;                   // N <= 256 (checked by the launcher)
        const int N = la.N;
        const int row0 = it * 16 + wv * 4;        // this wavefront's four rows of the strip
        // all four rows requested together, L1-bypassing (the lines were written by other CUs - possibly other XCDs - during this
        // launch); indices clamped instead of predicated (no branch joins in front of the waits)
        float nv[4][NPER];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gi = (int64_t)c * a.chunk + min(row0 + r, a.chunk - 1);
#pragma unroll
            for (int u = 0; u < NPER; ++u)
                nv[r][u] = __hip_atomic_load(a.S + gi * N + min(lane + 64 * u, N - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        float wr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) wr[r] = la.w ? la.w[(int64_t)c * a.chunk + min(row0 + r, a.chunk - 1)] : 1.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (row0 + r < a.chunk) {             // (wave-uniform)
                const int64_t gi = (int64_t)c * a.chunk + row0 + r;
#pragma unroll
                for (int u = 0; u < NPER; ++u) if (lane + 64 * u >= N) nv[r][u] = 0.f;
                // the edge half of this launch adds the row's positive share to slot gi of the running total: this half takes
                // the slot B further on (one add per slot and launch keeps the sums order-independent while 2 B <= KGE_ACC_SLOTS)
                loss_row_regs<NPER>(la, gi, nv[r], wr[r], 0.f, lane, (int)((gi + la.B) & (KGE_ACC_SLOTS - 1)));
            }
        }
    }
}

// the strict step's FIRST launch (round 3): forward GEMM tiles (first nbG workgroups) + the edge-forward rows of the SAME step
// (the rest).  The tiles do not wait for the pos-side vectors a_i: they rebuild their fragments from the table rows the edge
// half is reading (one packed add / multiply per fragment), emit raw products, and the row-wise results of the edge half
// (positive scores, |a|^2, |b|^2, dL/dp, P rows, the dense A the backward reads) are consumed one launch later, by the loss
// kernel (distance transform) and the backward GEMM.  One launch boundary and the whole edge-forward kernel (4.2 + 1.7 us at
// cfg-T, profiles/r02_v7_timeline.txt) leave the step's critical path.
#ifndef KGE_FWD_NB
#define KGE_FWD_NB 1               // direct-load instance: negative fragments per forward wavefront (tuning: 1, 2, 3; measured:
#endif                             // wider is SLOWER - 9.3 / 11.7 / 14 us wave life, the time follows the loads per wavefront)
// (round 5) ... + the phase-1 workgroups of the sampler tail building a batch of the NEXT group (kge_sampler_tail.hpp): the last
// workgroups of the grid, behind the nbM workgroups of the step itself
template <bool L2, int AM, int MODEL, bool LEAN, bool LDSA>
__global__ __launch_bounds__(KGE_BLOCK) void neg_fwd_edge_kernel(GemmArgs a, int ti, int tj, int nbG, EdgeFwdArgs e, int nbM, SmpTail st) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x < nbG) {
        KGE_TL(1);
        if constexpr (LDSA) neg_fwd_gemm_ldsa_body<AM>(a, ti, tj, (int)blockIdx.x, nbG, smem);
        else if constexpr (KGE_FWD_NB == 1) neg_fwd_gemm_body<L2, false, AM>(a, ti, tj, (int)blockIdx.x, nbG);
        else neg_fwd_gemm_wide_body<AM, KGE_FWD_NB>(a, ti, (tj + KGE_FWD_NB - 1) / KGE_FWD_NB, (int)blockIdx.x, nbG);
    } else if ((int)blockIdx.x < nbM) {
        KGE_TL(0);
        edge_fwd_body<MODEL, 4, LEAN>(e, (int)blockIdx.x - nbG);
    } else {
        sampler_tail_p1(st, (int)blockIdx.x - nbM);
    }
}

// round 4: the same launch with the loss rows inside (LF tiles, see neg_fwd_gemm_ldsa_body) - the strict step's first of THREE launches
template <int AM, int MODEL, bool LEAN>
__global__ __launch_bounds__(KGE_BLOCK) void neg_fwd_loss_edge_kernel(GemmArgs a, int ti, int tj, int nbG, EdgeFwdArgs e, LossArgs la,
                                                                      int *tickets) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x < nbG) {
        KGE_TL(1);
        neg_fwd_gemm_ldsa_body<AM, true, LEAN>(a, ti, tj, (int)blockIdx.x, nbG, smem, &la, tickets);
    } else {
        KGE_TL(0);
        edge_fwd_body<MODEL, 4, LEAN>(e, (int)blockIdx.x - nbG);
    }
}

bool neg_fwd_gemm_with_edge_supported(int model, int d_e, int d_r) {
    if (model == KGE_COMPLEX) return d_e % 8 == 0 && d_r == d_e && fwd_lds_bytes(d_e) <= 64 * 1024;   // (LDS tile instance only)
    return (model == KGE_TRANSE_L2 || model == KGE_DISTMULT) && d_e % 4 == 0 && d_r == d_e;
}

int launch_neg_fwd_gemm_with_edge(const GemmArgs &a, const EdgeFwdArgs &e, hipStream_t s, const SmpTail *tail) {
    if (a.C == 0 || a.PM || !a.xbase || !a.rbase || !a.xidx || !a.ridx) return KGE_ERR_ARG;
    SmpTail st{};
    if (tail && tail->phase) { st = *tail; st.phase = 1; }
    if (!neg_fwd_gemm_with_edge_supported(a.model, a.D, e.d_r) || e.model != a.model || e.d_e != a.D) return KGE_ERR_ARG;
    if (e.src.em.n || e.src.rm.n || e.nd_own) return KGE_ERR_ARG;
    const bool negjob = e.bsq || e.Bn;
    const int64_t waves = (int64_t)e.B + (negjob ? e.n_neg : 0);
    EdgeFwdArgs ee = e;
    if (!negjob) ee.n_neg = 0;
    const int nbP = (int)((waves + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK);
    const int ti = (a.chunk + 15) / 16, tj = (a.N + 15) / 16;
    // pos-side tile through LDS (one workgroup = 16 positives x 4 column tiles) whenever the tile fits the default 64 KB
    const size_t lds = fwd_lds_bytes(a.D);
    const bool ldsa = lds <= 64 * 1024 && !a.lds_off;
    const int64_t ntiles = (int64_t)a.C * ti * ((tj + KGE_FWD_NB - 1) / KGE_FWD_NB);
    const int nbG = ldsa ? a.C * ti * ((tj + 3) / 4) : (int)((ntiles + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK);
    const bool lean = e.lp.genre == KGE_LOSS_LOGSIGMOID && !e.row_pos && !e.Hc;
    const int nbM = nbG + nbP;
    const dim3 g(nbM + (st.phase ? ST_P1_WGS + (st.slot3 ? ST_P3_WGS : 0) : 0)), b(KGE_BLOCK);
#define KGE_FE2(L2_, AM_, M_, LE_) do { if (ldsa) hipLaunchKernelGGL((neg_fwd_edge_kernel<L2_, AM_, M_, LE_, true>), g, b, lds, s, a, ti, tj, nbG, ee, nbM, st); \
                                        else hipLaunchKernelGGL((neg_fwd_edge_kernel<L2_, AM_, M_, LE_, false>), g, b, 0, s, a, ti, tj, nbG, ee, nbM, st); } while (0)
#define KGE_FE(L2_, AM_, M_) do { if (lean) KGE_FE2(L2_, AM_, M_, true); else KGE_FE2(L2_, AM_, M_, false); } while (0)
    if (a.model == KGE_COMPLEX) {
        if (!ldsa) return KGE_ERR_ARG;
        if (lean) hipLaunchKernelGGL((neg_fwd_edge_kernel<false, 3, KGE_COMPLEX, true, true>), g, b, lds, s, a, ti, tj, nbG, ee, nbM, st);
        else hipLaunchKernelGGL((neg_fwd_edge_kernel<false, 3, KGE_COMPLEX, false, true>), g, b, lds, s, a, ti, tj, nbG, ee, nbM, st);
    } else if (a.model == KGE_TRANSE_L2) KGE_FE(true, 1, KGE_TRANSE_L2);
    else KGE_FE(false, 2, KGE_DISTMULT);
#undef KGE_FE
#undef KGE_FE2
    return check_launch_g();
}

// loss rows inside the first launch: LDS-tile instance only, one register-resident score row per wavefront (N <= 256), every
// strip's block of S on its own 128-byte lines (a line shared by two strips could be cached by one strip's reader before the
// other strip's tiles have written their part), ticket words for every strip
bool neg_fwd_loss_fold_supported(int model, int C, int chunk, int N, int d_e, int d_r) {
    if (!neg_fwd_gemm_with_edge_supported(model, d_e, d_r) || fwd_lds_bytes_lf(d_e) > 64 * 1024) return false;
    if (N > 256 || (N & 1) || (((int64_t)chunk * N) & 31)) return false;
    return (int64_t)C * ((chunk + 15) / 16) <= KGE_TICKET_INTS;
}

int launch_neg_fwd_gemm_with_edge_loss(const GemmArgs &a, const EdgeFwdArgs &e, const LossArgs &la, int *tickets, hipStream_t s) {
    if (a.C == 0 || a.PM || !a.xbase || !a.rbase || !a.xidx || !a.ridx || !tickets || a.lds_off) return KGE_ERR_ARG;
    if (!neg_fwd_loss_fold_supported(a.model, a.C, a.chunk, a.N, a.D, e.d_r) || e.model != a.model || e.d_e != a.D) return KGE_ERR_ARG;
    if (e.src.em.n || e.src.rm.n || e.nd_own || la.pairwise || !la.skip_pos || la.l2_raw || la.diag_chunk > 0 || la.neg != a.S ||
        la.dneg != a.S || la.N != a.N || la.B != a.C * a.chunk) return KGE_ERR_ARG;
    const bool negjob = e.bsq || e.Bn;
    const int64_t waves = (int64_t)e.B + (negjob ? e.n_neg : 0);
    EdgeFwdArgs ee = e;
    if (!negjob) ee.n_neg = 0;
    const int nbP = (int)((waves + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK);
    const int ti = (a.chunk + 15) / 16, tj = (a.N + 15) / 16;
    const size_t lds = fwd_lds_bytes_lf(a.D);
    const int nbG = a.C * ti * ((tj + 3) / 4);
    // ONE lean switch for both halves: Logsigmoid, no per-step outputs
    const bool lean = e.lp.genre == KGE_LOSS_LOGSIGMOID && !e.row_pos && !e.Hc &&
                      la.genre == KGE_LOSS_LOGSIGMOID && la.clampv == 0.f && !la.neg_copy && !la.row_pos && !la.row_neg;
    const dim3 g(nbG + nbP), b(KGE_BLOCK);
#define KGE_FL(AM_, M_) do { if (lean) hipLaunchKernelGGL((neg_fwd_loss_edge_kernel<AM_, M_, true>), g, b, lds, s, a, ti, tj, nbG, ee, la, tickets); \
                             else hipLaunchKernelGGL((neg_fwd_loss_edge_kernel<AM_, M_, false>), g, b, lds, s, a, ti, tj, nbG, ee, la, tickets); } while (0)
    if (a.model == KGE_COMPLEX) KGE_FL(3, KGE_COMPLEX);
    else if (a.model == KGE_TRANSE_L2) KGE_FL(1, KGE_TRANSE_L2);
    else KGE_FL(2, KGE_DISTMULT);
#undef KGE_FL
    return check_launch_g();
}

int launch_neg_fwd_gemm(const GemmArgs &a, hipStream_t s) {
    const int ti = (a.chunk + 15) / 16, tj = (a.N + 15) / 16;
    const int64_t ntiles = (int64_t)a.C * ti * tj;
    if (ntiles == 0) return KGE_OK;
    const int nb = (int)((ntiles + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK);
    const bool l2 = a.model == KGE_TRANSE_L2, st = a.PM != nullptr;       // PM/PS/PL given: fused loss
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

struct BwdStage { float w[4]; float4 r[4]; float pm; };

// combine the per-16-column partials of one score row (tj <= 16 tiles, i.e. N <= 256; longer rows use the stand-alone
// loss kernel): M = max_t m_t, coef = w_i / (2B) / (adv ? Z : N).  All partial loads are issued before the first use
// (independent registers) so that they overlap instead of forming a latency chain.
//   LOSS: also the row's negative loss term (loss.py:87-94) from the l_t partials;
//   fac != null: the factor f_t = exp(m_t - M) * coef of every tile is written to fac[t] (LDS, GN workgroups).
#define GB_TJ 16
template <bool LOSS>
__device__ __forceinline__ void row_stats(const GemmArgs &a, int64_t gi, int tj, float &M, float &coef, float &lrow,
                                          float *fac = nullptr) {
    const float w = a.w ? a.w[gi] : 1.f;
    const float base = w * 0.5f / (float)a.B;
    const float *pm = a.PM + gi * tj, *ps = a.PS + gi * tj, *pl = a.PL + gi * tj;
    float vm[GB_TJ], vs[GB_TJ], vl[GB_TJ];
#pragma unroll
    for (int k = 0; k < GB_TJ; ++k) {
        const bool ok = k < tj;
        vm[k] = (ok && a.lp.adv) ? pm[k] : -INFINITY;
        vs[k] = (ok && a.lp.adv) ? ps[k] : 0.f;
        vl[k] = (LOSS && ok) ? pl[k] : 0.f;
    }
    if (!a.lp.adv) {
        M = 0.f; coef = base / (float)a.N;
        if (LOSS) {
            float l = 0.f;
#pragma unroll
            for (int k = 0; k < GB_TJ; ++k) l += vl[k];
            lrow = 2.f * coef * l;
        }
        if (fac) for (int k = 0; k < tj; ++k) fac[k] = coef;
        return;
    }
    float mx = vm[0];
#pragma unroll
    for (int k = 1; k < GB_TJ; ++k) mx = fmaxf(mx, vm[k]);
    float z = 0.f, l = 0.f, e[GB_TJ];
#pragma unroll
    for (int k = 0; k < GB_TJ; ++k) { e[k] = __expf(vm[k] - mx); z += vs[k] * e[k]; if (LOSS) l += vl[k] * e[k]; }
    M = mx; coef = base / z;
    if (LOSS) lrow = 2.f * coef * l;                // sum_j softmax_ij * nl_ij * w_i / B
    if (fac) {
#pragma unroll
        for (int k = 0; k < GB_TJ; ++k) if (k < tj) fac[k] = e[k] * coef;
    }
}
#define GB_TJP 17                      // row stride of the LDS factor table (odd: conflict-free column reads)

// ISGA: the product is a compile-time constant of the instance (the workgroup picks it below): the prologue of each is
// straight-line code - with `isGA` a run-time value every request sat behind a branch join, and a join in front of a wait
// makes the wait cover everything requested so far (profiles/r02_waitcnt_fix.txt)
// DENSE (round 3): both operands are dense arrays (negative rows: the per-step copy Bn written by the edge-forward half of the
// first launch) - row numbers are arithmetic, so there is no index table, no LDS and NO BARRIER: the first operand requests leave
// with the wavefront's first instructions instead of after an id round + LDS round (never with FACT: its factor table lives in LDS)
// KS (round 3): wavefronts per tile along the reduction.  KS = 2: a workgroup = 8 wavefronts = 4 tiles x 2 halves of the macro
// steps (the second half also takes the tail step); two wavefronts per SIMD overlap each other's operand latency - with one
// wavefront per tile the MFMA pipe of a SIMD was busy 35 % of the wavefront's life; the second half hands its accumulators over
// through LDS (17 floats per lane) and the first half runs the epilogue.  Sums: first half + second half, a fixed order.
template <bool L2, bool FACT, bool ISGA, bool DENSE = false, int KS = 1, int EW = 0>   // FACT: the streamed weights are u_ij and get the per-(row, tile) factor here
__device__ __forceinline__ void neg_bwd_gemm_tile(const GemmArgs &a, int ti, int tj, int td, int c, int bcl, int maxK,
                                                  float *smem) {
    const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & (KGE_WAVES_PER_BLOCK - 1);
    const int kh = KS == 1 ? 0 : (int)(threadIdx.x >> 6) / KGE_WAVES_PER_BLOCK;     // which part of the reduction (wave-uniform)
    constexpr bool isGA = ISGA;
    const int tl0 = bcl * KGE_WAVES_PER_BLOCK + wv;      // 4 consecutive tiles of the same (chunk, product) per workgroup
    const int tr = isGA ? ti : tj;                       // row tiles of this product
    // (a wavefront without a tile repeats the last one and stores nothing: KS > 1 needs every wavefront at the hand-over barrier)
    // EW == 2 (ComplEx, GA tiles): a tile = 32 real columns + the 32 imaginary columns of the SAME complex elements (lanes m < 8 /
    // m >= 8 of every 16-lane row), so that the epilogue's chain through a = x o r finds both halves inside the tile
    constexpr bool PAIRED = EW >= 2 && ISGA;             // 2: ComplEx, 3: SimplE (first / second half of the row instead of re / im)
    if (PAIRED) td = (a.D / 2 + 31) / 32;
    const bool tile_ok = tl0 < tr * td;
    const int tl = KS == 1 ? tl0 : min(tl0, tr * td - 1);
    const int dt = tl % td, rt = tl / td;                // (row tile fastest - 4 wavefronts sharing a column slab - measured the same)
    const int D = a.D, N = a.N, chunk = a.chunk;
    const int m = lane & 15, q = lane >> 4;
    const int K = isGA ? N : chunk;                      // reduction length
    const int R = isGA ? chunk : N;                      // output rows per chunk

    // ---- prologue: every request is issued as early as its address is known, oldest first where something waits for it ----
    const int Kc = a.D / 2;                              // (PAIRED) complex elements per row
    const int dre = dt * 32 + (m & 7) * 4;               // (PAIRED) this lane's 4 complex elements
    const int d = PAIRED ? (m < 8 ? dre : Kc + dre) : dt * 64 + m * 4;       // this lane's 4 output columns
    const bool dok = PAIRED ? dre < Kc : d < D;          // D % 4 == 0 (PAIRED: Kc % 4 == 0): all-or-nothing
    const int dc = dok ? d : 0;
    // workgroup-shared table in LDS: rix[k] = row index of reduction element k in the streamed operand (GA: negative row,
    // gathered through neg_ids or dense; GN: positive row of the chunk).  INDICES, not pointers: a pointer read back from
    // LDS turns the row loads into flat loads, which count on lgkmcnt and serialise behind every LDS wait (measured: 48 %
    // of the wavefront time parked in s_waitcnt).
    // FACT, GN: ftab[k][t] = factor f(k, t) of every positive row k of the chunk and every column tile t, computed by
    // thread k in the same pass as the row's statistics (one dependent round; the wavefront whose output rows are
    // column tile rt reads column rt).
    int64_t *rix = reinterpret_cast<int64_t *>(smem);                     // [maxK]
    float *ftab = smem + 2 * maxK;                                        // [maxK][GB_TJP]
    // (1) the index this thread contributes to rix (GA, gathered negatives: a global load - the OLDEST request, the LDS
    //     write below waits for it alone)
    const int k0 = threadIdx.x;
    const int64_t rbase = isGA ? (int64_t)c * N : (int64_t)c * chunk;     // DENSE: reduction element k is row rbase + k
    int64_t rix0 = 0;
    if (!DENSE) {
        if (isGA) rix0 = a.nidx ? a.nidx[(int64_t)c * N + min(k0, K - 1)] : (int64_t)c * N + k0;
        else rix0 = (int64_t)c * chunk + k0;
    }
    // (2) the output rows' own vectors (rank-1 term of the L2 expansion / regulariser; at the end they would be one more
    //     dependent round) and the P rows of the Q epilogue.  GA: addresses are arithmetic - requested now, unconditionally
    //     (no Q wanted: a valid dummy address).  GN: the row numbers are gathered - index loads now, rows after the barrier.
    const bool need_self = L2 || ((!isGA) && a.reg_coef > 0.f && a.reg_norm > 0);
    const bool wantQ = isGA && a.Q != nullptr;
    float4 selfv[4], pq[4];
    int64_t srow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { selfv[r] = zero4(); pq[r] = zero4(); srow[r] = (int64_t)c * R + min(rt * 16 + q * 4 + r, R - 1); }
    // EW (DistMult, GA tiles): ids of the output rows' edge ends and relations - requested now, the row slices behind the barrier
    int64_t ewh[4], ewt[4], ewr[4];
    float ewdp[4];
    if constexpr (EW && ISGA) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ewh[r] = a.ew_h[srow[r]]; ewt[r] = a.ew_t[srow[r]]; ewr[r] = a.ew_r[srow[r]];
            ewdp[r] = a.ew_dpos ? a.ew_dpos[srow[r]] : 0.f;
        }
    }
    if (isGA) {
        if (L2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) selfv[r] = ldg4(a.A + srow[r] * D + dc);
        }
        const float *qp = (wantQ ? a.QP : a.A) + dc;
#pragma unroll
        for (int r = 0; r < 4; ++r) pq[r] = ldg4(qp + srow[r] * D);
    } else if (!DENSE && need_self && a.nidx) {
#pragma unroll
        for (int r = 0; r < 4; ++r) srow[r] = a.nidx[srow[r]];
    }
    // (3) the index table
    if (!DENSE) {
        if (k0 < K) rix[k0] = rix0;
        for (int k = k0 + KS * KGE_BLOCK; k < K; k += KS * KGE_BLOCK)         // K > the workgroup's threads
            rix[k] = isGA ? (a.nidx ? a.nidx[(int64_t)c * N + k] : (int64_t)c * N + k) : (int64_t)c * chunk + k;
    }
    if (FACT && !isGA) {
        for (int k = threadIdx.x; k < K; k += KS * KGE_BLOCK) {
            float M, coef, lr;
            row_stats<false>(a, (int64_t)c * chunk + k, tj, M, coef, lr, ftab + k * GB_TJP);
        }
    }
    // FACT, GA: this lane's row statistics (lane m <-> output row rt*16 + m), requested BEFORE the barrier so that the
    // partial loads fly together with the index loads above; the d-tile-0 wavefronts also reduce the row's loss term
    float rM = 0.f, rcoef = 0.f;
    const float *PMrow = FACT ? a.PM : nullptr;      // GN workgroups: any valid address (their loop loads it unconditionally, unused)
    if (FACT && isGA) {
        const int rowg = min(rt * 16 + m, chunk - 1);
        const int64_t gi = (int64_t)c * chunk + rowg;
        const bool do_loss = dt == 0 && kh == 0 && (a.row_neg || a.acc);
        float lrow = 0.f;
        if (do_loss) row_stats<true>(a, gi, tj, rM, rcoef, lrow);
        else row_stats<false>(a, gi, tj, rM, rcoef, lrow);
        PMrow = a.PM + gi * tj;
        if (do_loss && q == 0 && tile_ok && rt * 16 + m < chunk) {
            if (a.row_neg) a.row_neg[gi] = lrow;
            if (a.acc) {
                const int slot = (int)(gi & (KGE_ACC_SLOTS - 1));
                atomicAdd(a.acc + 1 * KGE_ACC_SLOTS + slot, lrow);          // fire-and-forget; one add per slot and
                atomicAdd(a.acc + 2 * KGE_ACC_SLOTS + slot, 0.5f * lrow);   // kernel when B <= slots: deterministic
            }
        }
    }
    if (!DENSE) __syncthreads();
    if (KS == 1 && !tile_ok) return;
    const float *ft = ftab + rt;                         // GN: factor of reduction row k = ft[k * GB_TJP]

    const float *Wc = a.W + (int64_t)c * chunk * N;
    const float *Ac = a.A + (int64_t)c * chunk * D;
    const int row = rt * 16 + m;
    const bool rok = row < R;
    const int rowc = min(row, R - 1);
    const bool vecW = isGA && (N % 4 == 0);
    const float *Wrow = isGA ? Wc + (int64_t)rowc * N : Wc + rowc;
    const int64_t wstride = isGA ? 1 : N;
    f32x4 acc[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) acc[s_] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f;                                    // partial row (GA) / column (GN) sum of W
    BwdStage s0[BU], s1[BU];

    // Main loop over FULL macro steps: no per-lane predicates (rows / columns are clamped so every
    // load is in bounds; garbage in clamped rows only reaches outputs that are never stored), row
    // addresses = kernel-argument base + LDS index (global loads, vmcnt only).  One predicated tail
    // step handles K % 16.
#ifdef BWD_PROBE_NOLOOP            // tuning probe (wrong results): wavefront time without the main loop
    const int msall = 0;
#else
    const int msall = K >> 4;
#endif
    // this wavefront's full macro steps [mlo, msfull): all of them (KS = 1), or the first / second half
    const int mlo = KS == 1 ? 0 : kh * (msall / KS);
    const int msfull = (KS == 1 || kh == KS - 1) ? msall : (kh + 1) * (msall / KS);
    const float *Xb = (isGA ? a.nbase : a.A) + dc;                 // operand base (global address space)
    const float *Wq = Wrow + (int64_t)(q * 4) * wstride;

    // loads are UNCONDITIONAL (macro step clamped; see the forward kernel: a load under a branch defeats the double buffering),
    // the W-layout choice is a compile-time constant of the loop instance (VECW), not a branch per load
#define BWD_LOAD(ST, MS0, VECW)                                                                \
    _Pragma("unroll") for (int u = 0; u < BU; ++u) {                                           \
        const int ms = min((MS0) + u, msfull - 1);                                             \
        int64_t ri[4];                                                                         \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                          \
            ri[e] = DENSE ? rbase + (ms * 16 + q * 4 + e) : rix[ms * 16 + q * 4 + e];          \
        if (VECW) {                                                                            \
            const float4 t4 = ldg4(Wq + ms * 16);                                              \
            ST[u].w[0] = t4.x; ST[u].w[1] = t4.y; ST[u].w[2] = t4.z; ST[u].w[3] = t4.w;        \
        } else {                                                                               \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                      \
                ST[u].w[e] = Wq[(int64_t)(ms * 16 + e) * wstride];                             \
        }                                                                                      \
        if (FACT) ST[u].pm = PMrow[isGA ? ms : 0];    /* macro step ms = column tile ms */     \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) ST[u].r[e] = ldg4(Xb + ri[e] * D);       \
    }
    // u_ij -> dL/dn_ij: one factor per (row, macro step) for GA, one LDS read of four per-row factors for GN
#define BWD_XFORM(ST, MS0)                                                                     \
    if (FACT) {                                                                                \
        _Pragma("unroll") for (int u = 0; u < BU; ++u) {                                       \
            const int ms = min((MS0) + u, msfull - 1);                                         \
            if (isGA) {                                                                        \
                const float f_ = a.lp.adv ? __expf(ST[u].pm - rM) * rcoef : rcoef;             \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) ST[u].w[e] *= f_;               \
            } else {                                                                           \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                  \
                    ST[u].w[e] *= ft[(ms * 16 + q * 4 + e) * GB_TJP];                          \
            }                                                                                  \
        }                                                                                      \
    }
#define BWD_MMA1(ST, u)                                                                        \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
        const float wgt = ST[u].w[e];                                                          \
        wsum += wgt;                                                                           \
        acc[0] = MFMA16(wgt, ST[u].r[e].x, acc[0]);                                            \
        acc[1] = MFMA16(wgt, ST[u].r[e].y, acc[1]);                                            \
        acc[2] = MFMA16(wgt, ST[u].r[e].z, acc[2]);                                            \
        acc[3] = MFMA16(wgt, ST[u].r[e].w, acc[3]);                                            \
    }
#define BWD_MMA(ST) _Pragma("unroll") for (int u = 0; u < BU; ++u) { BWD_MMA1(ST, u) }
#define BWD_MMA_G(ST, MS0) _Pragma("unroll") for (int u = 0; u < BU; ++u) { if ((MS0) + u < msfull) { BWD_MMA1(ST, u) } }
#define BWD_PIPE(VECW)                                                                         \
    {                                                                                          \
        BWD_LOAD(s0, mlo, VECW);                                                               \
        int g = mlo;                                                                           \
        for (; g + 2 * BU <= msfull; g += 2 * BU) {                                            \
            BWD_LOAD(s1, g + BU, VECW);                                                        \
            KGE_ORDER();                           /* requests first, then the MFMAs on the other buffer */ \
            BWD_XFORM(s0, g);                                                                  \
            BWD_MMA(s0);                                                                       \
            KGE_ORDER();                                                                       \
            BWD_LOAD(s0, g + 2 * BU, VECW);                                                    \
            KGE_ORDER();                                                                       \
            BWD_XFORM(s1, g + BU);                                                             \
            BWD_MMA(s1);                                                                       \
            KGE_ORDER();                                                                       \
        }                                                                                      \
        if (g < msfull) {                    /* fewer than 2 * BU macro steps left */          \
            BWD_LOAD(s1, g + BU, VECW);                                                        \
            BWD_XFORM(s0, g);                                                                  \
            BWD_MMA_G(s0, g);                                                                  \
            BWD_XFORM(s1, g + BU);                                                             \
            BWD_MMA_G(s1, g + BU);                                                             \
        }                                                                                      \
    }

    // tail macro step (K % 16 reduction indices; K = 200 at the FB15k configs): its operands are requested NOW, before the
    // main loop, so that they arrive under it - fetched after the loop they were one more dependent load round (~1 us) at
    // the end of every wavefront.  Reduction indices beyond K get zero weight (clamped, in-bounds loads).
    const bool has_tail = (K & 15) != 0;
    float tw[4];
    float4 tx[4];
    float tpm = 0.f;
    {   // (requested unconditionally - indices clamped, zero weight beyond K - so that the waits after this point count exactly)
        const int kk = msall * 16 + q * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kc = min(kk + e, K - 1);
            tw[e] = Wrow[(int64_t)kc * wstride];
            tx[e] = ldg4(Xb + (DENSE ? rbase + kc : rix[kc]) * D);
        }
        if (FACT && isGA) tpm = PMrow[min(msall, tj - 1)];
    }
    // GN: the own rows, now that their (gathered) numbers have arrived under the barrier
    if (!isGA && need_self) {
#pragma unroll
        for (int r = 0; r < 4; ++r) selfv[r] = ldg4(a.nbase + srow[r] * D + dc);
    }
    float4 ehv[4], etv[4], erv[4];
    if constexpr (EW && ISGA) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ehv[r] = ldg4(a.ew_ent + ewh[r] * D + dc); etv[r] = ldg4(a.ew_ent + ewt[r] * D + dc);
            erv[r] = ldg4(a.ew_rel + ewr[r] * D + dc);
        }
    }
#ifdef KGE_TL_MARKS
    KGE_TL_MARK_K(3, 0);           // prologue done: ids, own rows, first operands and the tail operands have arrived
#endif
    if (msfull > mlo) {
        if (isGA && vecW) BWD_PIPE(true) else BWD_PIPE(false)
    }
#ifdef KGE_TL_MARKS
    KGE_TL_MARK_K(3, 1);           // main loop done
#endif
#undef BWD_LOAD
#undef BWD_XFORM
#undef BWD_MMA1
#undef BWD_MMA
#undef BWD_MMA_G
#undef BWD_PIPE
    if (has_tail && kh == KS - 1) {
        const int kk = msall * 16 + q * 4;
        float ftail = 1.f;
        if (FACT && isGA) ftail = a.lp.adv ? __expf(tpm - rM) * rcoef : rcoef;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float wgt = tw[e];
            if (FACT) wgt *= isGA ? ftail : ft[min(kk + e, K - 1) * GB_TJP];
            wgt = (kk + e < K) ? wgt : 0.f;
            wsum += wgt;
            acc[0] = MFMA16(wgt, tx[e].x, acc[0]);
            acc[1] = MFMA16(wgt, tx[e].y, acc[1]);
            acc[2] = MFMA16(wgt, tx[e].z, acc[2]);
            acc[3] = MFMA16(wgt, tx[e].w, acc[3]);
        }
    }

#ifdef KGE_TL_MARKS
    KGE_TL_MARK_K(3, 2);           // tail macro step issued
#endif
    if constexpr (KS > 1) {
        // hand-over: the later parts park accumulators + weight sum in LDS ([tile][value][lane]: conflict-free), the first adds
        __shared__ float kred[(KS - 1) * KGE_WAVES_PER_BLOCK * 17 * 64];
        float *rb = kred + ((max(kh, 1) - 1) * KGE_WAVES_PER_BLOCK + wv) * 17 * 64 + lane;
        if (kh > 0) {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int r = 0; r < 4; ++r) rb[(s_ * 4 + r) * 64] = acc[s_][r];
            rb[16 * 64] = wsum;
        }
        __syncthreads();
        if (kh > 0) return;
#pragma unroll
        for (int p = 0; p < KS - 1; ++p) {
            const float *pb = kred + (p * KGE_WAVES_PER_BLOCK + wv) * 17 * 64 + lane;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[s_][r] += pb[(s_ * 4 + r) * 64];
            wsum += pb[16 * 64];
        }
        if (!tile_ok) return;
    }
    // lanes with equal (lane&15) hold partial sums of the same W row/column: combine the 4 groups
    wsum += __shfl_xor(wsum, 16, 64);
    wsum += __shfl_xor(wsum, 32, 64);
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
                    if (a.reg_norm == 3) {      // (wave-uniform branch: the default norm without the two transcendentals per element)
                        o.x += reg_grad3(sv.x, a.reg_coef); o.y += reg_grad3(sv.y, a.reg_coef);
                        o.z += reg_grad3(sv.z, a.reg_coef); o.w += reg_grad3(sv.w, a.reg_coef);
                    } else {
                        o.x += reg_grad(sv.x, a.reg_coef, a.reg_norm);
                        o.y += reg_grad(sv.y, a.reg_coef, a.reg_norm);
                        o.z += reg_grad(sv.z, a.reg_coef, a.reg_norm);
                        o.w += reg_grad(sv.w, a.reg_coef, a.reg_norm);
                    }
                }
            }
            // write-through store: GA / GN are consumed by the update kernel (any XCD); lines left dirty in this XCD's L2 only
            // lengthen the write-back before the next launch (profiles/r02_store_policy.txt)
            if (O) { Pack<4> ov; ov.v[0] = o.x; ov.v[1] = o.y; ov.v[2] = o.z; ov.v[3] = o.w; st_wt<4>(O + ((int64_t)c * R + ro) * D + d, ov); }
            if constexpr (EW >= 2 && ISGA) {
                // per-edge gradient rows of ComplEx (kge_rowwise.hip edge_bwd_body, same expressions): this lane holds the real
                // (m < 8) or the imaginary (m >= 8) part of 4 complex elements, lane m ^ 8 the other part
                auto other = [](float v) {       // DPP row_ror:8 - the value of lane m ^ 8 of the same 16-lane row
                    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, true));
                };
                const bool isim = m >= 8;
                const float dp = ewdp[r];
                const float go[4] = {o.x, o.y, o.z, o.w};
                const float ho[4] = {ehv[r].x, ehv[r].y, ehv[r].z, ehv[r].w}, to[4] = {etv[r].x, etv[r].y, etv[r].z, etv[r].w};
                const float ro_[4] = {erv[r].x, erv[r].y, erv[r].z, erv[r].w};
                Pack<4> gh, gt, gr;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gp = other(go[e]), hp_ = other(ho[e]), tp_ = other(to[e]), rp = other(ro_[e]);
                    const float a_rh = isim ? hp_ : ho[e], a_ih = isim ? ho[e] : hp_;
                    const float a_rt = isim ? tp_ : to[e], a_it = isim ? to[e] : tp_;
                    const float cc = isim ? rp : ro_[e], ss = isim ? ro_[e] : rp;
                    const float gr_ = isim ? gp : go[e], gi_ = isim ? go[e] : gp;
                    float v_rh, v_ih, v_rt, v_it, v_rr, v_ir;
                    if constexpr (EW == 3) {
                        // SimplE: (h_i, h_j) = (a_rh, a_ih), (t_i, t_j) = (a_rt, a_it), (rel, rel_inv) = (cc, ss); the 1/2 of the score
                        const float g1 = 0.5f * gr_, g2 = 0.5f * gi_, dh = 0.5f * dp;
                        v_rh = dh * cc * a_it; v_ih = dh * a_rt * ss; v_rt = dh * ss * a_ih; v_it = dh * a_rh * cc;
                        v_rr = dh * a_rh * a_it; v_ir = dh * a_rt * a_ih;
                        if (a.ew_neg_head) { v_it += g1 * cc; v_rr += g1 * a_it; v_rt += g2 * ss; v_ir += g2 * a_rt; }
                        else               { v_ih += g1 * ss; v_ir += g1 * a_ih; v_rh += g2 * cc; v_rr += g2 * a_rh; }
                    } else {
                    v_rh = dp * (a_rt * cc + a_it * ss);
                    v_ih = dp * (a_it * cc - a_rt * ss);
                    v_rt = dp * (a_rh * cc - a_ih * ss);
                    v_it = dp * (a_ih * cc + a_rh * ss);
                    v_rr = dp * (a_rh * a_rt + a_ih * a_it);
                    v_ir = dp * (a_rh * a_it - a_ih * a_rt);
                    if (a.ew_neg_head) {   // a = t o conj(r)
                        v_rt += gr_ * cc - gi_ * ss;
                        v_it += gr_ * ss + gi_ * cc;
                        v_rr += gr_ * a_rt + gi_ * a_it;
                        v_ir += gr_ * a_it - gi_ * a_rt;
                    } else {               // a = h o r
                        v_rh += gr_ * cc + gi_ * ss;
                        v_ih += -gr_ * ss + gi_ * cc;
                        v_rr += gr_ * a_rh + gi_ * a_ih;
                        v_ir += -gr_ * a_ih + gi_ * a_rh;
                    }
                    }
                    if (a.ew_reg_coef > 0.f && a.ew_reg_norm > 0) {
                        if (a.ew_reg_norm == 3) { v_rr += reg_grad3(cc, a.ew_reg_coef); v_ir += reg_grad3(ss, a.ew_reg_coef); }
                        else { v_rr += reg_grad(cc, a.ew_reg_coef, a.ew_reg_norm); v_ir += reg_grad(ss, a.ew_reg_coef, a.ew_reg_norm); }
                    }
                    gh.v[e] = isim ? v_ih : v_rh; gt.v[e] = isim ? v_it : v_rt; gr.v[e] = isim ? v_ir : v_rr;
                }
                const int64_t eo = ((int64_t)c * R + ro) * D + d;
                if (a.ew_GH) st_wt<4>(a.ew_GH + eo, gh);
                if (a.ew_GT) st_wt<4>(a.ew_GT + eo, gt);
                st_wt<4>(a.ew_GR + eo, gr);
            }
            if constexpr (EW == 1 && ISGA) {
                // per-edge gradient rows of DistMult (kge_rowwise.hip edge_bwd_body, same expressions)
                const float dp = ewdp[r];
                const float gx[4] = {o.x, o.y, o.z, o.w};
                const float hh[4] = {ehv[r].x, ehv[r].y, ehv[r].z, ehv[r].w}, tt[4] = {etv[r].x, etv[r].y, etv[r].z, etv[r].w};
                const float rr[4] = {erv[r].x, erv[r].y, erv[r].z, erv[r].w};
                Pack<4> gh, gt, gr;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float o_h = dp * rr[e] * tt[e], o_r = dp * hh[e] * tt[e], o_t = dp * hh[e] * rr[e];
                    if (a.ew_neg_head) { o_t += gx[e] * rr[e]; o_r += gx[e] * tt[e]; }
                    else               { o_h += gx[e] * rr[e]; o_r += gx[e] * hh[e]; }
                    if (a.ew_reg_coef > 0.f && a.ew_reg_norm > 0) {
                        if (a.ew_reg_norm == 3) o_r += reg_grad3(rr[e], a.ew_reg_coef);
                        else o_r += reg_grad(rr[e], a.ew_reg_coef, a.ew_reg_norm);
                    }
                    gh.v[e] = o_h; gt.v[e] = o_t; gr.v[e] = o_r;
                }
                const int64_t eo = ((int64_t)c * R + ro) * D + d;
                if (a.ew_GH) st_wt<4>(a.ew_GH + eo, gh);
                if (a.ew_GT) st_wt<4>(a.ew_GT + eo, gt);
                st_wt<4>(a.ew_GR + eo, gr);
            }
            if (wantQ) {
                Pack<4> qv;
                qv.v[0] = o.x + a.qc * pq[r].x; qv.v[1] = o.y + a.qc * pq[r].y;
                qv.v[2] = o.z + a.qc * pq[r].z; qv.v[3] = o.w + a.qc * pq[r].w;
                st_wt<4>(a.Q + ((int64_t)c * R + ro) * D + d, qv);
            }
        }
    }
}

// `bid` / `nblk`: this workgroup's index and the number of workgroups doing GEMM work (the body is also one half of the
// horizontally fused launch below); workgroup -> (chunk, product, 4 consecutive tiles)
template <bool L2, bool FACT, bool DENSE = false, int KS = 1, int EW = 0>
__device__ __forceinline__ void neg_bwd_gemm_body(const GemmArgs &a, int ti, int tj, int td, int bpA, int bpN, int maxK,
                                                  int bid, int nblk, float *smem) {
    const int blk = xcd_remap(bid, nblk);
    const int c = blk / (bpA + bpN);
    const int bc = blk % (bpA + bpN);
    if (bc < bpA) neg_bwd_gemm_tile<L2, FACT, true, DENSE, KS, EW>(a, ti, tj, td, c, bc, maxK, smem);
    else neg_bwd_gemm_tile<L2, FACT, false, DENSE, KS, 0>(a, ti, tj, td, c, bc - bpA, maxK, smem);
}

// measured (MI355X, us/step, GB_KS 1 -> 2): cfg-T 30.65 -> 31.49, DistMult 35.7 -> 35.7, ComplEx wikikg2 38.4 -> 38.4, SimplE 57.7 ->
// 58.2 - the second wavefront per SIMD repeats the prologue (ids, own rows, first operands) and adds a barrier + LDS hand-over;
// what bounds the launch is not the overlap inside the k-loop.  Kept as a compile-time option, off.
#ifndef GB_KS
#define GB_KS 1                       // wavefronts per backward tile along the reduction (stand-alone launch)
#endif
template <bool L2, bool FACT, bool DENSE = false, int EW = 0>
__global__ __launch_bounds__(GB_KS * KGE_BLOCK) void neg_bwd_gemm_kernel(GemmArgs a, int ti, int tj, int td,
                                                                         int bpA, int bpN, int maxK, int nbM, SmpTail st) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // (round 5) phase 2 of the sampler tail: the grid's FIRST 8 workgroups when the launch carries one (its chain - counts, keys, a
    // two-pass radix sort, totals - is as long as the tiles': it must not also wait for 230 workgroups to be dispatched; 8 keeps the
    // tiles' block id -> XCD mapping)
    const int toff = st.phase ? 8 : 0;
    if ((int)blockIdx.x < toff) {
        if constexpr (GB_KS == 1) sampler_tail_p2(st, (int)blockIdx.x);
        return;
    }
    KGE_TL(3);
    neg_bwd_gemm_body<L2, FACT, DENSE, GB_KS, EW>(a, ti, tj, td, bpA, bpN, maxK, (int)blockIdx.x - toff, nbM, smem);
}

// --async_update pipeline, horizontal fusion.  The pipeline (kge_step_async) keeps the exact one-step staleness of the
// reference's --async_update while only THREE launches per step sit on the critical path:
//     [ forward GEMM(s) || UPDATE(s-1) ]  ->  loss(s)  ->  [ backward GEMM(s) || PREP(s+1) ]
// Each fused launch = one grid whose first nbG workgroups run the GEMM tiles and whose remaining workgroups run the other
// job (different workspace halves / the tables; the two halves share nothing).  Stream order gives every dependency:
// UPDATE(s-1) after PREP(s) (previous launch) and before PREP(s+1) (next fused launch); PREP(s+1) therefore gathers rows
// with every update up to s-1 and without update s.  No second stream, no events - cross-stream dependencies inside a
// hipGraph cost 9-15 us of dispatch gaps per step on this stack (profiles/r02_async_pipeline.txt).
template <bool L2, int MODEL>
__global__ __launch_bounds__(KGE_BLOCK) void neg_bwd_prep_kernel(GemmArgs a, int ti, int tj, int td, int bpA, int bpN,
                                                                 int maxK, int nbG, EdgeFwdArgs e) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x < nbG) {
        KGE_TL(3);
        neg_bwd_gemm_body<L2, false, true>(a, ti, tj, td, bpA, bpN, maxK, (int)blockIdx.x, nbG, smem);   // the pipeline's operands are dense
    } else {
        KGE_TL(0);
        edge_fwd_body<MODEL, 4, false>(e, (int)blockIdx.x - nbG);
    }
}

int launch_neg_bwd_gemm(const GemmArgs &a, hipStream_t s, const SmpTail *tail) {
    if (a.C == 0) return tail && tail->phase ? KGE_ERR_ARG : KGE_OK;
    SmpTail st{};
    if (tail && tail->phase && GB_KS == 1) { st = *tail; st.phase = 2; }
    const int nbT = st.phase ? 8 : 0;            // (ST_P2_WGS <= 8 tail workgroups in front of the tiles, see the kernel)
    if (!a.W) return KGE_ERR_ARG;
    const bool fact = a.PM != nullptr;                           // W holds u_ij: factorised fused loss
    if (fact && (a.lp.pairwise || !a.PS || !a.PL)) return KGE_ERR_ARG;
    const int maxK = a.chunk > a.N ? a.chunk : a.N;
    if (maxK > GB_MAXK) return KGE_ERR_ARG;
    const int ti = (a.chunk + 15) / 16, tj = (a.N + 15) / 16, td = (a.D + 63) / 64;
    const int bpA = (ti * td + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK;   // workgroups per chunk, GA
    const int bpN = (tj * td + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK;
    const int nb = a.C * (bpA + bpN);
    const int mk = (maxK + 3) & ~3;
    if (fact && !neg_gemm_fused_loss_supported(a.chunk, a.N)) return KGE_ERR_ARG;
    // row indices [mk] int64 (+ factorised: the factor table [mk][GB_TJP])
    const size_t sm = (size_t)mk * 8 + (fact ? (size_t)mk * GB_TJP * 4 : 0);
    const bool l2 = a.model == KGE_TRANSE_L2;
    const dim3 g(nb + nbT), b(GB_KS * KGE_BLOCK);
    if (a.ew_GR) {                                               // DistMult / ComplEx: per-edge gradient rows from the GA tiles' epilogue
        const bool simple = a.model == KGE_SIMPLE, cplx = a.model == KGE_COMPLEX || simple;
        if (fact || l2 || (a.model != KGE_DISTMULT && !cplx) || !a.ew_ent || !a.ew_rel || !a.ew_h || !a.ew_t || !a.ew_r ||
            a.D % (cplx ? 8 : 4))
            return KGE_ERR_ARG;
        if (cplx) {                                              // GA tiles: 32 complex elements (real + imaginary columns) each
            const int tdA = (a.D / 2 + 31) / 32;
            const int bpAc = (ti * tdA + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK;
            const int nbc = a.C * (bpAc + bpN);
            const dim3 gc(nbc + nbT);
            if (simple) {
                if (!a.nidx) hipLaunchKernelGGL((neg_bwd_gemm_kernel<false, false, true, 3>), gc, b, 0, s, a, ti, tj, td, bpAc, bpN, mk, nbc, st);
                else hipLaunchKernelGGL((neg_bwd_gemm_kernel<false, false, false, 3>), gc, b, sm, s, a, ti, tj, td, bpAc, bpN, mk, nbc, st);
            } else if (!a.nidx) hipLaunchKernelGGL((neg_bwd_gemm_kernel<false, false, true, 2>), gc, b, 0, s, a, ti, tj, td, bpAc, bpN, mk, nbc, st);
            else hipLaunchKernelGGL((neg_bwd_gemm_kernel<false, false, false, 2>), gc, b, sm, s, a, ti, tj, td, bpAc, bpN, mk, nbc, st);
            return check_launch_g();
        }
        if (!a.nidx) hipLaunchKernelGGL((neg_bwd_gemm_kernel<false, false, true, 1>), g, b, 0, s, a, ti, tj, td, bpA, bpN, mk, nb, st);
        else hipLaunchKernelGGL((neg_bwd_gemm_kernel<false, false, false, 1>), g, b, sm, s, a, ti, tj, td, bpA, bpN, mk, nb, st);
        return check_launch_g();
    }
    if (!fact && !a.nidx) {                                      // dense operands: the instance without index table / LDS / barrier
        if (l2) hipLaunchKernelGGL((neg_bwd_gemm_kernel<true, false, true>), g, b, 0, s, a, ti, tj, td, bpA, bpN, mk, nb, st);
        else hipLaunchKernelGGL((neg_bwd_gemm_kernel<false, false, true>), g, b, 0, s, a, ti, tj, td, bpA, bpN, mk, nb, st);
        return check_launch_g();
    }
    if (l2 && fact) hipLaunchKernelGGL((neg_bwd_gemm_kernel<true, true>), g, b, sm, s, a, ti, tj, td, bpA, bpN, mk, nb, st);
    else if (l2) hipLaunchKernelGGL((neg_bwd_gemm_kernel<true, false>), g, b, sm, s, a, ti, tj, td, bpA, bpN, mk, nb, st);
    else if (fact) hipLaunchKernelGGL((neg_bwd_gemm_kernel<false, true>), g, b, sm, s, a, ti, tj, td, bpA, bpN, mk, nb, st);
    else hipLaunchKernelGGL((neg_bwd_gemm_kernel<false, false>), g, b, sm, s, a, ti, tj, td, bpA, bpN, mk, nb, st);
    return check_launch_g();
}

// forward GEMM of one step + Adagrad update of ANOTHER step in one launch (neg_fwd_update_kernel).
// Returns KGE_ERR_ARG when the combination has no fused instantiation (the caller then launches the two kernels apart).
int launch_neg_fwd_gemm_with_update(const GemmArgs &a, const UpdateArgs &u, hipStream_t s) {
    if (a.C == 0 || a.PM) return KGE_ERR_ARG;
    const int dmax = u.model_d_e > u.d_r ? u.model_d_e : u.d_r;
    const bool inplace = !u.emit_ent && !u.emit_rel && !u.g0 && !u.g1 && !u.gs0 && !u.gs1 && !u.gr && !u.gsr && !u.rid &&
                         !u.dry && !u.nd_chunk && u.em.n == 0 && u.rm.n == 0;
    if (!inplace || u.model_d_e % 4 || u.d_r % 4 || dmax > 1024) return KGE_ERR_ARG;
    const int nbE = (u.UE + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK, nbR = (u.UR + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK;
    if (nbE + nbR == 0) return KGE_ERR_ARG;
    const int ti = (a.chunk + 15) / 16, tj = (a.N + 15) / 16;
    const int64_t ntiles = (int64_t)a.C * ti * tj;
    const int nbG = (int)((ntiles + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK);
    const bool l2 = a.model == KGE_TRANSE_L2;
    const int nit = dmax <= 256 ? 1 : (dmax <= 512 ? 2 : 4);
    // (the TransE fast path behind this GEMM always comes with Q: update variants 3 = TransE with Q, 2 = per-edge gradients)
    if (u.transe_fast && !u.Q) return KGE_ERR_ARG;
    const int lean = u.transe_fast ? 3 : 2;
    const dim3 g(nbG + nbE + nbR), b(KGE_BLOCK);
#define KGE_FU(L2_, N_, LE_) hipLaunchKernelGGL((neg_fwd_update_kernel<L2_, N_, LE_>), g, b, 0, s, a, ti, tj, nbG, u, nbE)
#define KGE_FU_N(N_) do { if (l2) { if (lean == 3) KGE_FU(true, N_, 3); else KGE_FU(true, N_, 2); }           \
                          else { if (lean == 3) return KGE_ERR_ARG; KGE_FU(false, N_, 2); } } while (0)
    if (nit == 1) KGE_FU_N(1); else if (nit == 2) KGE_FU_N(2); else KGE_FU_N(4);
#undef KGE_FU_N
#undef KGE_FU
    return check_launch_g();
}

// backward GEMM of one step + PREP (edge forward) of the NEXT step in one launch (neg_bwd_prep_kernel).
// Returns KGE_ERR_ARG when the combination has no fused instantiation.
int launch_neg_bwd_gemm_with_prep(const GemmArgs &a, const EdgeFwdArgs &e, hipStream_t s) {
    if (a.C == 0 || !a.W || a.PM || a.nidx) return KGE_ERR_ARG;       // (dense negative rows: PREP's copy)
    const int maxK = a.chunk > a.N ? a.chunk : a.N;
    if (maxK > GB_MAXK) return KGE_ERR_ARG;
    const bool cx = kge::is_complex_model(e.model);
    const bool vec = cx ? ((e.d_e / 2) % 4 == 0 && e.d_r % 4 == 0) : (e.d_e % 4 == 0);
    if (!vec || e.src.em.n || e.src.rm.n) return KGE_ERR_ARG;
    const bool negjob = e.bsq || e.Bn;
    const int64_t waves = (int64_t)e.B + (negjob ? e.n_neg : 0);
    EdgeFwdArgs ee = e;
    if (!negjob) ee.n_neg = 0;
    const int nbP = (int)((waves + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK);
    if (nbP == 0) return KGE_ERR_ARG;
    const int ti = (a.chunk + 15) / 16, tj = (a.N + 15) / 16, td = (a.D + 63) / 64;
    const int bpA = (ti * td + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK;
    const int bpN = (tj * td + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK;
    const int nbG = a.C * (bpA + bpN);
    const int mk = (maxK + 3) & ~3;
    const size_t sm = 0;
    const dim3 g(nbG + nbP), b(KGE_BLOCK);
#define KGE_BP(L2_, M_) hipLaunchKernelGGL((neg_bwd_prep_kernel<L2_, M_>), g, b, sm, s, a, ti, tj, td, bpA, bpN, mk, nbG, ee)
    if (a.model == KGE_TRANSE_L2 && e.model == KGE_TRANSE_L2) KGE_BP(true, KGE_TRANSE_L2);
    else if (a.model == KGE_DISTMULT && e.model == KGE_DISTMULT) KGE_BP(false, KGE_DISTMULT);
    else if (a.model == KGE_COMPLEX && e.model == KGE_COMPLEX) KGE_BP(false, KGE_COMPLEX);
    else if (a.model == KGE_SIMPLE && e.model == KGE_SIMPLE) KGE_BP(false, KGE_SIMPLE);
    else return KGE_ERR_ARG;
#undef KGE_BP
    return check_launch_g();
}
