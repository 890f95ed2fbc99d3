// kge_common.hpp - shared device helpers for the gfx950 KGE kernels.
// Wavefront = 64 lanes everywhere in this library (CDNA4); one wavefront per embedding row is the
// unit of work of all row-wise kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/kge_hip.h"

#define KGE_WAVE 64
#define KGE_BLOCK 256              // 4 wavefronts per workgroup, one per SIMD
#define KGE_WAVES_PER_BLOCK (KGE_BLOCK / KGE_WAVE)

typedef float f32x4 __attribute__((ext_vector_type(4)));


// ---- developer timeline (build with -DKGE_TIMELINE; tools/timeline.py): every wavefront of the step's kernels
// records {kernel id | hardware id, start, end} (100 MHz wall clock) into a per-translation-unit device buffer
// at slot kid * KGE_TL_PER_KERNEL + its wavefront number.  Compiled out of the product library.
#ifdef KGE_TIMELINE
#define KGE_TL_PER_KERNEL 8192
static __device__ unsigned long long *kge_tl_buf = nullptr;
// phase mark: drains the memory counters first, so the time since the previous mark is the latency of whatever was
// outstanding (changes the overlap - use for attribution, not for the headline time); written straight into the wavefront's
// record, so that device functions below the kernel can set marks too
__device__ __forceinline__ void kge_tl_mark(int kid, int n) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const int wid = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if ((threadIdx.x & 63) == 0 && kge_tl_buf && wid < KGE_TL_PER_KERNEL)
        kge_tl_buf[((size_t)kid * KGE_TL_PER_KERNEL + wid) * 8 + 4 + n] = wall_clock64();
}
struct KgeTlScope {
    unsigned long long t0; int kid; int wid;
    __device__ __forceinline__ KgeTlScope(int kid_, int wid_) : kid(kid_), wid(wid_) { t0 = wall_clock64(); }
    __device__ __forceinline__ void mark(int n) { kge_tl_mark(kid, n); }
    __device__ __forceinline__ ~KgeTlScope() {
        if ((threadIdx.x & 63) == 0 && kge_tl_buf && wid < KGE_TL_PER_KERNEL) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long *r = kge_tl_buf + ((size_t)kid * KGE_TL_PER_KERNEL + wid) * 8;
            r[0] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
            r[1] = t0;
            r[2] = wall_clock64();
            r[3] = (unsigned long long)kid + 1;
        }
    }
};
#define KGE_TL(kid) KgeTlScope kge_tl_scope_((kid), (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)))
#define KGE_TL_MARK(n) kge_tl_scope_.mark(n)
#define KGE_TL_MARK_K(kid, n) kge_tl_mark((kid), (n))
#define KGE_TL_DEFINE(name) extern "C" int kge_tl_set_##name(void *p) { \
        return hipMemcpyToSymbol(HIP_SYMBOL(kge_tl_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#else
#define KGE_TL(kid)
#define KGE_TL_MARK(n)
#define KGE_TL_MARK_K(kid, n)
#define KGE_TL_DEFINE(name)
#endif

namespace kge {

// Wavefront reductions on DPP (data-parallel primitives: the cross-lane operand is part of the VALU
// instruction) + v_readlane, instead of __shfl_xor butterflies: a shuffle compiles to ds_bpermute_b32 - an
// LDS-crossbar round trip of >100 cycles - and a 64-lane butterfly is a dependent chain of six of them; with
// one wavefront per SIMD nothing hides it (measured: 18-30 ds_bpermute per row-wise kernel, ~0.3 us per
// reduction).  Steps: xor 1, xor 2 (quad_perm), 8-group (row_half_mirror), 16-lane row (row_mirror): every
// lane of a row then holds the row's result; the four rows are combined through SGPRs in a fixed order.
// Every lane returns the same value; the summation order is fixed (bit-reproducible).
#define KGE_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))
#define KGE_DPP_QUAD_X1 0xB1      // quad_perm [1,0,3,2]
#define KGE_DPP_QUAD_X2 0x4E      // quad_perm [2,3,0,1]
#define KGE_DPP_HALF_MIRROR 0x141
#define KGE_DPP_MIRROR 0x140
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float row_sum16(float v) {       // sum over the 16 lanes of each DPP row, in every lane
    v += KGE_DPP_F(v, KGE_DPP_QUAD_X1);
    v += KGE_DPP_F(v, KGE_DPP_QUAD_X2);
    v += KGE_DPP_F(v, KGE_DPP_HALF_MIRROR);
    v += KGE_DPP_F(v, KGE_DPP_MIRROR);
    return v;
}
__device__ __forceinline__ float row_max16(float v) {
    v = fmaxf(v, KGE_DPP_F(v, KGE_DPP_QUAD_X1));
    v = fmaxf(v, KGE_DPP_F(v, KGE_DPP_QUAD_X2));
    v = fmaxf(v, KGE_DPP_F(v, KGE_DPP_HALF_MIRROR));
    v = fmaxf(v, KGE_DPP_F(v, KGE_DPP_MIRROR));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row_sum16(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row_max16(v);
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

// row pointer with optional one-level indirection: base + (idx ? idx[i] : i) * ld
__device__ __forceinline__ const float *row_ptr(const float *base, const int64_t *idx, int64_t i,
                                                int ld) {
    const int64_t r = idx ? idx[i] : i;
    return base + r * (int64_t)ld;
}

// Range-sharded table mapped peer-to-peer over xGMI (sharded training, kge_step_sharded): row `id`
// lives in shard id / per at local row id % per; `rows` / `state` are DEVICE arrays of the n shard
// bases as mapped into THIS process (own shard: the local allocation, peers: hipIpcOpenMemHandle).
// n == 0 means "one local table" and every helper degenerates to base + id * ld.
struct ShardMap {
    float *const *rows;
    float *const *state;
    int64_t per;
    int n;
};
__device__ __forceinline__ float *shard_row(const ShardMap &m, float *base, int64_t id, int ld) {
    if (m.n == 0) return base + id * (int64_t)ld;
    const int64_t o = id / m.per;
    return m.rows[o] + (id - o * m.per) * (int64_t)ld;
}
__device__ __forceinline__ float *shard_state(const ShardMap &m, float *base, int64_t id) {
    if (m.n == 0) return base + id;
    const int64_t o = id / m.per;
    return m.state[o] + (id - o * m.per);
}
// row pointer through an id list into a (possibly sharded) table
__device__ __forceinline__ const float *table_row(const ShardMap &m, const float *base, const int64_t *idx,
                                                  int64_t i, int ld) {
    const int64_t r = idx ? idx[i] : i;
    return shard_row(m, const_cast<float *>(base), r, ld);
}

// V-wide load/store (V = 4: one 16-byte access; V = 1: scalar)
template <int V> struct Pack { float v[V]; };
template <int V> __device__ __forceinline__ Pack<V> ld(const float *p);
template <> __device__ __forceinline__ Pack<4> ld<4>(const float *p) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    Pack<4> r; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r;
}
template <> __device__ __forceinline__ Pack<1> ld<1>(const float *p) {
    Pack<1> r; r.v[0] = *p; return r;
}
template <int V> __device__ __forceinline__ void st(float *p, const Pack<V> &x);
template <> __device__ __forceinline__ void st<4>(float *p, const Pack<4> &x) {
    *reinterpret_cast<float4 *>(p) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
}
template <> __device__ __forceinline__ void st<1>(float *p, const Pack<1> &x) { *p = x.v[0]; }
// streaming store: the line is not kept dirty in this XCD's L2 (the consumer is another kernel, usually on another XCD,
// and every dirty line must be written back before the next kernel of the stream may start)
template <int V> __device__ __forceinline__ void st_nt(float *p, const Pack<V> &x);
template <> __device__ __forceinline__ void st_nt<4>(float *p, const Pack<4> &x) {
    f32x4 v = {x.v[0], x.v[1], x.v[2], x.v[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
}
template <> __device__ __forceinline__ void st_nt<1>(float *p, const Pack<1> &x) { __builtin_nontemporal_store(x.v[0], p); }
// write-through store (system-coherent cache policy, no streaming hint): the line goes to the memory side (Infinity Cache)
// at once instead of staying dirty in this XCD's L2 until the end-of-kernel write-back
template <int V> __device__ __forceinline__ void st_wt(float *p, const Pack<V> &x);
// (s_nop 1: a VMEM store of more than 64 bits reads its data registers for a few cycles after issue, and a VALU instruction that
//  overwrites them needs 2 wait states behind it on gfx940+.  The compiler's hazard recogniser inserts them for its own stores
//  but does not look inside inline assembly: without the nop, a following instruction that reused the data registers corrupted
//  x / y of the stored vector in the last lanes of every row - found when three of these stores ran back to back)
template <> __device__ __forceinline__ void st_wt<4>(float *p, const Pack<4> &x) {
    f32x4 v = {x.v[0], x.v[1], x.v[2], x.v[3]};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
template <> __device__ __forceinline__ void st_wt<1>(float *p, const Pack<1> &x) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(x.v[0]) : "memory");
}
// round 4: the pairwise models' intermediates that the NEXT launch consumes leave write-through (GN partials and GA parts of the
// shared-pair backward, the summed GN rows, the dense copy of the negative rows): lines left dirty in the XCDs' L2s are written
// back at the end of the kernel, in front of the next launch (RotatE: 3.1 us instead of 1.2 between the shared-pair backward and
// edge_bwd).  Measured on one box (profiles/r04_store_policy.txt): RotatE FB15k shape 60.7 -> 58.3 us/step, cfg-R per GPU 99.5 ->
// 97.5 (p2p) / 116.4 -> 114.9 (a2a); the a2a engine's gradient messages and gathered cache rows stay plain (write-through: +1.2 us).
// -DKGE_PLAIN_INTERMEDIATES: plain stores everywhere, -DKGE_PLAIN_NEXT: all but the partials (A/B aids)
#if defined(KGE_PLAIN_INTERMEDIATES) || defined(KGE_PLAIN_NEXT)
#define KGE_ST_NEXT kge::st
#else
#define KGE_ST_NEXT kge::st_wt
#endif
template <int V> __device__ __forceinline__ Pack<V> zero_pack() {
    Pack<V> r;
#pragma unroll
    for (int e = 0; e < V; ++e) r.v[e] = 0.f;
    return r;
}

__device__ __forceinline__ float sgnf(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// d/dx coef*|x|^q  (general_models.py:572-576; norm = x.norm(p)**p, tensor_models.py:54).
// Branch-free on purpose (hardware exp2 / log2, ~1e-6 relative): a chain of `if (q == 3) ... else powf` at every
// use is unswitched / inlined by the compiler into several copies of every loop and made the update kernel
// 11 k instructions - larger than the instruction cache (measured 15.6 -> 12.4 us just from removing powf).
// (q known at compile time to be 3 - the reference's default regularization_norm and every BASELINE recipe's; the LEAN kernel
//  instances fix it - : plain multiplies, exact; the general form costs two quarter-rate transcendentals per element, which at four
//  wavefronts per SIMD was ~1 us of the update kernel, round 4)
__device__ __forceinline__ float reg_grad3(float x, float coef) { return 3.f * coef * x * fabsf(x); }
__device__ __forceinline__ float reg_val3(float x) { const float ax = fabsf(x); return ax * ax * ax; }
__device__ __forceinline__ float reg_grad(float x, float coef, int q) {
    if (__builtin_constant_p(q) && q == 3) return reg_grad3(x, coef);
    const float ax = fabsf(x);
    const float pw = __builtin_amdgcn_exp2f((float)(q - 1) * __builtin_amdgcn_logf(ax));     // |x|^(q-1)
    return ax > 0.f ? copysignf(coef * (float)q * pw, x) : 0.f;
}
__device__ __forceinline__ float reg_val(float x, int q) {
    if (__builtin_constant_p(q) && q == 3) return reg_val3(x);
    const float ax = fabsf(x);
    return ax > 0.f ? __builtin_amdgcn_exp2f((float)q * __builtin_amdgcn_logf(ax)) : 0.f;    // |x|^q
}

// rv + |x|^q with the rounding spelled out for q == 3: left as `rv += ax * ax * ax`, two instances of the update kernel contracted
// the last multiply into the addition differently and the regularisation VALUE (a reported sum, not part of the gradients) of a
// row came out 1 ulp apart between them (found when a different GA part count moved the data onto a rounding boundary, round 4)
__device__ __forceinline__ float reg_acc(float rv, float x, int q) {
    if (__builtin_constant_p(q) && q == 3) { const float ax = fabsf(x); return fmaf(ax * ax, ax, rv); }
    return rv + reg_val(x, q);
}

__device__ __forceinline__ float sigmoidf_(float x) {
    // numerically safe logistic
    if (x >= 0.f) { return 1.f / (1.f + expf(-x)); }
    const float e = expf(x);
    return e / (1.f + e);
}
// -logsigmoid(z) = softplus(-z) = max(-z,0) + log1p(exp(-|z|))
__device__ __forceinline__ float neg_logsigmoid(float z) {
    return fmaxf(-z, 0.f) + log1pf(expf(-fabsf(z)));
}

// models whose rows are two halves ([re | im], SimplE: [x_i | x_j]) processed element-pair-wise
constexpr bool is_complex_model(int m) { return m == KGE_COMPLEX || m == KGE_ROTATE || m == KGE_SIMPLE; }
#define KGE_SIMPLE_CLAMP 20.f        // th.clamp(score, -20, 20), score_fun.py:568

// fast-math variants (hardware exp/log, |rel err| ~1e-6) for the hot kernels
__device__ __forceinline__ float fast_sigmoid(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_neg_logsigmoid(float z) {
    return fmaxf(-z, 0.f) + __logf(1.f + __expf(-fabsf(z)));
}
__device__ __forceinline__ void criterion_fast(int genre, float s, float label, float margin,
                                               float &val, float &dval) {
    if (genre == KGE_LOSS_HINGE) {
        const float v = margin - label * s;
        val = v < 0.f ? 0.f : v;
        dval = v < 0.f ? 0.f : -label;
    } else if (genre == KGE_LOSS_BCE) {
        val = label * fast_neg_logsigmoid(s) + (1.f - label) * fast_neg_logsigmoid(-s);
        dval = fast_sigmoid(s) - label;
    } else {
        const float z = label * s;
        val = fast_neg_logsigmoid(z);
        dval = -label * fast_sigmoid(-z);
    }
}

// criterion value and derivative w.r.t. the score for label l (models/pytorch/loss.py:10-38): the hardware
// exp/log version everywhere (rel. error ~1e-6, inside the 1e-5 loss tolerance) - the libm expansions of
// expf / log1pf are hundreds of instructions per call site
__device__ __forceinline__ void criterion(int genre, float s, float label, float margin,
                                          float &val, float &dval) {
    criterion_fast(genre, s, label, margin, val, dval);
}

// Edge importance and the POSITIVE loss part: the reference multiplies `pos_loss [B]` by `edge_weight.view(-1, 1) [B, 1]`
// (loss.py:75, 82) - a [B, B] broadcast whose mean is mean_j crit(p_j) * mean_i w_i: every positive edge is weighted by the MEAN
// importance of the batch, not by its own (the negative part, [B, N] * [B, 1], is weighted per edge).  Kept as it is (found in
// round 4 by the per-golden row tolerance: goldens/transe_l2_impts).  `pre` > 0: the mean computed ONCE per batch by the batch's
// builder (kge_batch.edge_w_mean, ABI 8); otherwise (modular ops, older callers) every wavefront that needs it sums the B weights
// itself (lane-strided, then the fixed-order wave reduction: deterministic).
__device__ __forceinline__ float mean_edge_weight(const float *w, int B, int lane, float pre = 0.f) {
    if (!w) return 1.f;
    if (pre > 0.f) return pre;
    float s = 0.f;
    for (int j = lane; j < B; j += KGE_WAVE) s += w[j];
    return wave_sum(s) / (float)B;
}

// bijective XCD-aware remap: hardware block b runs on XCD b%8; give the blocks of one XCD
// consecutive logical ids (L2 locality only, never correctness).
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

}  // namespace kge

// ------------------------------------------------------------------------------------------
// kernel launch entry points implemented in the .hip files (host side, all enqueue on `stream`)
// ------------------------------------------------------------------------------------------
struct EdgeSrc {           // where the rows of one positive edge come from
    const float *hbase; const int64_t *hidx;   // head rows  hbase + (hidx? hidx[i] : i)*d_e
    const float *tbase; const int64_t *tidx;   // tail rows
    const float *rbase; const int64_t *ridx;   // relation rows, ld = d_r
    kge::ShardMap em, rm;                      // entity / relation table sharding (n = 0: local tables)
};

struct LossParams {                 // LossGenerator configuration (loss.py:41-61)
    int genre, adv, pairwise;
    float adv_temp, margin;
};

struct EdgeFwdArgs {
    EdgeSrc src;
    int B, d_e, d_r, neg_head, model;
    float gamma, rot_div;            // rot_div = emb_init / pi (RotatE: phase = r / rot_div, score_fun.py:464)
    float *pos_score;                // [B] or null
    float *A;                        // [B,d_e] pos-side vectors or null
    float *asq;                      // [B] |a|^2 (TransE_l2 GEMM form) or null
    // extra row jobs folded into the same launch (fused step):
    const float *nbase; const int64_t *nidx; int n_neg;  // negative rows -> bsq
    float *bsq;                      // [n_neg] or null
    float *Bn;                       // [n_neg,d_e] dense copy of the negative rows or null
    // positive-loss part (pointwise losses): dpos_i = dL/dp_i needs only p_i
    int do_pos_loss; LossParams lp; const float *w; float w_mean;   // w_mean: kge_batch.edge_w_mean (0: summed by the kernel)
    float *dpos;                     // [B] or null
    float *row_pos;                  // [B] per-row positive loss terms or null
    float *acc;                      // running sums or null
    float *P;                        // [B,d_e] TransE only: dpos_i * d|u_i|/du_i (u = h+r-t) or null
    // --neg_deg_sample: the negative rows of chunk c are [the chunk's own corrupted-side entities | the sampled ids]
    // (general_models.py:396-400, 424-427): nd_own != null -> negative job j reads id nd_own[c*nd_chunk + jj] for
    // jj = j % (nd_chunk + nd_Ns) < nd_chunk, else nidx[c*nd_Ns + jj - nd_chunk] (no combined id list is materialised)
    const int64_t *nd_own; int nd_chunk, nd_Ns;
    float *Hc, *Tc, *Rc;             // [B,d_e], [B,d_e], [B,d_r] dense copies of the gathered h / t / r rows or null
                                     // (--async_update pipeline: later kernels of the step must not re-read the tables)
};

struct EdgeBwdArgs {
    EdgeSrc src;
    int B, d_e, d_r, neg_head, model;
    float gamma, rot_div;
    const float *dpos;               // [B] dL/dp or null (no positive-score part)
    const float *GA;                 // [B,d_e] dL/da or null (no negative-score part)
    float reg_coef; int reg_norm;    // regularisation gradient added to the relation row grad
    int clamp_pos;                   // SimplE, modular op: dpos is the gradient w.r.t. the CLAMPED score -> recompute
                                     // the raw score and drop it where saturated (the fused step masks in edge_fwd)
    float *GH, *GT;                  // [B,d_e] (either may be null)
    float *GR;                       // [B,d_r] or null
    // neg_deg_sample: gradient of the in-batch negative row of edge i, GNd[((i / nd_chunk) * nd_Np + i % nd_chunk) * d_e],
    // is added to the corrupted side's gradient (positive trace of that entity); null = off
    const float *GNd; int nd_chunk, nd_Np;
    // GA arrives as ga_parts (> 1) partial sums, ga_stride floats apart (shared-pair backward with the negatives split over
    // workgroups, neg_bwd_lc_splits): summed in part order while the row is read
    int ga_parts; int64_t ga_stride;
};

struct NegArgs {                     // chunked negative scoring, forward and backward
    int model, C, chunk, N, d_e;
    float gamma;
    const float *A;                  // [C*chunk, d_e] pos-side vectors (dense)
    const float *asq;                // [C*chunk] (L2 GEMM form) or null
    const float *nbase; const int64_t *nidx;   // negative rows nbase + (nidx? nidx[j] : j)*d_e
    const float *bsq;                // [C*N] or null
    float *S;                        // fwd: out [C,chunk,N]
    const float *W;                  // bwd: dL/dn (for L2: already divided by the distance)
    float *GA;                       // bwd out [C*chunk, d_e]
    float *GN;                       // bwd out [C*N, d_e]
    float reg_coef; int reg_norm;    // (unused by the GEMM/pair kernels; regularisation is added
                                     //  by the consumers of GN)
    float clampv;                    // > 0: clamp the scores to [-clampv, clampv] (SimplE)
    float *GNp;                      // bwd, TransE_l1 / RotatE: room for the per-row-group GN partials
                                     // (neg_bwd_lc_partial_floats) or null: two-pass kernels
    int defer_reduce;                // shared-pair backward: leave the partials unsummed (the caller's next launch sums them)
    // shared-pair backward, RotatE: the negatives of a chunk split over ga_parts (> 1) workgroups per (chunk, slab, row group) -
    // GA then leaves as ga_parts partial sums ga_stride floats apart and the CONSUMER adds them (edge_bwd: EdgeBwdArgs.ga_parts)
    int ga_parts; int64_t ga_stride;
    // balanced split (round 4, filled by the launcher; lc_P > 0): exactly 1024 workgroups = 4 per CU - the first lc_nB columns
    // (chunk, slab, row group) are cut into lc_P parts, the others into lc_P + 1, and the workgroups are dealt heaviest-first in
    // alternating direction over four rounds of 256 block ids, so that the block ids k, k + 256, k + 512, k + 768 (observed: one
    // CU; for speed only) carry equal work.  ga_parts = the larger part count; a shorter column's last part zero-fills the rest.
    int lc_P, lc_nB;
    // forward, merged launch (launch_neg_fwd_bcast_with_edge, TransE_l1): pos-side vectors built on the fly, a = x + asign * r
    const float *xbase; const int64_t *xidx; const float *rbase; const int64_t *ridx; float asign;
};
// GN[j, :] = sum over the nrw row groups of the shared-pair backward's partials (fixed order) + regulariser of the negative row;
// t = float4 index into GN
__device__ __forceinline__ void gn_reduce_body(const NegArgs &a, int nrw, int64_t t) {
    const int64_t n4 = (int64_t)a.C * a.N * a.d_e / 4;
    if (t >= n4) return;
    const int64_t stride = (int64_t)a.C * a.N * a.d_e;
    // the first eight partials are requested together (part index clamped: surplus requests re-read the last part), then added in
    // part order - the same sum as one load -> add round per partial, without eight dependent rounds (cfg-R: nrw = 8)
    float4 pv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) pv[r] = *reinterpret_cast<const float4 *>(a.GNp + (int64_t)(r < nrw ? r : nrw - 1) * stride + 4 * t);
    float4 acc = pv[0];
#pragma unroll
    for (int r = 1; r < 8; ++r)
        if (r < nrw) { acc.x += pv[r].x; acc.y += pv[r].y; acc.z += pv[r].z; acc.w += pv[r].w; }
    for (int r = 8; r < nrw; ++r) {
        const float4 v = *reinterpret_cast<const float4 *>(a.GNp + r * stride + 4 * t);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (a.reg_coef > 0.f && a.reg_norm > 0) {
        const int64_t j = (4 * t) / a.d_e;
        const int k = (int)((4 * t) % a.d_e);
        const float4 x = *reinterpret_cast<const float4 *>(kge::row_ptr(a.nbase, a.nidx, j, a.d_e) + k);
        acc.x += kge::reg_grad(x.x, a.reg_coef, a.reg_norm); acc.y += kge::reg_grad(x.y, a.reg_coef, a.reg_norm);
        acc.z += kge::reg_grad(x.z, a.reg_coef, a.reg_norm); acc.w += kge::reg_grad(x.w, a.reg_coef, a.reg_norm);
    }
    { kge::Pack<4> o; o.v[0] = acc.x; o.v[1] = acc.y; o.v[2] = acc.z; o.v[3] = acc.w; KGE_ST_NEXT<4>(a.GN + 4 * t, o); }
}

struct LossArgs {
    int B, N, genre, adv, pairwise;
    float adv_temp, margin;
    const float *pos, *neg, *w;
    float w_mean;                    // kge_batch.edge_w_mean (0: the rows' wavefronts sum the weights themselves)
    float *dpos, *dneg;              // dneg may alias neg (in place)
    float *row_pos, *row_neg;        // [B] per-row loss terms (already divided by B), or null
    float *acc;                      // [4][KGE_ACC_SLOTS] running loss sums (slot = row & mask) or null
    int l2_scale; float gamma;       // if set: dneg /= (gamma - n)   (TransE_l2 GEMM backward)
    // l2_raw: `neg` holds the raw products a_i . b_j of the merged forward launch; the score is rebuilt here as
    // gamma - sqrt(max(asq[i] + bsq[(i / l2_chunk) * N + j] - 2 neg[i][j], 1e-30))   (score_fun.py:26-34)
    int l2_raw, l2_chunk; const float *asq, *bsq;
    float clampv;                    // > 0: scores were clamped to [-clampv, clampv] (SimplE): no gradient where saturated
    float *neg_copy;                 // optional copy of the scores before overwrite
    int skip_pos;                    // the positive-loss part was already done by edge_fwd
    int diag_chunk;                  // > 0 (neg_deg_sample): column i % diag_chunk of row i is masked - score 0, no gradient
};

struct UpdateArgs {
    int model_d_e, d_r, UE, UR, reg_norm;
    float lr, eps, reg_coef;
    float *ent, *ent_state, *rel, *rel_state;
    kge::ShardMap em, rm;            // sharded tables (n = 0: ent / rel above are the whole tables)
    const int64_t *ue_id; const int32_t *ue_pos_ptr, *ue_pos_adj, *ue_neg_ptr, *ue_neg_slot;
    const int64_t *ur_id; const int32_t *ur_ptr, *ur_edge;
    const int32_t *ue_rec, *ur_rec;  // [UE][8], [UR][8] packed plan records (see kge_batch)
    const int32_t *counts_dev;       // device {UE, UR} for device-built plans (UE/UR are bounds) or null
    const float *GH, *GT, *GN, *GR;
    // TransE fast path: per-edge gradients are rebuilt from P (positive part) and GA (negative
    // part) instead of reading GH/GT/GR written by edge_bwd:  GH = -P (+GA in tail mode),
    // GT = +P (+GA in head mode), GR = -P +/- GA + regulariser
    int transe_fast, neg_head; const float *P, *GA;
    const float *Q;                  // TransE fast path, matrix-core backward: Q = GA +/- P (GemmArgs::Q) - when set, the corrupted-side
                                     // entity gradient is Q, the relation gradient +/-Q (+ regulariser), GA is not read
    float *reg_ent, *reg_rel;        // [UE], [UR] regularisation value partials (or null)
    float *acc;                      // [4][KGE_ACC_SLOTS] running sums (row 3 = regularisation) or null
    // emit mode (sharded training): write gradients instead of updating the entity table
    float *g0, *gs0, *g1, *gs1, *gr, *gsr;
    int emit_ent, emit_rel;
    int emit_by_id;                  // entity messages are written at row `id` (the cache row of dist.py) instead of union entry u
    const int32_t *msg_rows;         // packed single-trace entity messages (kge_emit.msg_rows, ABI 8): [UE][2] {message row, position of the
    int msg_cap, msg_capT;           // second message in its bucket's extra region or -1}, or null; bucket = msg_capT rows, extra region from msg_cap
    int ld_e, ld_r;                  // row strides of the emit buffers (floats)
    int32_t *rid;                    // optional relation-id words inside the relation message
    int ld_gs_e, ld_gs_r;            // strides of gs0/gs1 and gsr
    int dry;                         // tuning probe: read everything, write nothing
    // --async_update pipeline: per-edge dense copies of the h / t / r rows AS GATHERED ([B,d_e], [B,d_e], [B,d_r]);
    // the regulariser gradient / value of a positive-trace row is evaluated on that copy (the reference computes it in
    // forward, from the gathered rows), not on the row as it is when the deferred update lands.  null: current rows.
    const float *Hs, *Ts, *Rs;
    const float *Ns;                 // same for the --neg_deg_sample regulariser of the sampled negative rows: PREP's dense copy
                                     // of the negative rows, indexed like GN (gn_row); null: current rows
    // neg_deg_sample (nd_chunk > 0): GN has nd_Np = nd_chunk + nd_Ns rows per chunk, plan slot k (sampled negative)
    // is row (k / nd_Ns) * nd_Np + nd_chunk + k % nd_Ns, and the regulariser of the negative rows is added here
    int nd_chunk, nd_Ns, nd_Np;
    // TransE_l1 strict step, round 4 (gn_parts > 0): the shared-pair backward's outputs are consumed UNSUMMED - a negative-slot row
    // is the sum of gn_parts rows of GNp gn_stride floats apart (+ the regulariser gradient of the row, gn_reg_*: what
    // gn_reduce_body adds), a GA row the sum of ga_parts rows ga_stride apart; same additions in the same order as the
    // stand-alone reduction launch this replaces (5.2 us + a launch boundary per step)
    const float *GNp; int gn_parts; int64_t gn_stride; float gn_reg_coef; int gn_reg_norm;
    int ga_parts; int64_t ga_stride;
};
__device__ __forceinline__ int64_t gn_row(const UpdateArgs &a, int slot) {
    return a.nd_chunk ? (int64_t)(slot / a.nd_Ns) * a.nd_Np + a.nd_chunk + slot % a.nd_Ns : (int64_t)slot;
}

struct FinalizeArgs {
    int B, UE, UR, pairwise;
    const float *row_pos, *row_neg, *reg_ent, *reg_rel;
    const int32_t *counts_dev;
    float *loss4;
};

int kge_fail(int code, const char *msg);      // records the message kge_last_error() returns (kge_api.hip); returns code
int launch_gather_rows(const float *table, int dim, const int64_t *idx, int64_t n, float *out,
                       hipStream_t s);
int launch_nd_ids(const int64_t *own, const int64_t *neg_ids, int C, int chunk, int Ns, int64_t *out, hipStream_t s);   // [own | sampled] ids per chunk
int launch_nd_fold(const float *GN, float *G, int B, int chunk, int Np, int d, hipStream_t s);   // G[i] += GN[in-batch row of edge i]
int launch_gather3_sharded(const kge::ShardMap &m, int dim, const int64_t *h, const int64_t *t, const int64_t *neg, int B, int n_neg,
                           float *out, int64_t *iota, hipStream_t s);     // [h | t | neg] rows through the shard map + identity ids
int launch_gather_rows_sharded(float *const *shard_rows, int n_shards, int64_t per, int dim,
                               const int64_t *idx, int64_t n, float *out, hipStream_t s);
int launch_edge_fwd(const EdgeFwdArgs &a, hipStream_t s);
int launch_edge_bwd(const EdgeBwdArgs &a, hipStream_t s);
int launch_loss(const LossArgs &a, hipStream_t s);
int launch_finalize(const FinalizeArgs &a, hipStream_t s);
struct SmpTail;                  // sampler tail workgroups riding on a step launch (kge_sampler_tail.hpp); null / phase 0: none
int launch_update(const UpdateArgs &a, hipStream_t s, const SmpTail *tail = nullptr);
int launch_adagrad_scatter(float *table, float *state, int dim, const int64_t *idx,
                           const float *grad, int64_t n, float lr, float eps, hipStream_t s);
int launch_adagrad_apply_packed(float *table, float *state, int dim, const int64_t *idx, const float *msg,
                                int ld, int64_t n, int ntraces, float lr, float eps, hipStream_t s);
int launch_adagrad_apply_rows(float *table, float *state, int dim, const int64_t *idx,
                              const float *g, const float *gs, int64_t n, float lr, float eps,
                              hipStream_t s);
int launch_reduce_acc(float *acc, float *out4, int zero_after, hipStream_t s);
int launch_scatter_add_rows(float *out, int dim, const int64_t *idx, const float *src, int64_t n, hipStream_t s);
int launch_pnorm(const float *x, int64_t n, int dim, int p, float *part, float *out, hipStream_t s);
int launch_pnorm_bwd(const float *x, int64_t total, int p, const float *gout, float *gx, hipStream_t s);
int launch_mask_diag(float *x, int C, int chunk, int Np, hipStream_t s);
int launch_rank_mask(const float *neg, const float *pos, const float *bias, int64_t E, int64_t N, int64_t *ranks, hipStream_t s);
int launch_rank_count(const float *S, const float *P, int rows, int64_t N, const int64_t *filt_ptr,
                      const int64_t *filt_ids, int64_t e0, int32_t *ranks, hipStream_t s);
bool rank_gemm_supported(int model, int d_e);          // kge_rank_gemm.hip: LDS-tiled fp32-MFMA ranking of the matrix-form models
size_t rank_gemm_mask_bytes(int rows, int64_t N);
int launch_rank_gemm(int model, const float *A, int rows, const float *nbase, const int64_t *nidx, int64_t N, int D, float gamma,
                     float clampv, const float *asq, const float *bsq, const float *P, void *mask, const int64_t *filt_ptr,
                     const int64_t *filt_ids, int64_t e0, int32_t *ranks, hipStream_t s);
struct GemmArgs {                   // LDS-staged fp32-MFMA negative scoring (kge_neg_gemm.hip)
    int model, C, chunk, N, D;
    float gamma;
    const float *A;                  // [C*chunk, D] pos-side vectors (dense)
    const float *nbase; const int64_t *nidx;   // negative rows: nbase + (nidx ? nidx[j] : j)*D
    const float *asq, *bsq;          // [C*chunk], [C*N] squared norms (TransE_l2)
    // forward, merged launch (launch_neg_fwd_gemm_with_edge): the pos-side fragments are built ON THE FLY from the table rows
    // the edge-forward half of the same launch is reading - a_i = x_i + asign * r_i (TransE) or x_i * r_i (DistMult), x = head
    // (tail-corrupted step) or tail rows through xidx, r through ridx; A / asq / bsq above are NOT read by the forward tiles
    // then (the other half of the launch is still writing them) and S receives the RAW products a_i . b_j (TransE_l2: the loss
    // kernel applies gamma - sqrt(|a|^2 + |b|^2 - 2 S), LossArgs::l2_raw)
    const float *xbase; const int64_t *xidx; const float *rbase; const int64_t *ridx; float asign;
    int lds_off;                     // merged launch: keep the direct-load tiles instead of the pos-side tile through LDS (A/B aid)
    // forward
    float *S;                        // out [C,chunk,N]
    float clampv;                    // > 0: clamp the scores to [-clampv, clampv] (SimplE)
    // fused loss (PM != null; pointwise criteria): S receives u_ij = crit'(n_ij) * exp(T n_ij - m_it) (/ dist for
    // TransE_l2) instead of the scores, and per (row, 16-column tile): PM = m_it = max_j T*n_ij (0 without -adv),
    // PS = sum_j exp(T n_ij - m_it), PL = sum_j exp(T n_ij - m_it) * criterion(n_ij)
    float *PM, *PS, *PL;             // [C*chunk, tj] or null
    float *Sraw;                     // fused loss: optional copy of the raw scores [C,chunk,N] or null
    float adv_temp;
    // backward
    const float *W;                  // dL/dn (TransE_l2: already / dist); with PM != null: u_ij, scaled per (row, tile) here
    const float *w;                  // [B] edge weights or null
    LossParams lp; int B;
    float *GA, *GN;                  // out [C*chunk, D] (or null when only Q is wanted), [C*N, D]
    // TransE: Q = GA + qc * QP with QP = the per-edge positive-part gradient rows P and qc = +1 (head-corrupted) / -1 -
    // exactly the gradient of the corrupted-side entity row and, up to the sign, of the relation row, so that the update
    // reads ONE row per list entry instead of two (P and GA).  null: not emitted.
    float *Q; const float *QP; float qc;
    // DistMult (round 3): the GA tiles chain GA through the pos-side transform in their epilogue and write the per-edge gradient
    // rows themselves - GH = dp r.t (+ GA.r), GT = dp h.r (+ GA.r), GR = dp h.t + GA.x + regulariser (elementwise in the
    // column, so a 16 x 64 tile has everything it needs) - no edge_bwd launch.  ew_GR != null switches it on; rows gathered from
    // the tables through the batch's edge-end / relation ids
    const float *ew_ent, *ew_rel; const int64_t *ew_h, *ew_t, *ew_r; const float *ew_dpos;
    float *ew_GH, *ew_GT, *ew_GR; float ew_reg_coef; int ew_reg_norm, ew_neg_head;
    float reg_coef; int reg_norm;
    float *row_neg;                  // [B] per-row negative loss terms or null
    float *acc;                      // running sums or null
};
bool neg_mfma_supported(int model, int d_e, int N);
bool neg_gemm_fused_loss_supported(int chunk, int N);   // fused loss of the matrix-core path (else: stand-alone loss kernel)
int launch_neg_fwd_gemm(const GemmArgs &a, hipStream_t s);
int launch_neg_bwd_gemm(const GemmArgs &a, hipStream_t s, const SmpTail *tail = nullptr);
struct UpdateArgs;
// horizontally fused launches of the --async_update pipeline (KGE_ERR_ARG: no fused instantiation for the combination)
int launch_neg_fwd_gemm_with_update(const GemmArgs &a, const UpdateArgs &u, hipStream_t s);
// the strict step's first launch: forward GEMM tiles with on-the-fly pos-side fragments + the edge-forward rows of the SAME step
// (KGE_ERR_ARG: no fused instantiation for the combination)
bool neg_fwd_gemm_with_edge_supported(int model, int d_e, int d_r);
int launch_neg_fwd_gemm_with_edge(const GemmArgs &a, const EdgeFwdArgs &e, hipStream_t s, const SmpTail *tail = nullptr);
// ... and with the loss rows inside (round 4; `tickets`: kge_step_out.tickets, zero on entry and on exit)
struct LossArgs;
bool neg_fwd_loss_fold_supported(int model, int C, int chunk, int N, int d_e, int d_r);
int launch_neg_fwd_gemm_with_edge_loss(const GemmArgs &a, const EdgeFwdArgs &e, const LossArgs &la, int *tickets, hipStream_t s);
int launch_neg_bwd_gemm_with_prep(const GemmArgs &a, const EdgeFwdArgs &e, hipStream_t s);
int launch_neg_fwd_pair(const NegArgs &a, hipStream_t s);
int launch_neg_bwd_pair(const NegArgs &a, hipStream_t s);
// ---- RESCAL (kge_rescal.hip) ----
struct RescalMatvecArgs {            // one pass over M_i = rel + (ridx ? ridx[i] : i) * D*Dc per edge i
    int B, D;                        // D rows; Dc columns (0: square).  y vectors have Dc entries, z / pd vectors D
    int Dc;
    const float *rel; const int64_t *ridx;
    const float *y1; const int64_t *y1idx;   // r1 = M y1   (vectors: base + (idx ? idx[i] : i) * D; null = absent)
    const float *y2; const int64_t *y2idx;   // r2 = M y2
    const float *z1; const int64_t *z1idx;   // c1 = M^T z1
    const float *z2; const int64_t *z2idx;   // c2 = M^T z2
    const float *pd; const int64_t *pdidx;   // p  = pd . (M y1)
    float *r1, *r2, *c1, *c2;                // [B,D] outputs or null
    float *p;                                // [B] or null
};
struct RescalOuterArgs {             // G_i = c_i * u_i v_i^T (+ G_i) (+ regulariser gradient of M_i)
    int B, D;
    const float *c; const float *u; const int64_t *uidx; const float *v; const int64_t *vidx;
    const float *rel; const int64_t *ridx; float reg_coef; int reg_norm;
    int accumulate;
    float *G;                                // [B, D*D]
};
#define RESCAL_RB 8                  // row blocks per relation matrix in the update kernels
#ifndef RESCAL_RBN
#define RESCAL_RBN 16                // row blocks per relation matrix in the per-unique-relation passes of the fused step
#endif
struct RescalUpdateArgs {            // fused Adagrad of the relation matrices (kge_rescal.hip)
    int B, D, UE, UR, neg_head, reg_norm;
    float lr, eps, reg_coef;
    float *rel, *rel_state;
    const float *ent; const int64_t *hidx, *tidx, *rel_ids;
    const float *dpos, *GA;
    const int64_t *ur_id; const int32_t *ur_ptr, *ur_edge; const int32_t *counts_dev;
    float *gs;                       // scratch [B, RESCAL_RB]: mean-square of every traced-row gradient (parts)
    float *inv_std;                  // scratch [UR]
    float *reg_part;                 // scratch [UR, RESCAL_RB] or null (no regularisation value wanted)
    float *reg_rel, *acc;
    float *c1p, *c2p;                // [B, RESCAL_RBN, D] or null.  Non-null: the Adagrad pass over M also produces the backward's
                                     // column products M^T h (c1p) and M^T GA (c2p), one part per row block (ONE pass over M)
    // regulariser products of the forward pass (RescalRelFwdArgs.PV / PW / rho) or null: with them the traced rows' mean squares
    // come in closed form and the regulariser costs no pass over the matrices of its own (kge_rescal.hip rescal_edge_sq_regx_kernel)
    const float *PV, *PW, *rho;
};
struct RescalRelFwdArgs {            // forward products per UNIQUE relation: one pass over M serves every edge of the relation
    int B, D, UR;
    const float *rel, *ent; const int64_t *hidx, *tidx;
    const int64_t *ur_id; const int32_t *ur_ptr, *ur_edge; const int32_t *counts_dev;
    float *V;                        // [B, D]  M t
    float *W;                        // [B, D]  M h, or null
    float *ppart;                    // [B, RESCAL_RBN] parts of p = h . (M t)
    float *P;                        // [B]
    // round 6: with R = the regulariser's gradient of M (elementwise), the same pass also leaves R t (PV [B, D]), R h (PW [B, D], with W)
    // and the row blocks' parts of sum(R^2) per edge (rho [B, RESCAL_RBN]); all null: no regulariser
    float *PV, *PW, *rho; float reg_coef; int reg_norm;
};
struct RescalCombineArgs {           // GH = dp V (+ M^T GA, tail mode), GT = dp M^T h (+ M^T GA, head mode) from the row-block parts
    int B, D, neg_head;
    const float *dpos, *V, *c1p, *c2p;
    float *GH, *GT;
};
int launch_rescal_matvec(const RescalMatvecArgs &a, hipStream_t s);
int launch_rescal_axpy(const float *s1, const float *u1, const float *u2, int B, int D, float *out, hipStream_t s,
                       float alpha = 1.f);      // out_i = alpha * s1_i * u1_i + u2_i
int launch_rescal_outer(const RescalOuterArgs &a, hipStream_t s);
int launch_rescal_update_rel(const RescalUpdateArgs &a, hipStream_t s);
int launch_rescal_rel_fwd(const RescalRelFwdArgs &a, hipStream_t s);
int launch_rescal_combine(const RescalCombineArgs &a, hipStream_t s);
// ---- TransR (kge_transr.hip) ----
struct TransRArgs {
    int B, C, chunk, N, De, Dr, neg_head, UR, reg_norm;
    float gamma, lr, eps, reg_coef;
    const float *ent; const int64_t *h_gid, *t_gid, *neg_ids, *rel_ids;
    const float *rel; float *proj, *proj_state;
    float *HP, *TP, *Q, *SG;         // [B, Dr] projected head / tail, q = x P - r, sign(hp + r - tp)
    float *P;                        // [B] positive scores
    float *S;                        // [B, N] negative scores, later dL/dn
    signed char *Z;                  // [B, N, Dr] sign(Y - q) or null (evaluation)
    float *DQ, *GN, *GP, *GR;        // [B, Dr], [C*N, De], [B, De*Dr], [B, Dr]
    const float *dpos;
    const int64_t *ur_id; const int32_t *ur_ptr, *ur_edge, *counts_dev;
    float *gs0, *gs1, *k0, *k1;      // [B], [B], [UR], [UR] scratch of the projection update
    float *gs1p;                     // [B, tiles of the D_e x D_r matrix]: sum of squares of every tile of GP, written by its producer
    int nd_chunk;                    // > 0: neg_deg_sample - neg_ids is the combined [own | sampled] list of N = nd_chunk + sampled ids per chunk,
                                     // the negative rows' regulariser is left to the update kernel (sampled rows only)
    int nG; float *GNp;              // split-K groups of the negative-row gradient and their partial tiles [nG, C*N, De]
};
#ifndef TRANSR_GN_GROUPS
#define TRANSR_GN_GROUPS 16
#endif
#ifndef TRANSR_GN_GROUPS_WIDE
#define TRANSR_GN_GROUPS_WIDE 64     // split-K groups of the 128 x 208-tile kernel (kge_transr_wide.hpp): 4 chunks x 2 row tiles need them
#endif
int transr_gn_groups(int De, int Dr, int chunk, int N);      // how many groups launch_transr_bwd will use for this shape
int launch_transr_pos(const TransRArgs &a, hipStream_t s);
int launch_transr_fwd(const TransRArgs &a, hipStream_t s);
int launch_transr_bwd(const TransRArgs &a, hipStream_t s);
int launch_transr_proj_update(const TransRArgs &a, hipStream_t s);
bool neg_bcast_supported(int model, int d_e);          // kge_neg_bcast.hip: lane = row, other operand wave-uniform
int launch_neg_fwd_bcast(const NegArgs &a, hipStream_t s);
bool neg_fwd_bcast_with_edge_supported(int model, int d_e, int d_r);
int launch_neg_fwd_bcast_with_edge(const NegArgs &a, const EdgeFwdArgs &e, hipStream_t s);   // TransE_l1: forward tasks || edge-forward rows
int launch_neg_bwd_bcast(const NegArgs &a, hipStream_t s);
bool neg_bwd_lc_supported(int model, int d_e);         // kge_neg_bcast.hip: lane = column, one pair evaluation feeds GA and GN
size_t neg_bwd_lc_partial_floats(int model, int C, int chunk, int N, int d_e);
int neg_bwd_lc_splits(int model, int C, int chunk, int N, int d_e);   // ga_parts that fills the chip (1: not worth it / not supported)
int neg_bwd_lc_nrw(int model, int C, int chunk, int d_e);      // row groups whose partials gn_reduce sums
struct EdgeBwdArgs;
int launch_edge_bwd_with_gn_reduce(const EdgeBwdArgs &a, const NegArgs &n, int nrw, hipStream_t s);
