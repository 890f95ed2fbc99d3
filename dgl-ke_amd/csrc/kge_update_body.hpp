// kge_update_body.hpp - device body of the register-resident owner-computes Adagrad kernel (ExternalEmbedding.update,
// models/pytorch/tensor_models.py:304-362), shared by the stand-alone update kernel (kge_rowwise.hip) and the
// horizontally fused "backward GEMM of step s + entity update of step s-1" launch of the --async_update pipeline
// (kge_neg_gemm.hip).
#pragma once
#include "kge_common.hpp"

#define KGE_ST_ROW kge::st_wt      // updated table rows: read next by another kernel on any XCD - write-through store
#ifndef LANE
#define LANE() (threadIdx.x & 63)
#endif
// running-sum slot update: a fire-and-forget hardware float atomic (global_atomic_add_f32, no return value).
// A plain read-modify-write costs a dependent global load - one more ~1.5 us round trip at the end of every
// wavefront's chain (update kernel 15.7 -> see profiles/r01_microbench_mi355x.txt).  The result is still
// deterministic whenever `unique` holds (every slot gets at most ONE add per kernel, kernels are stream
// ordered): a single atomic add performs exactly old + v.
__device__ __forceinline__ void acc_add(float *p, float v, bool unique) {
    (void)unique;
    atomicAdd(p, v);
}

// sum of `parts` rows `stride` floats apart, added in part order; MAXP requests are issued unconditionally (part index clamped: the
// surplus ones re-read the last part) so that they fly together
template <int MAXP>
__device__ __forceinline__ kge::Pack<4> ld_parts(const float *p, int parts, int64_t stride) {
    using namespace kge;
    Pack<4> v[MAXP];
#pragma unroll
    for (int q = 0; q < MAXP; ++q) v[q] = ld<4>(p + (int64_t)min(q, parts - 1) * stride);
#pragma unroll
    for (int q = 1; q < MAXP; ++q) {
        if (q < parts) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[0].v[e] += v[q].v[e];
        }
    }
    return v[0];
}
#define KGE_GA_MAXP 4          // most GA parts / GN partials the folded update takes (the launcher falls back to the reduction launch)
#define KGE_GN_MAXP 6

// one relation row per wavefront (second half of update_reg_body).  COOP: the instance for batches with a long relation list.
// (Sums of squares are written as explicit fmaf: left to the compiler, the two instances contracted `ss += g * g` differently and
//  a row without any shared list came out 1 ulp apart - device-built batches pick the instance per batch, host plans always COOP.)
template <int NIT, bool SHARDED, int LEAN, bool COOP, bool FOLD = false>
__device__ __forceinline__ void update_rel_row(const UpdateArgs &a, int bx, int nb_ent, int lane, bool reg, bool qm) {
    using namespace kge;
    constexpr int LB2 = NIT <= 2 ? 2 : 1;
    constexpr int LBR = NIT <= 2 ? 6 : 2;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t u = ((int64_t)bx - nb_ent) * KGE_WAVES_PER_BLOCK + wv;
    // COOP (long lists shared by the workgroup, see below): no early exit - the four wavefronts meet at barriers, a wavefront
    // without a row takes part with an empty record.  Otherwise wavefronts without a row leave at once.
    if (!COOP && u >= a.UR) return;
    const int d = a.d_r;
    const int64_t uc = u < a.UR ? u : (int64_t)a.UR - 1;                   // a.UR >= 1: this workgroup exists
    int4 r0 = reinterpret_cast<const int4 *>(a.ur_rec)[2 * uc];           // count and record together (see the entity part)
    const int4 r1 = reinterpret_cast<const int4 *>(a.ur_rec)[2 * uc + 1];
    const int cnt_r = a.counts_dev ? a.counts_dev[1] : a.UR;
    const bool valid = u < a.UR && u < cnt_r;
    // gradient-emitting step on a device-built plan: message rows beyond the batch's relation count are pads (id -1)
    if (!valid && a.rid && u < a.UR && lane == 0) { a.rid[u * (int64_t)a.ld_r] = -1; a.rid[u * (int64_t)a.ld_r + 1] = -1; }
    if (!COOP && !valid) return;
    const int64_t id = valid ? ((int64_t)(uint32_t)r0.x | ((int64_t)r0.y << 32)) : 0;
    const int e0 = valid ? r0.z : 0, e1 = valid ? r0.w : 1, edge0 = valid ? r1.x : 0;
    float *row = shard_row(a.rm, a.rel, id, d);
    float *srow = shard_state(a.rm, a.rel_state, id);
    const int nex = e1 - e0 - 1;
    const int nit = d >> 2;
    const float sgr = a.neg_head ? -1.f : 1.f;       // TransE fast path: GR = -P +/- GA + regulariser
    const int64_t eo0 = (int64_t)edge0 * d;
    const float *pA = a.transe_fast ? (qm ? a.Q : a.P) + eo0 : a.GR + eo0;
    const float *pB = (a.transe_fast && !qm) ? a.GA + eo0 : pA;
    const float *pX = a.Rs ? a.Rs + eo0 : row;        // row the regulariser is evaluated on (see the entity part)
    // ---- long lists are shared by the four wavefronts of the workgroup (single-source modes): every entry is a dependent
    // index -> row round and one wavefront keeps LBR of them in flight - FB15k's most frequent relation has 30 - 40 edges
    // per batch and its wavefront alone set the end of the update kernel (profiles/r02_heavy_lists.txt).  Wavefront w takes
    // the entries w, w + 4, ... of every long list, partial sums meet in LDS and are added by the owner in wavefront order
    // (deterministic; the order differs from the serial one only for such rows). ----
    constexpr int COOP_MIN = 2 * LBR + 1;            // a list is "long" from here
    constexpr bool coop_mode = COOP;
    constexpr int CO_W = COOP ? KGE_WAVES_PER_BLOCK : 1, CO_F = COOP ? NIT * 256 + 64 : 1;
    __shared__ int co_nex[CO_W], co_e0[CO_W];
    __shared__ const float *co_px[CO_W];
    __shared__ float co_part[CO_W][CO_F];                  // per helper: the partial row + one ss partial per lane
    if constexpr (COOP) { if (lane == 0) { co_nex[wv] = valid ? nex : -1; co_e0[wv] = e0; co_px[wv] = pX; } }
    // the workgroup decides HERE, right behind the record (nothing else is in flight yet, the four records arrive together):
    // a barrier after the own rows made every wavefront wait for the slowest row of its workgroup (relation wavefronts p50
    // 2.6 -> 4.7 us on uniform ids)
    bool any_long = false;
    if constexpr (COOP) {                            // (this instance runs when the batch has a long list: kernel-uniform)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KGE_WAVES_PER_BLOCK; ++j) any_long = any_long || co_nex[j] >= COOP_MIN;
    }
    const bool own_long = valid && coop_mode && nex >= COOP_MIN;
    const float st0 = *srow;
    const int nown = own_long ? 0 : nex;             // entries this wavefront adds on its own
    const int edgev = lane < nown ? a.ur_edge[e0 + 1 + lane] : 0;     // rest of the edge list, one entry per lane
    Pack<4> x[NIT], gsum[NIT], xr[NIT];
    float rv = 0.f, ss = 0.f;
    int itc[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) itc[k] = min(lane + 64 * k, nit - 1) * 4;
    Pack<4> fva[NIT], fvb[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {                         // every pack requested before the first use
        x[k] = ld<4>(row + itc[k]);
        fva[k] = ld<4>(pA + itc[k]);
        if constexpr (FOLD) fvb[k] = ld_parts<KGE_GA_MAXP>(pB + itc[k], a.ga_parts, a.ga_stride);    // (FOLD: fast path without Q - pB is a GA row)
        else fvb[k] = ld<4>(pB + itc[k]);
        xr[k] = ld<4>(pX + itc[k]);
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        if (lane + 64 * k < nit) {
            const Pack<4> va = fva[k], vb = fvb[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (reg) rv = reg_acc(rv, xr[k].v[e], a.reg_norm);
                float g;
                if (a.transe_fast) {
                    g = qm ? sgr * va.v[e] : sgr * vb.v[e] - va.v[e];
                    if (reg) g += reg_grad(xr[k].v[e], a.reg_coef, a.reg_norm);
                } else g = va.v[e];
                ss = fmaf(g, g, ss);
                gsum[k].v[e] = g;
            }
        } else { x[k] = zero_pack<4>(); gsum[k] = zero_pack<4>(); xr[k] = zero_pack<4>(); }
    }
    // ---- rest of the edge list, LB entries in flight (see the entity part) ----
    const int nex64 = nown < 64 ? nown : 64;
    if (a.transe_fast && !qm) {
#pragma unroll 1
        for (int i0 = 0; i0 < nex64; i0 += LB2) {
            Pack<4> vp[LB2][NIT], vg[LB2][NIT];
#pragma unroll
            for (int j = 0; j < LB2; ++j) {
                const int64_t eo = (int64_t)__builtin_amdgcn_readlane(edgev, min(i0 + j, nex64 - 1)) * d;
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    vp[j][k] = ld<4>(a.P + eo + itc[k]);
                    if constexpr (FOLD) vg[j][k] = ld_parts<KGE_GA_MAXP>(a.GA + eo + itc[k], a.ga_parts, a.ga_stride);
                    else vg[j][k] = ld<4>(a.GA + eo + itc[k]);
                }
            }
#pragma unroll
            for (int j = 0; j < LB2; ++j) {
                if (i0 + j < nex64) {
#pragma unroll
                    for (int k = 0; k < NIT; ++k) {
                        if (lane + 64 * k < nit) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float g = sgr * vg[j][k].v[e] - vp[j][k].v[e];
                                if (reg) g += reg_grad(xr[k].v[e], a.reg_coef, a.reg_norm);
                                ss = fmaf(g, g, ss); gsum[k].v[e] += g;
                            }
                        }
                    }
                }
            }
        }
    } else {
        const float *gbase = a.transe_fast ? a.Q : a.GR;
#pragma unroll 1
        for (int i0 = 0; i0 < nex64; i0 += LBR) {
            Pack<4> vv[LBR][NIT];
#pragma unroll
            for (int j = 0; j < LBR; ++j) {
                const int64_t eo = (int64_t)__builtin_amdgcn_readlane(edgev, min(i0 + j, nex64 - 1)) * d;
#pragma unroll
                for (int k = 0; k < NIT; ++k) vv[j][k] = ld<4>(gbase + eo + itc[k]);
            }
#pragma unroll
            for (int j = 0; j < LBR; ++j) {
                if (i0 + j < nex64) {
#pragma unroll
                    for (int k = 0; k < NIT; ++k) {
                        if (lane + 64 * k < nit) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float g = vv[j][k].v[e];
                                if (a.transe_fast) {
                                    g = sgr * g;
                                    if (reg) g += reg_grad(xr[k].v[e], a.reg_coef, a.reg_norm);
                                }
                                ss = fmaf(g, g, ss); gsum[k].v[e] += g;
                            }
                        }
                    }
                }
            }
        }
    }
#pragma unroll 1
    for (int i = 64; i < nown; ++i) {                      // lists longer than 65 entries: one entry at a time
        const int64_t eo = (int64_t)a.ur_edge[e0 + 1 + i] * d;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            if (lane + 64 * k < nit) {
                if (a.transe_fast) {
                    const Pack<4> pv = ld<4>((qm ? a.Q : a.P) + eo + itc[k]);
                    Pack<4> gv;
                    if constexpr (FOLD) gv = ld_parts<KGE_GA_MAXP>(a.GA + eo + itc[k], a.ga_parts, a.ga_stride);
                    else gv = ld<4>((qm ? a.Q : a.GA) + eo + itc[k]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float g = qm ? sgr * pv.v[e] : sgr * gv.v[e] - pv.v[e];
                        if (reg) g += reg_grad(xr[k].v[e], a.reg_coef, a.reg_norm);
                        ss = fmaf(g, g, ss); gsum[k].v[e] += g;
                    }
                } else {
                    const Pack<4> g = ld<4>(a.GR + eo + itc[k]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ss = fmaf(g.v[e], g.v[e], ss); gsum[k].v[e] += g.v[e]; }
                }
            }
        }
    }
    // ---- the shared part ----
    if (COOP && any_long) {                          // workgroup-uniform
#pragma unroll 1
        for (int j = 0; j < KGE_WAVES_PER_BLOCK; ++j) {
            const int nj = co_nex[j];
            if (nj < COOP_MIN) continue;
            const int ej = co_e0[j];
            const float *pxj = co_px[j];
            const float *gbase = a.transe_fast ? a.Q : a.GR;
            const int mine = (nj - wv + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK;      // entries wv, wv + 4, ... < nj
            Pack<4> gp[NIT], xj[NIT];
            float pss = 0.f;
#pragma unroll
            for (int k = 0; k < NIT; ++k) { gp[k] = zero_pack<4>(); xj[k] = (reg && a.transe_fast) ? ld<4>(pxj + itc[k]) : zero_pack<4>(); }
#pragma unroll 1
            for (int base = 0; base < mine; base += 64) {
                const int cntb = min(mine - base, 64);
                const int ev = lane < cntb ? a.ur_edge[ej + 1 + wv + KGE_WAVES_PER_BLOCK * (base + lane)] : 0;
#pragma unroll 1
                for (int i0 = 0; i0 < cntb; i0 += LBR) {
                    Pack<4> vv[LBR][NIT];
#pragma unroll
                    for (int jj = 0; jj < LBR; ++jj) {
                        const int64_t eo = (int64_t)__builtin_amdgcn_readlane(ev, min(i0 + jj, cntb - 1)) * d;
#pragma unroll
                        for (int k = 0; k < NIT; ++k) vv[jj][k] = ld<4>(gbase + eo + itc[k]);
                    }
#pragma unroll
                    for (int jj = 0; jj < LBR; ++jj) {
                        if (i0 + jj < cntb) {
#pragma unroll
                            for (int k = 0; k < NIT; ++k) {
                                if (lane + 64 * k < nit) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        float g = vv[jj][k].v[e];
                                        if (a.transe_fast) {
                                            g = sgr * g;
                                            if (reg) g += reg_grad(xj[k].v[e], a.reg_coef, a.reg_norm);
                                        }
                                        pss = fmaf(g, g, pss); gp[k].v[e] += g;
                                    }
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) co_part[wv][(lane + 64 * k) * 4 + e] = gp[k].v[e];
            co_part[wv][NIT * 256 + lane] = pss;
            __syncthreads();
            if (wv == j) {                           // the owner adds the partials, wavefront 0 first
#pragma unroll 1
                for (int w2 = 0; w2 < KGE_WAVES_PER_BLOCK; ++w2) {
#pragma unroll
                    for (int k = 0; k < NIT; ++k)
#pragma unroll
                        for (int e = 0; e < 4; ++e) gsum[k].v[e] += co_part[w2][(lane + 64 * k) * 4 + e];
                    ss += co_part[w2][NIT * 256 + lane];
                }
            }
            __syncthreads();                         // before the next long row overwrites the partials
        }
    }
    if (!valid) return;
    ss = wave_sum(ss) / (float)d;
    if (a.dry) return;
    const float sN = st0 + ss;
    const float kr = -a.lr / (sqrtf(sN) + a.eps);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int it = lane + 64 * k;
        if (it < nit) {
            if (!a.emit_rel) {
                Pack<4> y = x[k];
#pragma unroll
                for (int e = 0; e < 4; ++e) y.v[e] = fmaf(gsum[k].v[e], kr, y.v[e]);
                KGE_ST_ROW<4>(row + it * 4, y);
            }
            if (a.gr) st<4>(a.gr + u * (int64_t)a.ld_r + it * 4, gsum[k]);
        }
    }
    if (lane == 0) {
        if (!a.emit_rel) *srow = sN;
        if (a.gsr) a.gsr[u * (int64_t)a.ld_gs_r] = ss;
        if (a.rid) { a.rid[u * (int64_t)a.ld_r] = (int32_t)(id & 0xFFFFFFFF); a.rid[u * (int64_t)a.ld_r + 1] = (int32_t)(id >> 32); }
    }
    if (reg && (a.reg_rel || a.acc)) {
        rv = wave_sum(rv);
        const float val = a.reg_coef * rv * (float)(e1 - e0);
        if (lane == 0) {
            if (a.reg_rel) a.reg_rel[u] = val;
            if (a.acc) acc_add(&a.acc[3 * KGE_ACC_SLOTS + (int)((u + a.UE) & (KGE_ACC_SLOTS - 1))], val, a.UE + a.UR <= KGE_ACC_SLOTS);
        }
    } else if (a.reg_rel && lane == 0) a.reg_rel[u] = 0.f;
}

// single-pass variant: the row and its gradients stay in registers (row width <= 256*NIT floats),
// so every table row is read once and written once - the algorithmic minimum.  Memory-level
// parallelism: one 32-byte plan record per row gives the row id, the list bounds AND the first
// list entries, so that the row itself and its first positive / negative gradient rows are
// requested together (one dependent round instead of five); longer lists continue in loops.
// SHARDED: rows resolved through the shard map (two 64-bit divisions per wavefront otherwise compiled in);
// LEAN: the common fused-step case - TransE fast path, tables updated in place, no gradient outputs - with the
// generic / emitting code removed at compile time.  Code size matters: the five kernels of a step do not fit the
// instruction cache together, every launch starts cold, and the full-featured kernel was 11 k instructions.
// `bid` / `nblk`: this workgroup's index and the number of workgroups doing update work (the body also runs as one
// half of a horizontally fused launch, see neg_bwd_update_kernel in kge_neg_gemm.hip).
template <int NIT, bool SHARDED, int LEAN_>     // LEAN: 0 = everything at run time, 1 = in-place + TransE fast path,
__device__ __forceinline__ void update_reg_body(const UpdateArgs &a_in, int nb_ent, int bid, int nblk) {   // 2 = in-place + per-edge gradients
    using namespace kge;                         // 3 = 1 with Q rows; 4 (round 4) = 0 with the regulariser's norm fixed at 3 - the
    constexpr int LEAN = (LEAN_ == 4 || LEAN_ == 6 || LEAN_ == 7) ? 0 : (LEAN_ == 5 ? 1 : LEAN_); // gradient-emitting (sharded all-to-all) step of every BASELINE recipe
    // 6 (round 4) = 4 narrowed to what the all-to-all engine's step asks for when the model has per-edge gradient rows (RotatE /
    // cfg-R): messages out, no TransE fast path, no neg_deg_sample, no stale-row copies - the generic instance allocates 155 VGPRs
    // (three wavefronts per SIMD for the step's 4 096 rows of 800 floats)
    // 7 (round 5) = 6 with the RELATION trace applied in place (entity messages out, no relation messages): the all-to-all engine
    // under relation partitioning, where every relation row of a batch belongs to this rank - no relation apply launch at all
    constexpr bool EMIT6 = LEAN_ == 6 || LEAN_ == 7;
    constexpr bool EMIT7 = LEAN_ == 7;
    constexpr bool FOLD = LEAN_ == 5;            // 5 (round 4) = 1 with the shared-pair backward's GN partials / GA parts summed HERE (UpdateArgs::gn_parts)
    UpdateArgs a = a_in;
    if constexpr (!SHARDED) { a.em.n = 0; a.rm.n = 0; }
    if constexpr (LEAN_ == 4 || EMIT6) a.reg_norm = 3;
    if constexpr (EMIT6) {
        a.transe_fast = 0; a.Q = nullptr; a.P = nullptr; a.GA = nullptr; a.dry = 0; a.nd_chunk = 0;
        a.Hs = a.Ts = a.Rs = a.Ns = nullptr; a.emit_ent = 1; a.emit_rel = EMIT7 ? 0 : 1;
        if constexpr (EMIT7) { a.gr = a.gsr = nullptr; a.rid = nullptr; }
    }
    if constexpr (LEAN != 0) {
        a.transe_fast = (LEAN == 1 || LEAN == 3) ? 1 : 0; a.emit_ent = 0; a.emit_rel = 0;
        if (LEAN != 3) a.Q = nullptr;
        a.g0 = a.g1 = a.gs0 = a.gs1 = a.gr = a.gsr = nullptr; a.rid = nullptr; a.dry = 0; a.nd_chunk = 0; a.emit_by_id = 0; a.msg_rows = nullptr;
        a.reg_norm = 3;             // (launch_update sends any other norm to the generic instance)
    }
    const int lane = LANE();
    const bool reg = a.reg_coef > 0.f && a.reg_norm > 0;
    // Q mode (TransE fast path behind the matrix-core backward): one gradient row per list entry, see UpdateArgs::Q
    const bool qm = LEAN == 3 ? true : (LEAN == 0 ? (a.transe_fast && a.Q != nullptr) : false);
    // list entries in flight per wavefront: every entry is a dependent index -> row round (~0.8 us from L2 / MALL), taken one
    // at a time the longest list set the kernel time (FB15k's hub entity / most frequent relation: 20 - 40 entries per batch,
    // 15 - 30 us; profiles/r02_heavy_lists.txt).  Bounded by registers: 4 * NIT per entry and source row.
    constexpr int LB1 = NIT <= 2 ? 4 : 2;       // entity lists: entries with one source row
    constexpr int LB2 = NIT <= 2 ? 2 : 1;       // entries with two (TransE fast path without Q: P and GA)
    constexpr int LBR = NIT <= 2 ? 6 : 2;       // relation lists (one source row; the relation part holds less other state)
    // the (fewer) relation workgroups are dispatched FIRST: measured 14.9 vs 16.5 us - a relation wavefront has
    // the same dependent-load chain as an entity wavefront and must not start after all entity workgroups
    const int nb_rel = nblk - nb_ent;
    const int bx = bid < nb_rel ? bid + nb_ent : bid - nb_rel;
    if (bx < nb_ent) {
#ifdef UPD_PROBE_NOENT
        return;
#endif
        // (the wavefront number through readfirstlane: the compiler then knows that the record / count addresses are wave-uniform
        //  and fetches them with scalar loads)
        const int64_t u = (int64_t)bx * KGE_WAVES_PER_BLOCK + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        if (u >= a.UE) return;                 // UE: the bound the grid and the record array were sized for
        const int d = a.model_d_e;
        // the device-side row count and this wavefront's plan record are requested TOGETHER (the record of a row beyond the
        // count is allocated, just unused): count -> record -> rows was one dependent round more in every wavefront
        int4 r0 = reinterpret_cast<const int4 *>(a.ue_rec)[2 * u];
        const int4 r1 = reinterpret_cast<const int4 *>(a.ue_rec)[2 * u + 1];
        const int cnt_e = a.counts_dev ? a.counts_dev[0] : a.UE;
        asm volatile("" : "+v"(r0.x));         // (keeps the record loads above the early exit)
        if (u >= cnt_e) return;
        const int64_t id = (int64_t)(uint32_t)r0.x | ((int64_t)r0.y << 32);
#if defined(UPD_PROBE_NOGRAD)      // tuning probe: table read-modify-write only (no gradient rows)
        const int p0 = 0, p1 = 0, n0 = 0, n1 = 0, adj0 = 0, slot0 = 0; (void)r1;
#elif defined(UPD_PROBE_NONEG)     // tuning probe: no negative-gradient rows
        const int p0 = r0.z, p1 = r0.w, n0 = 0, n1 = 0, adj0 = r1.z, slot0 = 0;
#else
        const int p0 = r0.z, p1 = r0.w, n0 = r1.x, n1 = r1.y, adj0 = r1.z, slot0 = r1.w;
#endif
        float *row = shard_row(a.em, a.ent, id, d);
        float *srow = shard_state(a.em, a.ent_state, id);
        const bool has_pos = p1 > p0, has_neg = n1 > n0;
        const int nit = d >> 2;
        // first positive contribution: two source rows (fast path: P and maybe GA; generic: GH|GT)
        const int64_t e0 = has_pos ? (int64_t)(adj0 >> 1) * d : 0;
        const int side0 = adj0 & 1;
        const bool ga0 = a.transe_fast && has_pos && side0 == a.neg_head;
        const float sg0 = a.transe_fast ? (side0 ? 1.f : -1.f) : 1.f;
        const float *pA = !has_pos ? row : (a.transe_fast ? ((qm && ga0) ? a.Q : a.P) + e0 : (side0 ? a.GT : a.GH) + e0);
        const float *pB = (ga0 && !qm) ? a.GA + e0 : pA;  // aliases pA when unused (same lines, no extra traffic)
        const float *pC = has_neg ? a.GN + gn_row(a, slot0) * d : row;
        // row the regulariser is evaluated on: as gathered by this step (async pipeline) or as it is now
        const float *pX = (a.Hs && has_pos) ? (side0 ? a.Ts : a.Hs) + e0 : row;
        const bool ndreg = a.nd_chunk && reg;       // neg_deg_sample: regulariser of the negative rows added here
        // ... evaluated on the negative row as gathered (async pipeline: PREP's dense copy of the sampled rows)
        const float *pXn = (a.Ns && has_neg) ? a.Ns + gn_row(a, slot0) * d : row;
        const float st0 = *srow;
        // the rest of the two lists (entries 1..) is requested NOW, one entry per lane, together with the
        // rows above: a serial "load index -> load row" chain per extra entry made the longest list set the
        // kernel time (8 of 16 us came from the few rows with 3-5 contributions)
        const int npx = p1 - p0 - 1, nnx = n1 - n0 - 1;
        const int adjv = lane < npx ? a.ue_pos_adj[p0 + 1 + lane] : 0;
        const int slotv = lane < nnx ? a.ue_neg_slot[n0 + 1 + lane] : 0;
        Pack<4> x[NIT], g0[NIT], g1[NIT], xn[NIT];
        float rv = 0.f, s0 = 0.f, s1 = 0.f;
        // all packs of the row and of its first contributions are requested before the first use (lane offsets clamped, not
        // predicated): with the loads inside `if (it < nit)` every 64 packs were a dependent round of their own
        int itc[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) itc[k] = min(lane + 64 * k, nit - 1) * 4;
        Pack<4> fva[NIT], fvb[NIT], fvc[NIT], fxr[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            x[k] = ld<4>(row + itc[k]);
            fva[k] = ld<4>(pA + itc[k]);
            if constexpr (FOLD) {
                fvb[k] = ld_parts<KGE_GA_MAXP>(pB + itc[k], (ga0 && !qm) ? a.ga_parts : 1, a.ga_stride);
                fvc[k] = ld_parts<KGE_GN_MAXP>((has_neg ? a.GNp + gn_row(a, slot0) * d : row) + itc[k], has_neg ? a.gn_parts : 1, a.gn_stride);
            } else { fvb[k] = ld<4>(pB + itc[k]); fvc[k] = ld<4>(pC + itc[k]); }
            fxr[k] = ld<4>(pX + itc[k]);                    // aliases the row itself outside the async pipeline
            xn[k] = ndreg ? ld<4>(pXn + itc[k]) : x[k];
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            if (lane + 64 * k < nit) {
                const Pack<4> va = fva[k], vb = fvb[k], vc = fvc[k], xr = fxr[k];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float g = 0.f;
                    if (reg) { rv = reg_acc(rv, xr.v[e], a.reg_norm); g = reg_grad(xr.v[e], a.reg_coef, a.reg_norm); }
                    if (has_pos) g += (qm && ga0) ? va.v[e] : sg0 * va.v[e] + (ga0 ? vb.v[e] : 0.f);
                    g0[k].v[e] = g;
                    float gn = has_neg ? vc.v[e] : 0.f;
                    if constexpr (FOLD) { if (has_neg && a.gn_reg_coef > 0.f) gn += reg_grad(x[k].v[e], a.gn_reg_coef, a.gn_reg_norm); }   // (gn_reduce_body's term, on the row as it is)
                    if (ndreg && has_neg) gn += reg_grad(xn[k].v[e], a.reg_coef, a.reg_norm);
                    g1[k].v[e] = gn;
                    s1 = fmaf(gn, gn, s1);
                }
            } else { x[k] = zero_pack<4>(); g0[k] = zero_pack<4>(); g1[k] = zero_pack<4>(); xn[k] = zero_pack<4>(); }
        }
        // ---- rest of the positive list: LB entries requested together (loads unconditional - lane offsets clamped, list index
        // clamped - and consumed in list order: the sums are the same, bit for bit, as one entry at a time) ----
        const int npx64 = npx < 64 ? npx : 64;               // the first 64 extra entries sit in adjv (one per lane)
        if (a.transe_fast && !qm) {
#pragma unroll 1
            for (int i0 = 0; i0 < npx64; i0 += LB2) {
                Pack<4> vp[LB2][NIT], vg[LB2][NIT]; float sgj[LB2]; bool wg[LB2];
#pragma unroll
                for (int j = 0; j < LB2; ++j) {
                    const int adj = __builtin_amdgcn_readlane(adjv, min(i0 + j, npx64 - 1));
                    const int64_t eo = (int64_t)(adj >> 1) * d;
                    const int side = adj & 1;
                    sgj[j] = side ? 1.f : -1.f; wg[j] = side == a.neg_head;
#pragma unroll
                    for (int k = 0; k < NIT; ++k) {
                        vp[j][k] = ld<4>(a.P + eo + itc[k]);
                        if constexpr (FOLD) vg[j][k] = ld_parts<KGE_GA_MAXP>((wg[j] ? a.GA : a.P) + eo + itc[k], wg[j] ? a.ga_parts : 1, a.ga_stride);
                        else vg[j][k] = ld<4>((wg[j] ? a.GA : a.P) + eo + itc[k]);
                    }
                }
#pragma unroll
                for (int j = 0; j < LB2; ++j) {
                    if (i0 + j < npx64) {
#pragma unroll
                        for (int k = 0; k < NIT; ++k)
#pragma unroll
                            for (int e = 0; e < 4; ++e) g0[k].v[e] += sgj[j] * vp[j][k].v[e] + (wg[j] ? vg[j][k].v[e] : 0.f);
                    }
                }
            }
        } else {
#pragma unroll 1
            for (int i0 = 0; i0 < npx64; i0 += LB1) {
                Pack<4> vv[LB1][NIT]; float sgj[LB1];
#pragma unroll
                for (int j = 0; j < LB1; ++j) {
                    const int adj = __builtin_amdgcn_readlane(adjv, min(i0 + j, npx64 - 1));
                    const int64_t eo = (int64_t)(adj >> 1) * d;
                    const int side = adj & 1;
                    const bool isq = a.transe_fast && side == a.neg_head;            // corrupted side: the row of Q
                    const float *src = (a.transe_fast ? (isq ? a.Q : a.P) : (side ? a.GT : a.GH)) + eo;
                    sgj[j] = (a.transe_fast && !isq) ? (side ? 1.f : -1.f) : 1.f;
#pragma unroll
                    for (int k = 0; k < NIT; ++k) vv[j][k] = ld<4>(src + itc[k]);
                }
#pragma unroll
                for (int j = 0; j < LB1; ++j) {
                    if (i0 + j < npx64) {
#pragma unroll
                        for (int k = 0; k < NIT; ++k)
#pragma unroll
                            for (int e = 0; e < 4; ++e) g0[k].v[e] += sgj[j] * vv[j][k].v[e];
                    }
                }
            }
        }
#pragma unroll 1
        for (int i = 64; i < npx; ++i) {                      // lists longer than 65 entries: one entry at a time
            const int adj = a.ue_pos_adj[p0 + 1 + i];
            const int64_t eo = (int64_t)(adj >> 1) * d;
            const int side = adj & 1;
            const bool wga = a.transe_fast && side == a.neg_head;
            const float sg = a.transe_fast ? ((qm && wga) ? 1.f : (side ? 1.f : -1.f)) : 1.f;
            const float *src = (a.transe_fast ? ((qm && wga) ? a.Q : a.P) : (side ? a.GT : a.GH)) + eo;
            const float *src2 = (wga && !qm) ? a.GA + eo : src;
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const Pack<4> g = ld<4>(src + itc[k]);
                Pack<4> g2;
                if constexpr (FOLD) g2 = ld_parts<KGE_GA_MAXP>(src2 + itc[k], (wga && !qm) ? a.ga_parts : 1, a.ga_stride);
                else g2 = ld<4>(src2 + itc[k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) g0[k].v[e] += sg * g.v[e] + ((wga && !qm) ? g2.v[e] : 0.f);
            }
        }
        // ---- rest of the negative list (rows of GN) ----
        if constexpr (FOLD) {                                  // partial rows summed here: one entry (gn_parts requests per pack) at a time
#pragma unroll 1
            for (int i = 0; i < nnx; ++i) {
                const float *src = a.GNp + gn_row(a, a.ue_neg_slot[n0 + 1 + i]) * d;
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    Pack<4> g = ld_parts<KGE_GN_MAXP>(src + itc[k], a.gn_parts, a.gn_stride);
                    if (a.gn_reg_coef > 0.f) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) g.v[e] += reg_grad(x[k].v[e], a.gn_reg_coef, a.gn_reg_norm);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (lane + 64 * k < nit) s1 = fmaf(g.v[e], g.v[e], s1);
                        g1[k].v[e] += g.v[e];
                    }
                }
            }
        }
        const int nnx64 = FOLD ? 0 : (nnx < 64 ? nnx : 64);
#pragma unroll 1
        for (int i0 = 0; i0 < nnx64; i0 += LB1) {
            Pack<4> vv[LB1][NIT];
#pragma unroll
            for (int j = 0; j < LB1; ++j) {
                const int slot = __builtin_amdgcn_readlane(slotv, min(i0 + j, nnx64 - 1));
                const float *src = a.GN + gn_row(a, slot) * d;
#pragma unroll
                for (int k = 0; k < NIT; ++k) vv[j][k] = ld<4>(src + itc[k]);
            }
#pragma unroll
            for (int j = 0; j < LB1; ++j) {
                if (i0 + j < nnx64) {
#pragma unroll
                    for (int k = 0; k < NIT; ++k) {
                        Pack<4> g = vv[j][k];
                        if (ndreg) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) g.v[e] += reg_grad(xn[k].v[e], a.reg_coef, a.reg_norm);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (lane + 64 * k < nit) s1 = fmaf(g.v[e], g.v[e], s1);
                            g1[k].v[e] += g.v[e];
                        }
                    }
                }
            }
        }
#pragma unroll 1
        for (int i = FOLD ? nnx : 64; i < nnx; ++i) {
            const float *src = a.GN + gn_row(a, a.ue_neg_slot[n0 + 1 + i]) * d;
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                Pack<4> g = ld<4>(src + itc[k]);
                if (ndreg) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) g.v[e] += reg_grad(xn[k].v[e], a.reg_coef, a.reg_norm);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (lane + 64 * k < nit) s1 = fmaf(g.v[e], g.v[e], s1);
                    g1[k].v[e] += g.v[e];
                }
            }
        }
        if (has_pos) {
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                if (lane + 64 * k < nit) {        // (lanes beyond the row hold clamped duplicates from the list loops)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s0 = fmaf(g0[k].v[e], g0[k].v[e], s0);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NIT; ++k) g0[k] = zero_pack<4>();
        }
        // two wave reductions (DPP + readlane, see kge_common.hpp)
        s0 = wave_sum(s0); s1 = wave_sum(s1);
        s0 /= (float)d; s1 /= (float)d;
        if (a.dry) return;
        const float sA = has_pos ? st0 + s0 : st0;
        const float sB = has_neg ? sA + s1 : sA;
        const int64_t mu = a.emit_by_id ? id : u;       // message row: union entry, or the row id itself (cache rows, dist.py)
        int mr0 = 0, mr1 = -1, mlink = -1;              // packed messages: rows of the first / second message, the second one's bucket position
        if (a.msg_rows) {
            const int2 mr = reinterpret_cast<const int2 *>(a.msg_rows)[u];
            mr0 = mr.x; mlink = mr.y;
            // (second message: position mlink of the extra region of the SAME owner bucket - buckets are msg_capT rows apart, the
            //  extra region starts msg_cap rows in; a division per wavefront, only for the rare rows that are in both traces)
            if (mlink >= 0) mr1 = (mr0 / a.msg_capT) * a.msg_capT + a.msg_cap + mlink;
        }
        // one division per row (-lr / std), then multiply-adds: 24 IEEE division sequences per wavefront were
        // 10 % of this kernel's instructions (<= 1 ulp from the reference's per-element division)
        const float k0 = -a.lr / (sqrtf(sA) + a.eps), k1 = -a.lr / (sqrtf(sB) + a.eps);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int it = lane + 64 * k;
            if (it < nit) {
                if (!a.emit_ent) {
                    Pack<4> y = x[k];
                    if (has_pos) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) y.v[e] = fmaf(g0[k].v[e], k0, y.v[e]);
                    }
                    if (has_neg) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) y.v[e] = fmaf(g1[k].v[e], k1, y.v[e]);
                    }
                    KGE_ST_ROW<4>(row + it * 4, y);
                }
                if (a.msg_rows) {
                    // packed single-trace messages (round 6): ONE message [g | gs | link] per row - its positive trace, or its negative
                    // trace when it has no positive one; a row in BOTH traces writes the negative trace as a second message into the
                    // owner bucket's extra region (row mr1 >= 0, assigned by kge_route_build)
                    st<4>(a.g0 + (int64_t)mr0 * a.ld_e + it * 4, has_pos ? g0[k] : g1[k]);
                    if (has_pos && has_neg && mr1 >= 0) st<4>(a.g0 + (int64_t)mr1 * a.ld_e + it * 4, g1[k]);
                } else {
                if (a.g0) st<4>(a.g0 + mu * (int64_t)a.ld_e + it * 4, g0[k]);      // (messages: plain - write-through measured 1.2 us slower on the a2a step)
                if (a.g1) st<4>(a.g1 + mu * (int64_t)a.ld_e + it * 4, g1[k]);
                }
            }
        }
        if (lane == 0) {
            if (!a.emit_ent) *srow = sB;
            if (a.msg_rows) {
                const bool two = has_pos && has_neg && mr1 >= 0;
                float *h0 = a.g0 + (int64_t)mr0 * a.ld_e + d;
                h0[0] = has_pos ? s0 : (has_neg ? s1 : 0.f);
                // link: the second message's position inside the bucket's extra region (the bucket keeps its layout through the all-to-all)
                h0[1] = __int_as_float(two ? mlink : -1);
                if (two) { float *h1 = a.g0 + (int64_t)mr1 * a.ld_e + d; h1[0] = s1; h1[1] = __int_as_float(-1); }
            } else {
            if (a.gs0) a.gs0[mu * (int64_t)a.ld_gs_e] = has_pos ? s0 : 0.f;
            if (a.gs1) a.gs1[mu * (int64_t)a.ld_gs_e] = has_neg ? s1 : 0.f;
            }
        }
#ifdef UPD_PROBE_NOACC
        if (false) {
#else
        if (reg && (a.reg_ent || a.acc)) {
#endif
            rv = wave_sum(rv);
            const float val = a.reg_coef * rv * (float)((has_pos ? 1 : 0) + (n1 - n0));
            if (lane == 0) {
                if (a.reg_ent) a.reg_ent[u] = val;
                if (a.acc) acc_add(&a.acc[3 * KGE_ACC_SLOTS + (int)(u & (KGE_ACC_SLOTS - 1))], val, a.UE + a.UR <= KGE_ACC_SLOTS);
            }
        } else if (a.reg_ent && lane == 0) a.reg_ent[u] = 0.f;
    } else {
#ifdef UPD_PROBE_NOREL
        return;
#endif
        // two instances of the relation part: long relation lists are shared by the four wavefronts of a workgroup (barriers, LDS);
        // device-built plans carry the length of the batch's longest relation list in counts[3], and a batch without a long one
        // (kernel-uniform) runs the plain instance - the shared-list code costs 0.25 us per step on uniform ids just by being
        // there (registers, code layout; profiles/r02_heavy_lists.txt)
        constexpr int COOP_MIN_R = 2 * (NIT <= 2 ? 6 : 2) + 1;
        const bool coop = !(a.transe_fast && !qm) && (!a.counts_dev || a.counts_dev[3] - 1 >= COOP_MIN_R);
        if constexpr (FOLD) update_rel_row<NIT, SHARDED, LEAN, false, true>(a, bx, nb_ent, lane, reg, qm);   // (fast path without Q: never the shared-list instance)
        else if (coop) update_rel_row<NIT, SHARDED, LEAN, true>(a, bx, nb_ent, lane, reg, qm);
        else update_rel_row<NIT, SHARDED, LEAN, false>(a, bx, nb_ent, lane, reg, qm);
    }
}

