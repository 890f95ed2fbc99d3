// kge_edge_fwd_body.hpp - device body of the edge-forward / PREP kernel (row gather, positive score, pos-side vector,
// positive-loss part, TransE P rows, dense copies for the async pipeline), shared by the stand-alone kernel
// (kge_rowwise.hip) and the fused "backward GEMM + PREP of the next step" launch (kge_neg_gemm.hip).
// Reference: ExternalEmbedding.__call__ (tensor_models.py:292), score_func.edge_func and the head of the create_neg
// closures (score_fun.py:54-59, 94-105, 229-235, 270-283, 297-307, 347-371, 460-472, 516-545).
#pragma once
#include "kge_common.hpp"
#include "kge_update_body.hpp"      // LANE(), acc_add

// Store policies (A/B measured on MI355X, profiles/r02_store_policy.txt): lines left dirty in the producer XCD's L2 are written
// back before the next launch may start, and their consumers run on any XCD.  A (read by the very next kernel): write-through to the
// memory side; P (read two kernels later, by the update): streaming store.
#define KGE_ST_A kge::st_wt
#define KGE_ST_OUT kge::st_nt
//      // P rows: consumed two kernels later by the update (any XCD) - streaming store

// LEAN: local (un-sharded) tables and the Logsigmoid criterion fixed at compile time (no 64-bit divisions of
// the shard map, no three-way loss switch) - the configuration of every single-GPU BASELINE workload
// `bid`: this workgroup's index among the workgroups doing edge_fwd work (the body is also one half of the horizontally
// fused "backward GEMM of step s + PREP of step s+1" launch of the --async_update pipeline, kge_neg_gemm.hip)
template <int MODEL, int V, bool LEAN>
__device__ __forceinline__ void edge_fwd_body(const EdgeFwdArgs &a_in, int bid) {
    using namespace kge;
    EdgeFwdArgs a = a_in;
    if constexpr (LEAN) { a.src.em.n = 0; a.src.rm.n = 0; a.lp.genre = KGE_LOSS_LOGSIGMOID; a.row_pos = nullptr; a.Hc = nullptr; a.nd_own = nullptr; }
    const int64_t w = (int64_t)bid * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int lane = LANE();
    if (w < a.B) {
        const int64_t i = w;
        const float *h = table_row(a.src.em, a.src.hbase, a.src.hidx, i, a.d_e);
        const float *t = table_row(a.src.em, a.src.tbase, a.src.tidx, i, a.d_e);
        const float *r = table_row(a.src.rm, a.src.rbase, a.src.ridx, i, a.d_r);
        float *A = a.A ? a.A + i * (int64_t)a.d_e : nullptr;
        float ps = 0.f, as = 0.f;
        // rows of <= 128 packs (d <= 512 floats) stay in registers: ONE round of row loads (all requested before the first use,
        // lane offsets clamped instead of predicated) serves the score, the pos-side vector, the P rows and the dense copies -
        // the loop below took one dependent round per 64 packs and the later passes re-read the rows
        constexpr int EK = 2;
        Pack<V> hreg[EK], rreg[EK], treg[EK];
        bool regres = false;
        if constexpr (!is_complex_model(MODEL)) regres = a.d_e / V <= 64 * EK && a.d_r == a.d_e;
        if constexpr (!is_complex_model(MODEL)) {
            const int nit = a.d_e / V;
            if (regres) {
#pragma unroll
                for (int k = 0; k < EK; ++k) {
                    const int off = min(lane + 64 * k, nit - 1) * V;
                    hreg[k] = ld<V>(h + off); rreg[k] = ld<V>(r + off); treg[k] = ld<V>(t + off);
                }
#pragma unroll
                for (int k = 0; k < EK; ++k) {
                    if (lane + 64 * k < nit) {
                        const int off = (lane + 64 * k) * V;
                        Pack<V> av;
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            const float hh = hreg[k].v[e], rr = rreg[k].v[e], tt = treg[k].v[e];
                            const float x = a.neg_head ? tt : hh;
                            if constexpr (MODEL == KGE_DISTMULT) {
                                ps += hh * rr * tt;
                                av.v[e] = x * rr;
                            } else {
                                const float u = hh + rr - tt;
                                if constexpr (MODEL == KGE_TRANSE_L1) ps += fabsf(u); else ps += u * u;
                                av.v[e] = a.neg_head ? (x - rr) : (x + rr);
                            }
                            as += av.v[e] * av.v[e];
                        }
                        if (A) KGE_ST_A<V>(A + off, av);
                    }
                }
            } else
            for (int it = lane; it < nit; it += 64) {
                const int off = it * V;
                const Pack<V> hv = ld<V>(h + off), rv = ld<V>(r + off), tv = ld<V>(t + off);
                Pack<V> av;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float hh = hv.v[e], rr = rv.v[e], tt = tv.v[e];
                    const float x = a.neg_head ? tt : hh;
                    if constexpr (MODEL == KGE_DISTMULT) {
                        ps += hh * rr * tt;
                        av.v[e] = x * rr;
                    } else {
                        const float u = hh + rr - tt;
                        if constexpr (MODEL == KGE_TRANSE_L1) ps += fabsf(u); else ps += u * u;
                        av.v[e] = a.neg_head ? (x - rr) : (x + rr);
                    }
                    as += av.v[e] * av.v[e];
                }
                if (A) KGE_ST_A<V>(A + off, av);
            }
        } else {
            const int hd = a.d_e / 2;
            const int nit = hd / V;
            // one column pack of the row: rr_in / ir_in are the relation's (re | im) halves - RotatE: rr_in is the phase pack
            auto cx_pack = [&](int off, const Pack<V> &rh, const Pack<V> &ih, const Pack<V> &rt, const Pack<V> &it_,
                               const Pack<V> &rr_in, const Pack<V> &ir_in) {
                Pack<V> rr, ir;
                if constexpr (MODEL == KGE_COMPLEX || MODEL == KGE_SIMPLE) {
                    rr = rr_in; ir = ir_in;                            // SimplE: rel | rel_inv
                } else {
#pragma unroll
                    for (int e = 0; e < V; ++e) sincosf(rr_in.v[e] / a.rot_div, &ir.v[e], &rr.v[e]);
                }
                Pack<V> are, aim;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float c = rr.v[e], s = ir.v[e];
                    if constexpr (MODEL == KGE_SIMPLE) {
                        // (h_i, h_j) = (rh, ih), (t_i, t_j) = (rt, it_), (rel, rel_inv) = (c, s); score_fun.py:562-568
                        ps += 0.5f * (rh.v[e] * c * it_.v[e] + rt.v[e] * s * ih.v[e]);
                        // pos-side vector, 1/2 folded in, laid out so that a . neg = the chunked score
                        // (score_fun.py:611-622 head mode, :626-641 tail mode)
                        if (a.neg_head) { are.v[e] = 0.5f * c * it_.v[e]; aim.v[e] = 0.5f * s * rt.v[e]; }
                        else            { are.v[e] = 0.5f * s * ih.v[e];  aim.v[e] = 0.5f * rh.v[e] * c; }
                        continue;
                    } else if constexpr (MODEL == KGE_COMPLEX) {
                        ps += rh.v[e] * rt.v[e] * c + ih.v[e] * it_.v[e] * c +
                              rh.v[e] * it_.v[e] * s - ih.v[e] * rt.v[e] * s;
                    } else {
                        const float re = rh.v[e] * c - ih.v[e] * s - rt.v[e];
                        const float im = rh.v[e] * s + ih.v[e] * c - it_.v[e];
                        ps += sqrtf(re * re + im * im);
                    }
                    // (explicit fmaf forms: the forward tiles of the merged first launch build the same vector, bit for bit)
                    if (a.neg_head) {   // a = t o conj(r)
                        are.v[e] = fmaf(it_.v[e], s, rt.v[e] * c);
                        aim.v[e] = fmaf(-rt.v[e], s, it_.v[e] * c);
                    } else {            // a = h o r
                        are.v[e] = fmaf(-ih.v[e], s, rh.v[e] * c);
                        aim.v[e] = fmaf(rh.v[e], s, ih.v[e] * c);
                    }
                    as += are.v[e] * are.v[e] + aim.v[e] * aim.v[e];
                }
                if (A) { KGE_ST_A<V>(A + off, are); KGE_ST_A<V>(A + hd + off, aim); }
            };
            constexpr bool RPH = MODEL == KGE_ROTATE;                  // the relation row is one pack of phases per column pack
            if (nit > 64 && nit <= 64 * EK) {
                // rows of 65 .. 128 column packs (cfg-R: 100): every pack of both passes requested before the first use, lane offsets
                // clamped - the loop below takes one dependent load round PER 64 packs (round 5: edge_fwd's wavefronts lived
                // 6.4 us at D_e = 800 against 3.1 us at 400, tools/timeline.py).  Same packs, same order: bit-identical sums.
                Pack<V> rhv[EK], ihv[EK], rtv[EK], itv[EK], r0v[EK], r1v[EK];
#pragma unroll
                for (int k = 0; k < EK; ++k) {
                    const int off = min(lane + 64 * k, nit - 1) * V;
                    rhv[k] = ld<V>(h + off); ihv[k] = ld<V>(h + hd + off);
                    rtv[k] = ld<V>(t + off); itv[k] = ld<V>(t + hd + off);
                    r0v[k] = ld<V>(r + off);
                    if constexpr (!RPH) r1v[k] = ld<V>(r + hd + off); else r1v[k] = r0v[k];
                }
#pragma unroll
                for (int k = 0; k < EK; ++k)
                    if (lane + 64 * k < nit) cx_pack((lane + 64 * k) * V, rhv[k], ihv[k], rtv[k], itv[k], r0v[k], r1v[k]);
            } else
            for (int it = lane; it < nit; it += 64) {
                const int off = it * V;
                const Pack<V> rh = ld<V>(h + off), ih = ld<V>(h + hd + off);
                const Pack<V> rt = ld<V>(t + off), it_ = ld<V>(t + hd + off);
                const Pack<V> r0 = ld<V>(r + off);
                Pack<V> r1 = r0;
                if constexpr (!RPH) r1 = ld<V>(r + hd + off);
                cx_pack(off, rh, ih, rt, it_, r0, r1);
            }
        }
        if (a.pos_score || a.do_pos_loss) {
            ps = wave_sum(ps);          // xor butterfly: every lane holds the sum
            float p;
            if constexpr (MODEL == KGE_TRANSE_L1 || MODEL == KGE_ROTATE) p = a.gamma - ps;
            else if constexpr (MODEL == KGE_TRANSE_L2) p = a.gamma - sqrtf(ps);
            else if constexpr (MODEL == KGE_SIMPLE) p = fminf(fmaxf(ps, -KGE_SIMPLE_CLAMP), KGE_SIMPLE_CLAMP);
            else p = ps;
            if (lane == 0 && a.pos_score) a.pos_score[i] = p;
            if (a.do_pos_loss) {
                // pointwise losses: d loss / d p_i depends on p_i only (loss.py:82-94)
                const float w = mean_edge_weight(a.w, a.B, lane, a.w_mean);      // (the batch's MEAN importance: see kge_common.hpp)
                const float invB = 1.f / (float)a.B;
                float pl, dpl;
                criterion(a.lp.genre, p, 1.f, a.lp.margin, pl, dpl);
                if constexpr (MODEL == KGE_SIMPLE) { if (fabsf(ps) >= KGE_SIMPLE_CLAMP) dpl = 0.f; }   // saturated clamp
                const float dp = dpl * w * 0.5f * invB;
                if (lane == 0) {
                    const float plw = pl * w * invB;
                    if (a.dpos) a.dpos[i] = dp;
                    if (a.row_pos) a.row_pos[i] = plw;
                    if (a.acc) {
                        const bool uq = a.B <= KGE_ACC_SLOTS;
                        const int slot = (int)(i & (KGE_ACC_SLOTS - 1));
                        acc_add(&a.acc[0 * KGE_ACC_SLOTS + slot], plw, uq);
                        acc_add(&a.acc[2 * KGE_ACC_SLOTS + slot], 0.5f * plw, uq);
                    }
                }
                if constexpr (MODEL == KGE_TRANSE_L1 || MODEL == KGE_TRANSE_L2) {
                    if (a.P) {
                        // P_i = dp * d|u|/du, u = h + r - t  (second pass: rows are L1/L2 hot)
                        float inv = 0.f;
                        if constexpr (MODEL == KGE_TRANSE_L2) { const float nr = sqrtf(ps); inv = nr > 0.f ? 1.f / nr : 0.f; }
                        float *P = a.P + i * (int64_t)a.d_e;
                        if (regres) {
#pragma unroll
                            for (int k = 0; k < EK; ++k) {
                                if (lane + 64 * k < a.d_e / V) {
                                    Pack<V> pv;
#pragma unroll
                                    for (int e = 0; e < V; ++e) {
                                        const float u = hreg[k].v[e] + rreg[k].v[e] - treg[k].v[e];
                                        pv.v[e] = dp * ((MODEL == KGE_TRANSE_L1) ? sgnf(u) : u * inv);
                                    }
                                    KGE_ST_OUT<V>(P + (lane + 64 * k) * V, pv);
                                }
                            }
                        } else
                        for (int it = lane; it < a.d_e / V; it += 64) {
                            const int off = it * V;
                            const Pack<V> hv = ld<V>(h + off), rv = ld<V>(r + off), tv = ld<V>(t + off);
                            Pack<V> pv;
#pragma unroll
                            for (int e = 0; e < V; ++e) {
                                const float u = hv.v[e] + rv.v[e] - tv.v[e];
                                pv.v[e] = dp * ((MODEL == KGE_TRANSE_L1) ? sgnf(u) : u * inv);
                            }
                            KGE_ST_OUT<V>(P + off, pv);
                        }
                    }
                }
            }
        }
        if (a.asq) {
            as = wave_sum(as);
            if (lane == 0) a.asq[i] = as;
        }
        if (a.Hc && regres) {
#pragma unroll
            for (int k = 0; k < EK; ++k) {
                if (lane + 64 * k < a.d_e / V) {
                    const int64_t o = i * (int64_t)a.d_e + (lane + 64 * k) * V;
                    st<V>(a.Hc + o, hreg[k]); st<V>(a.Tc + o, treg[k]); st<V>(a.Rc + o, rreg[k]);
                }
            }
        } else if (a.Hc) {      // dense copies of the gathered rows (second pass: the rows are L1 / L2 hot)
            for (int it = lane; it < a.d_e / V; it += 64) {
                st<V>(a.Hc + i * (int64_t)a.d_e + it * V, ld<V>(h + it * V));
                st<V>(a.Tc + i * (int64_t)a.d_e + it * V, ld<V>(t + it * V));
            }
            for (int it = lane; it < a.d_r / V; it += 64) st<V>(a.Rc + i * (int64_t)a.d_r + it * V, ld<V>(r + it * V));
        }
    } else if (w < (int64_t)a.B + a.n_neg) {
        // negative row job: dense copy for the GEMM / pairwise kernels and |b|^2
        const int64_t j = w - a.B;
        int64_t jx = j;
        const int64_t *jidx = a.nidx;
        if (!LEAN && a.nd_own) {         // neg_deg_sample: in-batch rows first, then the sampled ones
            const int Np = a.nd_chunk + a.nd_Ns, c = (int)(j / Np), jj = (int)(j % Np);
            if (jj < a.nd_chunk) { jidx = a.nd_own; jx = (int64_t)c * a.nd_chunk + jj; }
            else jx = (int64_t)c * a.nd_Ns + jj - a.nd_chunk;
        }
        const float *x = table_row(a.src.em, a.nbase, jidx, jx, a.d_e);
        float *cp = a.Bn ? a.Bn + j * (int64_t)a.d_e : nullptr;
        float s = 0.f;
        const int nitn = a.d_e / V;
        if (nitn <= 128) {               // both packs of the row requested together (see the edge job)
            Pack<V> v2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) v2[k] = ld<V>(x + min(lane + 64 * k, nitn - 1) * V);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (lane + 64 * k < nitn) {
                    if (cp) KGE_ST_NEXT<V>(cp + (lane + 64 * k) * V, v2[k]);
#pragma unroll
                    for (int e = 0; e < V; ++e) s += v2[k].v[e] * v2[k].v[e];
                }
            }
        } else if (nitn <= 256) {        // ... and the four packs of a D_e = 800 row (cfg-R; same order of the sum as the loop below)
            Pack<V> v4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v4[k] = ld<V>(x + min(lane + 64 * k, nitn - 1) * V);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (lane + 64 * k < nitn) {
                    if (cp) KGE_ST_NEXT<V>(cp + (lane + 64 * k) * V, v4[k]);
#pragma unroll
                    for (int e = 0; e < V; ++e) s += v4[k].v[e] * v4[k].v[e];
                }
            }
        } else
        for (int it = lane; it < nitn; it += 64) {
            const Pack<V> v = ld<V>(x + it * V);
            if (cp) KGE_ST_NEXT<V>(cp + it * V, v);
#pragma unroll
            for (int e = 0; e < V; ++e) s += v.v[e] * v.v[e];
        }
        if (a.bsq) {
            s = wave_sum(s);
            if (lane == 0) a.bsq[j] = s;
        }
    }
}

