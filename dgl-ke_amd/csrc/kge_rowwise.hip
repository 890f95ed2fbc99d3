// kge_rowwise.hip - the row-wise (one wavefront per embedding row) kernels of the KGE hot path:
//   gather                       ExternalEmbedding.__call__      tensor_models.py:292
//   edge_fwd  (A3 + pos-side)    score_func.edge_func, head of create_neg closures
//                                score_fun.py:54-59,94-105,229-235,270-283,297-307,347-371,
//                                460-472,516-545
//   loss      (A6)               LossGenerator.get_total_loss    loss.py:69-98
//   edge_bwd  (A8)               autograd of the above, analytic
//   update    (A9)               ExternalEmbedding.update        tensor_models.py:304-362
// All of them are HBM/L2-bandwidth work: 16-byte loads per lane, one row per wavefront,
// reductions by cross-lane shuffles, no LDS, no atomics on the fused path.
#include "kge_common.hpp"
#include "kge_update_body.hpp"
#include "kge_sampler_tail.hpp"
#include "kge_edge_fwd_body.hpp"
#include "kge_loss_body.hpp"

using namespace kge;
KGE_TL_DEFINE(rowwise)

#define WAVE_ID() ((int64_t)blockIdx.x * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6))
#define LANE() (threadIdx.x & 63)

static inline int blocks_for_waves(int64_t waves) {
    return (int)((waves + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK);
}
static inline int check_launch() {
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------
// gather
// ------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(KGE_BLOCK) void gather_kernel(const float *__restrict__ table, int dim,
                                                           const int64_t *__restrict__ idx,
                                                           int64_t n, float *__restrict__ out) {
    const int64_t k = WAVE_ID();
    if (k >= n) return;
    const int lane = LANE();
    const float *src = table + idx[k] * (int64_t)dim;
    float *dst = out + k * (int64_t)dim;
    for (int it = lane; it < dim / V; it += 64) st<V>(dst + it * V, ld<V>(src + it * V));
}

int launch_gather_rows(const float *table, int dim, const int64_t *idx, int64_t n, float *out,
                       hipStream_t s) {
    if (n == 0) return KGE_OK;
    const int nb = blocks_for_waves(n);
    if (dim % 4 == 0)
        hipLaunchKernelGGL(gather_kernel<4>, dim3(nb), dim3(KGE_BLOCK), 0, s, table, dim, idx, n, out);
    else
        hipLaunchKernelGGL(gather_kernel<1>, dim3(nb), dim3(KGE_BLOCK), 0, s, table, dim, idx, n, out);
    return check_launch();
}

template <int V>
__global__ __launch_bounds__(KGE_BLOCK) void gather_sharded_kernel(ShardMap m, int dim,
                                                                   const int64_t *__restrict__ idx,
                                                                   int64_t n, float *__restrict__ out) {
    const int64_t k = WAVE_ID();
    if (k >= n) return;
    const int lane = LANE();
    const float *src = shard_row(m, nullptr, idx[k], dim);
    float *dst = out + k * (int64_t)dim;
    for (int it = lane; it < dim / V; it += 64) st<V>(dst + it * V, ld<V>(src + it * V));
}

int launch_gather_rows_sharded(float *const *shard_rows, int n_shards, int64_t per, int dim,
                               const int64_t *idx, int64_t n, float *out, hipStream_t s) {
    if (n == 0) return KGE_OK;
    const int nb = blocks_for_waves(n);
    const ShardMap m{shard_rows, nullptr, per, n_shards};
    if (dim % 4 == 0)
        hipLaunchKernelGGL(gather_sharded_kernel<4>, dim3(nb), dim3(KGE_BLOCK), 0, s, m, dim, idx, n, out);
    else
        hipLaunchKernelGGL(gather_sharded_kernel<1>, dim3(nb), dim3(KGE_BLOCK), 0, s, m, dim, idx, n, out);
    return check_launch();
}

// --neg_deg_sample for the relation-matrix models (round 6; general_models.py:396-402, 417-432).  nd_ids: the combined id list of a
// chunk's negatives, [the chunk's own corrupted-side entities | the sampled ids] (TransR's kernels address the negatives through one
// list).  nd_fold: the gradient of the in-batch negative row of edge i - row (i / chunk) * Np + i % chunk of GN - joins the
// positive-trace gradient row of that entity (GH in head mode, GT in tail mode): the row is a slice of pos_g.ndata['emb'].
__global__ __launch_bounds__(256) void nd_ids_kernel(const int64_t *__restrict__ own, const int64_t *__restrict__ neg_ids, int C, int chunk,
                                                     int Ns, int64_t *__restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int Np = chunk + Ns;
    if (k >= (int64_t)C * Np) return;
    const int c = (int)(k / Np), jj = (int)(k % Np);
    out[k] = jj < chunk ? own[(int64_t)c * chunk + jj] : neg_ids[(int64_t)c * Ns + jj - chunk];
}
int launch_nd_ids(const int64_t *own, const int64_t *neg_ids, int C, int chunk, int Ns, int64_t *out, hipStream_t s) {
    const int64_t n = (int64_t)C * (chunk + Ns);
    if (n == 0) return KGE_OK;
    hipLaunchKernelGGL(nd_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, own, neg_ids, C, chunk, Ns, out);
    return check_launch();
}
__global__ __launch_bounds__(KGE_BLOCK) void nd_fold_kernel(const float *__restrict__ GN, float *__restrict__ G, int B, int chunk, int Np, int d) {
    const int64_t i = WAVE_ID();
    if (i >= B) return;
    const int lane = LANE();
    const float *src = GN + ((i / chunk) * Np + i % chunk) * (int64_t)d;
    float *dst = G + i * (int64_t)d;
    for (int k = lane; k < d; k += 64) dst[k] += src[k];
}
int launch_nd_fold(const float *GN, float *G, int B, int chunk, int Np, int d, hipStream_t s) {
    if (B == 0) return KGE_OK;
    hipLaunchKernelGGL(nd_fold_kernel, dim3(blocks_for_waves(B)), dim3(KGE_BLOCK), 0, s, GN, G, B, chunk, Np, d);
    return check_launch();
}

// TransR / RESCAL on sharded entity tables (round 6): the batch's head, tail and negative rows through the shard map into ONE dense
// block [B | B | n_neg] x dim + the identity index array the kernels of the two families address it with (kge_api.hip sh_dense)
__global__ __launch_bounds__(KGE_BLOCK) void gather3_sharded_kernel(ShardMap m, int dim, const int64_t *__restrict__ h,
                                                                    const int64_t *__restrict__ t, const int64_t *__restrict__ neg,
                                                                    int B, int n_neg, float *__restrict__ out,
                                                                    int64_t *__restrict__ iota) {
    const int64_t k = WAVE_ID();
    if (k >= 2 * (int64_t)B + n_neg) return;
    const int lane = LANE();
    const int64_t id = k < B ? h[k] : (k < 2 * (int64_t)B ? t[k - B] : neg[k - 2 * (int64_t)B]);
    const float *src = shard_row(m, nullptr, id, dim);
    float *dst = out + k * (int64_t)dim;
    for (int it = lane; it < dim / 4; it += 64) st<4>(dst + it * 4, ld<4>(src + it * 4));
    if (lane == 0) iota[k] = k;
}

int launch_gather3_sharded(const ShardMap &m, int dim, const int64_t *h, const int64_t *t, const int64_t *neg, int B, int n_neg,
                           float *out, int64_t *iota, hipStream_t s) {
    if (dim % 4 || m.n < 1) return KGE_ERR_ARG;
    const int64_t n = 2 * (int64_t)B + n_neg;
    if (n == 0) return KGE_OK;
    hipLaunchKernelGGL(gather3_sharded_kernel, dim3(blocks_for_waves(n)), dim3(KGE_BLOCK), 0, s, m, dim, h, t, neg, B, n_neg, out, iota);
    return check_launch();
}

// ------------------------------------------------------------------------------------------
// edge forward: positive score p_i, pos-side vector a_i (and |a_i|^2), |neg_j|^2
// ------------------------------------------------------------------------------------------
// edge forward: body in kge_edge_fwd_body.hpp
template <int MODEL, int V, bool LEAN>
__global__ __launch_bounds__(KGE_BLOCK) void edge_fwd_kernel(EdgeFwdArgs a) {
    KGE_TL(0);
    edge_fwd_body<MODEL, V, LEAN>(a, (int)blockIdx.x);
}

template <int MODEL>
static int launch_edge_fwd_m(const EdgeFwdArgs &a, hipStream_t s) {
    const bool negjob = a.bsq || a.Bn;
    const int64_t waves = (int64_t)a.B + (negjob ? a.n_neg : 0);
    EdgeFwdArgs b = a;
    if (!negjob) b.n_neg = 0;
    const int nb = blocks_for_waves(waves);
    const bool cx = is_complex_model(MODEL);
    const bool vec = cx ? ((a.d_e / 2) % 4 == 0 && a.d_r % 4 == 0) : (a.d_e % 4 == 0);
    const bool lean = vec && a.src.em.n == 0 && a.src.rm.n == 0 && a.lp.genre == KGE_LOSS_LOGSIGMOID && !a.row_pos && !a.Hc &&
                      !a.nd_own;
    if (lean)
        hipLaunchKernelGGL((edge_fwd_kernel<MODEL, 4, true>), dim3(nb), dim3(KGE_BLOCK), 0, s, b);
    else if (vec)
        hipLaunchKernelGGL((edge_fwd_kernel<MODEL, 4, false>), dim3(nb), dim3(KGE_BLOCK), 0, s, b);
    else
        hipLaunchKernelGGL((edge_fwd_kernel<MODEL, 1, false>), dim3(nb), dim3(KGE_BLOCK), 0, s, b);
    return check_launch();
}

int launch_edge_fwd(const EdgeFwdArgs &a, hipStream_t s) {
    if (a.B == 0 && !((a.bsq || a.Bn) && a.n_neg > 0)) return KGE_OK;
    switch (a.model) {
        case KGE_TRANSE_L1: return launch_edge_fwd_m<KGE_TRANSE_L1>(a, s);
        case KGE_TRANSE_L2: return launch_edge_fwd_m<KGE_TRANSE_L2>(a, s);
        case KGE_DISTMULT: return launch_edge_fwd_m<KGE_DISTMULT>(a, s);
        case KGE_COMPLEX: return launch_edge_fwd_m<KGE_COMPLEX>(a, s);
        case KGE_ROTATE: return launch_edge_fwd_m<KGE_ROTATE>(a, s);
        case KGE_SIMPLE: return launch_edge_fwd_m<KGE_SIMPLE>(a, s);
    }
    return KGE_ERR_ARG;
}

// ------------------------------------------------------------------------------------------
// edge backward: per-edge gradients of the head row, the tail row and the relation row
//   = dpos_i * d p_i/d(h,r,t)  +  chain of GA_i = dL/da_i through a_i = T(x_i, r_i)
//   (+ regularisation gradient of the relation row copy)
// ------------------------------------------------------------------------------------------
// wavefronts per edge of edge_bwd (see edge_bwd_body): one per 64 column packs for the column-wise models, vector instances only
__host__ __device__ inline int edge_bwd_wpe(int model, int V, int d_e) {
    if (V != 4) return 1;
    if (model == KGE_ROTATE || model == KGE_COMPLEX) return (d_e / 2 / V + 63) / 64 > 0 ? (d_e / 2 / V + 63) / 64 : 1;
    if (model == KGE_DISTMULT || model == KGE_TRANSE_L1) return (d_e / V + 63) / 64 > 0 ? (d_e / V + 63) / 64 : 1;
    return 1;
}
template <int MODEL, int V, bool LOCAL>       // LOCAL: un-sharded tables (no shard-map divisions compiled in)
__device__ __forceinline__ void edge_bwd_body(const EdgeBwdArgs &a_in, int64_t w_) {
    EdgeBwdArgs a = a_in;
    if constexpr (LOCAL) { a.src.em.n = 0; a.src.rm.n = 0; }
    // wide rows (round 5): an edge's columns are dealt to `wpe` wavefronts, 64 column packs each - the models whose gradient is
    // elementwise in the column (no row norm / clamp: RotatE, ComplEx, DistMult, TransE_l1).  A wavefront looping over 100 packs took
    // one dependent [rows -> arithmetic -> stores] round per 64 of them: 11.0 us of life at D_e = 800 against 6.4 us at 400
    // (tools/timeline.py), the longest chain of the edge_bwd launch.  Same arithmetic per column: bit-identical gradients.
    const int wpe = edge_bwd_wpe(MODEL, V, a.d_e);
    const int64_t i = w_ / wpe;
    const int part = (int)(w_ % wpe);
    if (i >= a.B) return;
    const int lane = LANE();
    const float *h = table_row(a.src.em, a.src.hbase, a.src.hidx, i, a.d_e);
    const float *t = table_row(a.src.em, a.src.tbase, a.src.tidx, i, a.d_e);
    const float *r = table_row(a.src.rm, a.src.rbase, a.src.ridx, i, a.d_r);
    const float dp_in = a.dpos ? a.dpos[i] : 0.f;
    const float *ga = a.GA ? a.GA + i * (int64_t)a.d_e : nullptr;
    // GA in parts (shared-pair backward with split negatives): added in part order
    // (the first four parts are requested together - part index clamped, surplus requests re-read the last part - and added in
    //  part order: the same sum as one dependent load -> add round per part)
    auto ld_ga = [&](const float *p) {
        const int np = a.ga_parts > 1 ? a.ga_parts : 1;
        Pack<V> u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = ld<V>(p + (int64_t)(q < np ? q : np - 1) * a.ga_stride);
        Pack<V> v = u[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            if (q < np) {
#pragma unroll
                for (int e = 0; e < V; ++e) v.v[e] += u[q].v[e];
            }
        }
        for (int q = 4; q < np; ++q) {
            const Pack<V> w = ld<V>(p + q * a.ga_stride);
#pragma unroll
            for (int e = 0; e < V; ++e) v.v[e] += w.v[e];
        }
        return v;
    };
    float *GH = a.GH ? a.GH + i * (int64_t)a.d_e : nullptr;
    float *GT = a.GT ? a.GT + i * (int64_t)a.d_e : nullptr;
    float *GR = a.GR ? a.GR + i * (int64_t)a.d_r : nullptr;
    const bool reg = a.reg_coef > 0.f && a.reg_norm > 0;
    // neg_deg_sample: the in-batch negative row of this edge feeds the corrupted side (general_models.py:396-402)
    const float *gnd = a.GNd ? a.GNd + ((i / a.nd_chunk) * a.nd_Np + i % a.nd_chunk) * (int64_t)a.d_e : nullptr;

    if constexpr (!is_complex_model(MODEL)) {
        const int nit = a.d_e / V;
        const float dp = dp_in;
        float inv = 0.f;
        if constexpr (MODEL == KGE_TRANSE_L2) {
            if (a.dpos) {   // |h + r - t|_2 for the positive-score gradient
                float ss = 0.f;
                for (int it = lane; it < nit; it += 64) {
                    const int off = it * V;
                    const Pack<V> hv = ld<V>(h + off), rv = ld<V>(r + off), tv = ld<V>(t + off);
#pragma unroll
                    for (int e = 0; e < V; ++e) { const float u = hv.v[e] + rv.v[e] - tv.v[e]; ss += u * u; }
                }
                ss = sqrtf(wave_sum(ss));
                inv = ss > 0.f ? 1.f / ss : 0.f;
            }
        }
        for (int it = part * 64 + lane; it < nit; it += 64 * wpe) {
            const int off = it * V;
            const Pack<V> hv = ld<V>(h + off), rv = ld<V>(r + off), tv = ld<V>(t + off);
            const Pack<V> gav = ga ? ld_ga(ga + off) : zero_pack<V>();
            Pack<V> gh, gt, gr;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float hh = hv.v[e], rr = rv.v[e], tt = tv.v[e], g = gav.v[e];
                float o_h, o_t, o_r;
                if constexpr (MODEL == KGE_DISTMULT) {
                    o_h = dp * rr * tt; o_r = dp * hh * tt; o_t = dp * hh * rr;
                    if (a.neg_head) { o_t += g * rr; o_r += g * tt; }
                    else            { o_h += g * rr; o_r += g * hh; }
                } else {
                    const float u = hh + rr - tt;
                    const float d = (MODEL == KGE_TRANSE_L1) ? sgnf(u) : u * inv;
                    o_h = -dp * d; o_r = -dp * d; o_t = dp * d;
                    if (a.neg_head) { o_t += g; o_r -= g; }
                    else            { o_h += g; o_r += g; }
                }
                if (reg) o_r += reg_grad(rr, a.reg_coef, a.reg_norm);
                gh.v[e] = o_h; gt.v[e] = o_t; gr.v[e] = o_r;
            }
            if (gnd) {
                const Pack<V> gv = ld<V>(gnd + off);
#pragma unroll
                for (int e = 0; e < V; ++e) { if (a.neg_head) gh.v[e] += gv.v[e]; else gt.v[e] += gv.v[e]; }
            }
            if (GH) st_wt<V>(GH + off, gh);
            if (GT) st_wt<V>(GT + off, gt);
            if (GR) st_wt<V>(GR + off, gr);
        }
    } else {
        const int hd = a.d_e / 2;
        const int nit = hd / V;
        float dp = dp_in;
        if constexpr (MODEL == KGE_SIMPLE) {
            if (a.clamp_pos && a.dpos) {
                float ps = 0.f;
                for (int it = lane; it < nit; it += 64) {
                    const int off = it * V;
                    const Pack<V> hi = ld<V>(h + off), hj = ld<V>(h + hd + off), ti = ld<V>(t + off), tj = ld<V>(t + hd + off);
                    const Pack<V> c = ld<V>(r + off), sv = ld<V>(r + hd + off);
#pragma unroll
                    for (int e = 0; e < V; ++e) ps += 0.5f * (hi.v[e] * c.v[e] * tj.v[e] + ti.v[e] * sv.v[e] * hj.v[e]);
                }
                if (fabsf(wave_sum(ps)) >= KGE_SIMPLE_CLAMP) dp = 0.f;
            }
        }
        for (int it = part * 64 + lane; it < nit; it += 64 * wpe) {
            const int off = it * V;
            const Pack<V> rh = ld<V>(h + off), ih = ld<V>(h + hd + off);
            const Pack<V> rt = ld<V>(t + off), it_ = ld<V>(t + hd + off);
            const Pack<V> gre = ga ? ld_ga(ga + off) : zero_pack<V>();
            const Pack<V> gim = ga ? ld_ga(ga + hd + off) : zero_pack<V>();
            Pack<V> o_rh, o_ih, o_rt, o_it;
            if constexpr (MODEL == KGE_SIMPLE) {
                const Pack<V> rr = ld<V>(r + off), ir = ld<V>(r + hd + off);
                Pack<V> o_rr, o_ir;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    // p = 1/2 sum(h_i r t_j + t_i r_inv h_j); a = 1/2 [r_inv h_j | h_i r] (tail mode) or
                    // 1/2 [r t_j | r_inv t_i] (head mode); (gre, gim) = dL/da halves
                    const float hi = rh.v[e], hj = ih.v[e], ti = rt.v[e], tj = it_.v[e];
                    const float c = rr.v[e], s = ir.v[e], g1 = 0.5f * gre.v[e], g2 = 0.5f * gim.v[e], d = 0.5f * dp;
                    float v_hi = d * c * tj, v_hj = d * ti * s, v_ti = d * s * hj, v_tj = d * hi * c;
                    float v_r = d * hi * tj, v_ri = d * ti * hj;
                    if (a.neg_head) { v_tj += g1 * c; v_r += g1 * tj; v_ti += g2 * s; v_ri += g2 * ti; }
                    else            { v_hj += g1 * s; v_ri += g1 * hj; v_hi += g2 * c; v_r += g2 * hi; }
                    if (reg) {
                        v_r += reg_grad(c, a.reg_coef, a.reg_norm);
                        v_ri += reg_grad(s, a.reg_coef, a.reg_norm);
                    }
                    o_rh.v[e] = v_hi; o_ih.v[e] = v_hj; o_rt.v[e] = v_ti; o_it.v[e] = v_tj;
                    o_rr.v[e] = v_r; o_ir.v[e] = v_ri;
                }
                if (GR) { st_wt<V>(GR + off, o_rr); st_wt<V>(GR + hd + off, o_ir); }
            } else if constexpr (MODEL == KGE_COMPLEX) {
                const Pack<V> rr = ld<V>(r + off), ir = ld<V>(r + hd + off);
                Pack<V> o_rr, o_ir;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float a_rh = rh.v[e], a_ih = ih.v[e], a_rt = rt.v[e], a_it = it_.v[e];
                    const float c = rr.v[e], s = ir.v[e], gr_ = gre.v[e], gi_ = gim.v[e];
                    float v_rh = dp * (a_rt * c + a_it * s);
                    float v_ih = dp * (a_it * c - a_rt * s);
                    float v_rt = dp * (a_rh * c - a_ih * s);
                    float v_it = dp * (a_ih * c + a_rh * s);
                    float v_rr = dp * (a_rh * a_rt + a_ih * a_it);
                    float v_ir = dp * (a_rh * a_it - a_ih * a_rt);
                    if (a.neg_head) {   // a = t o conj(r)
                        v_rt += gr_ * c - gi_ * s;
                        v_it += gr_ * s + gi_ * c;
                        v_rr += gr_ * a_rt + gi_ * a_it;
                        v_ir += gr_ * a_it - gi_ * a_rt;
                    } else {            // a = h o r
                        v_rh += gr_ * c + gi_ * s;
                        v_ih += -gr_ * s + gi_ * c;
                        v_rr += gr_ * a_rh + gi_ * a_ih;
                        v_ir += -gr_ * a_ih + gi_ * a_rh;
                    }
                    if (reg) {
                        v_rr += reg_grad(c, a.reg_coef, a.reg_norm);
                        v_ir += reg_grad(s, a.reg_coef, a.reg_norm);
                    }
                    o_rh.v[e] = v_rh; o_ih.v[e] = v_ih; o_rt.v[e] = v_rt; o_it.v[e] = v_it;
                    o_rr.v[e] = v_rr; o_ir.v[e] = v_ir;
                }
                if (GR) { st_wt<V>(GR + off, o_rr); st_wt<V>(GR + hd + off, o_ir); }
            } else {   // RotatE
                const Pack<V> ph = ld<V>(r + off);
                Pack<V> o_r;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    float s, c;
                    sincosf(ph.v[e] / a.rot_div, &s, &c);
                    const float a_rh = rh.v[e], a_ih = ih.v[e], a_rt = rt.v[e], a_it = it_.v[e];
                    const float re = a_rh * c - a_ih * s - a_rt;
                    const float im = a_rh * s + a_ih * c - a_it;
                    const float m = sqrtf(re * re + im * im);
                    const float iv = m > 0.f ? 1.f / m : 0.f;
                    // p = gamma - sum m  ->  upstream on the rotated head is -dp*(re,im)/m
                    const float ure = -dp * re * iv, uim = -dp * im * iv;
                    float v_rh = ure * c + uim * s;
                    float v_ih = -ure * s + uim * c;
                    float v_rt = -ure, v_it = -uim;
                    float gphi = ure * (-a_rh * s - a_ih * c) + uim * (a_rh * c - a_ih * s);
                    const float gr_ = gre.v[e], gi_ = gim.v[e];
                    if (a.neg_head) {   // a = t o e^{-i phi}
                        v_rt += gr_ * c - gi_ * s;
                        v_it += gr_ * s + gi_ * c;
                        gphi += gr_ * (-a_rt * s + a_it * c) + gi_ * (-a_rt * c - a_it * s);
                    } else {            // a = h o e^{i phi}
                        v_rh += gr_ * c + gi_ * s;
                        v_ih += -gr_ * s + gi_ * c;
                        gphi += gr_ * (-a_rh * s - a_ih * c) + gi_ * (a_rh * c - a_ih * s);
                    }
                    float v_r = gphi / a.rot_div;
                    if (reg) v_r += reg_grad(ph.v[e], a.reg_coef, a.reg_norm);
                    o_rh.v[e] = v_rh; o_ih.v[e] = v_ih; o_rt.v[e] = v_rt; o_it.v[e] = v_it;
                    o_r.v[e] = v_r;
                }
                if (GR) st_wt<V>(GR + off, o_r);
            }
            if (gnd) {
                const Pack<V> gv0 = ld<V>(gnd + off), gv1 = ld<V>(gnd + hd + off);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    if (a.neg_head) { o_rh.v[e] += gv0.v[e]; o_ih.v[e] += gv1.v[e]; }
                    else { o_rt.v[e] += gv0.v[e]; o_it.v[e] += gv1.v[e]; }
                }
            }
            if (GH) { st_wt<V>(GH + off, o_rh); st_wt<V>(GH + hd + off, o_ih); }
            if (GT) { st_wt<V>(GT + off, o_rt); st_wt<V>(GT + hd + off, o_it); }
        }
    }
}

template <int MODEL, int V, bool LOCAL>
__global__ __launch_bounds__(KGE_BLOCK) void edge_bwd_kernel(EdgeBwdArgs a) {
    KGE_TL(5);
    edge_bwd_body<MODEL, V, LOCAL>(a, WAVE_ID());
}

// RotatE / TransE_l1 pairwise path: the per-edge gradient rows and the sum of the shared-pair backward's GN partials
// (kge_neg_bcast.hip) depend on the same launch and not on each other: ONE launch, the first nbE workgroups run edge_bwd, the
// rest the reduction (round 2: two launches, 6.0 + 5.4 us + a boundary at the RotatE FB15k shape)
template <int MODEL, int V, bool LOCAL>
__global__ __launch_bounds__(KGE_BLOCK) void edge_bwd_gnred_kernel(EdgeBwdArgs a, NegArgs n, int nrw, int nbE) {
    if ((int)blockIdx.x < nbE) {
        KGE_TL(5);
        edge_bwd_body<MODEL, V, LOCAL>(a, WAVE_ID());
    } else {
        KGE_TL(6);
        gn_reduce_body(n, nrw, ((int64_t)blockIdx.x - nbE) * KGE_BLOCK + threadIdx.x);
    }
}

template <int MODEL>
static int launch_edge_bwd_gnred_m(const EdgeBwdArgs &a, const NegArgs &n, int nrw, hipStream_t s) {
    const int nbE = blocks_for_waves((int64_t)a.B * edge_bwd_wpe(MODEL, 4, a.d_e));      // (vector instances only, checked below)
    const int64_t n4 = (int64_t)n.C * n.N * n.d_e / 4;
    const int nbR = (int)((n4 + KGE_BLOCK - 1) / KGE_BLOCK);
    const bool cx = is_complex_model(MODEL);
    const bool vec = cx ? ((a.d_e / 2) % 4 == 0 && a.d_r % 4 == 0) : (a.d_e % 4 == 0);
    const bool local = a.src.em.n == 0 && a.src.rm.n == 0;
    if (!vec) return KGE_ERR_ARG;
    if (local) hipLaunchKernelGGL((edge_bwd_gnred_kernel<MODEL, 4, true>), dim3(nbE + nbR), dim3(KGE_BLOCK), 0, s, a, n, nrw, nbE);
    else hipLaunchKernelGGL((edge_bwd_gnred_kernel<MODEL, 4, false>), dim3(nbE + nbR), dim3(KGE_BLOCK), 0, s, a, n, nrw, nbE);   // sharded tables (round 4)
    return check_launch();
}

int launch_edge_bwd_with_gn_reduce(const EdgeBwdArgs &a, const NegArgs &n, int nrw, hipStream_t s) {
    if (a.B == 0 || !n.GNp || !n.GN || nrw < 1 || n.d_e % 4) return KGE_ERR_ARG;
    switch (a.model) {
        case KGE_ROTATE: return launch_edge_bwd_gnred_m<KGE_ROTATE>(a, n, nrw, s);
        case KGE_TRANSE_L1: return launch_edge_bwd_gnred_m<KGE_TRANSE_L1>(a, n, nrw, s);
    }
    return KGE_ERR_ARG;
}

template <int MODEL>
static int launch_edge_bwd_m(const EdgeBwdArgs &a, hipStream_t s) {
    const bool cx = is_complex_model(MODEL);
    const bool vec = cx ? ((a.d_e / 2) % 4 == 0 && a.d_r % 4 == 0) : (a.d_e % 4 == 0);
    const int nb = blocks_for_waves((int64_t)a.B * (vec ? edge_bwd_wpe(MODEL, 4, a.d_e) : 1));
    const bool local = a.src.em.n == 0 && a.src.rm.n == 0;
    if (vec && local)
        hipLaunchKernelGGL((edge_bwd_kernel<MODEL, 4, true>), dim3(nb), dim3(KGE_BLOCK), 0, s, a);
    else if (vec)
        hipLaunchKernelGGL((edge_bwd_kernel<MODEL, 4, false>), dim3(nb), dim3(KGE_BLOCK), 0, s, a);
    else
        hipLaunchKernelGGL((edge_bwd_kernel<MODEL, 1, false>), dim3(nb), dim3(KGE_BLOCK), 0, s, a);
    return check_launch();
}

int launch_edge_bwd(const EdgeBwdArgs &a, hipStream_t s) {
    if (a.B == 0) return KGE_OK;
    switch (a.model) {
        case KGE_TRANSE_L1: return launch_edge_bwd_m<KGE_TRANSE_L1>(a, s);
        case KGE_TRANSE_L2: return launch_edge_bwd_m<KGE_TRANSE_L2>(a, s);
        case KGE_DISTMULT: return launch_edge_bwd_m<KGE_DISTMULT>(a, s);
        case KGE_COMPLEX: return launch_edge_bwd_m<KGE_COMPLEX>(a, s);
        case KGE_ROTATE: return launch_edge_bwd_m<KGE_ROTATE>(a, s);
        case KGE_SIMPLE: return launch_edge_bwd_m<KGE_SIMPLE>(a, s);
    }
    return KGE_ERR_ARG;
}

// ------------------------------------------------------------------------------------------
// loss + d loss / d score.  One wavefront per positive edge (row of the [B,N] negative scores).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(KGE_BLOCK) void loss_kernel(LossArgs a) {
    const int64_t i = WAVE_ID();
    if (i >= a.B) return;
    const int lane = LANE();
    const int N = a.N;
    const float *n = a.neg + i * (int64_t)N;
    float *dn = a.dneg + i * (int64_t)N;
    float *cp = a.neg_copy ? a.neg_copy + i * (int64_t)N : nullptr;
    const float w = a.w ? a.w[i] : 1.f;
    const float p = a.pos[i];
    const float invB = 1.f / (float)a.B;
    // neg_deg_sample: the positive edge itself sits in column i % chunk - score 0, no gradient (general_models.py:401, 429-432)
    const int jd = a.diag_chunk > 0 ? (int)(i % a.diag_chunk) : -1;
    if (a.l2_raw) {     // merged forward launch: the row holds raw products - rebuild the scores in place first (score_fun.py:26-34)
        const float as = a.asq[i];
        const float *bq = a.bsq + (i / a.l2_chunk) * (int64_t)N;
        float *nw = const_cast<float *>(n);
        for (int j = lane; j < N; j += 64) nw[j] = a.gamma - sqrtf(fmaxf(fmaf(-2.f, n[j], as + bq[j]), 1e-30f));
        // (a lane re-reads only what it wrote itself: columns j = lane mod 64)
    }
    if (a.pairwise) {   // loss.py:76-80
        const float sc = w / ((float)a.B * (float)N);
        float lsum = 0.f, dsum = 0.f;
        for (int j = lane; j < N; j += 64) {
            const float nv = j == jd ? 0.f : n[j];
            float val, dv;
            criterion(a.genre, p - nv, 1.f, a.margin, val, dv);
            lsum += val * sc;
            const float dd = dv * sc;
            dsum += dd;
            if (cp) cp[j] = nv;
            float g = -dd;
            if (a.l2_scale) { const float d = a.gamma - nv; g = d > 1e-15f ? g / d : 0.f; }
            if (a.clampv > 0.f && fabsf(nv) >= a.clampv) g = 0.f;
            dn[j] = j == jd ? 0.f : g;
        }
        lsum = wave_sum(lsum);
        dsum = wave_sum(dsum);
        if (lane == 0) {
            a.dpos[i] = (a.clampv > 0.f && fabsf(p) >= a.clampv) ? 0.f : dsum;
            if (a.row_pos) { a.row_pos[i] = 0.f; a.row_neg[i] = lsum; }
            if (a.acc) acc_add(&a.acc[2 * KGE_ACC_SLOTS + (int)(i & (KGE_ACC_SLOTS - 1))], lsum, a.B <= KGE_ACC_SLOTS);
        }
        return;
    }
    float plw = 0.f;
    if (!a.skip_pos) {
        const float wm = mean_edge_weight(a.w, a.B, lane, a.w_mean);     // positive part: the batch's MEAN importance (kge_common.hpp)
        if (lane == 0) {
            float pl, dpl;
            criterion(a.genre, p, 1.f, a.margin, pl, dpl);
            a.dpos[i] = (a.clampv > 0.f && fabsf(p) >= a.clampv) ? 0.f : dpl * wm * 0.5f * invB;
            plw = pl * wm * invB;
            if (a.row_pos) a.row_pos[i] = plw;
        }
    }
    const float neg_label = a.genre == KGE_LOSS_BCE ? 0.f : -1.f;
    float mx = -INFINITY, Z = 1.f;
    if (a.adv) {   // softmax(neg * T) over the row, detached (loss.py:87-88)
        for (int j = lane; j < N; j += 64) mx = fmaxf(mx, (j == jd ? 0.f : n[j]) * a.adv_temp);
        mx = wave_max(mx);
        float z = 0.f;
        for (int j = lane; j < N; j += 64) z += expf((j == jd ? 0.f : n[j]) * a.adv_temp - mx);
        Z = wave_sum(z);
    }
    const float invZ = 1.f / Z, invN = 1.f / (float)N;
    float acc = 0.f;
    for (int j = lane; j < N; j += 64) {
        const float nv = j == jd ? 0.f : n[j];
        float nl, dnl;
        criterion(a.genre, nv, neg_label, a.margin, nl, dnl);
        const float A = a.adv ? expf(nv * a.adv_temp - mx) * invZ : invN;
        acc += A * nl * w;
        float g = dnl * w * A * 0.5f * invB;
        if (cp) cp[j] = nv;
        if (a.l2_scale) { const float d = a.gamma - nv; g = d > 1e-15f ? g / d : 0.f; }
        if (a.clampv > 0.f && fabsf(nv) >= a.clampv) g = 0.f;
        dn[j] = j == jd ? 0.f : g;
    }
    acc = wave_sum(acc) * invB;
    if (lane == 0) {
        if (a.row_neg) a.row_neg[i] = acc;
        if (a.acc) {
            const int slot = (int)(i & (KGE_ACC_SLOTS - 1));
            const bool uq = a.B <= KGE_ACC_SLOTS;
            if (!a.skip_pos) acc_add(&a.acc[0 * KGE_ACC_SLOTS + slot], plw, uq);
            acc_add(&a.acc[1 * KGE_ACC_SLOTS + slot], acc, uq);
            acc_add(&a.acc[2 * KGE_ACC_SLOTS + (int)((a.skip_pos ? i + a.B : i) & (KGE_ACC_SLOTS - 1))], 0.5f * (plw + acc), uq);
        }
    }
}

// register-resident variant: the whole score row (N <= 64*NPER) is loaded once.
// LEAN: the common configuration (Logsigmoid, point-wise, positive part done by edge_fwd, no score clamp) with
// every other loss genre / option compiled out - the generic instantiation carries NPER copies of a three-way
// criterion switch, the pairwise variant and the option handling (2.1 k instructions vs ~0.7 k)
// RAW: the row holds the raw products a_i . b_j of the merged forward launch (LossArgs::l2_raw): the TransE_l2 score
// gamma - sqrt(|a_i|^2 + |b_j|^2 - 2 a_i.b_j) is rebuilt here, from one more coalesced row of |b|^2 requested with the scores
// (the arithmetic on the register-resident row lives in kge_loss_body.hpp: the forward tiles' last arriver runs the same code)
template <int NPER, bool LEAN, bool RAW>
__global__ __launch_bounds__(KGE_BLOCK) void loss_kernel_reg(LossArgs a_in) {
    KGE_TL(2);
    LossArgs a = a_in;
    if constexpr (LEAN) loss_args_lean(a);
    const int64_t i = WAVE_ID();
    if (i >= a.B) return;
    const int lane = LANE();
    const int N = a.N;
    const float *n = a.neg + i * (int64_t)N;
    float nv[NPER];
    // neg_deg_sample: the positive edge itself sits in column i % chunk - score 0, no gradient
    const int jd = a.diag_chunk > 0 ? (int)(i % a.diag_chunk) : -1;
#pragma unroll
    for (int u = 0; u < NPER; ++u) { const int j = lane + 64 * u; nv[u] = (j < N && j != jd) ? n[j] : 0.f; }
    float bq[RAW ? NPER : 1], asq_i = 0.f;
    if constexpr (RAW) {
        const float *bqp = a.bsq + (i / a.l2_chunk) * (int64_t)N;
#pragma unroll
        for (int u = 0; u < NPER; ++u) { const int j = lane + 64 * u; bq[u] = j < N ? bqp[j] : 0.f; }
        asq_i = a.asq[i];
    }
    const float w = a.w ? a.w[i] : 1.f;
    const float p = a.pos[i];
    if constexpr (RAW) {
#pragma unroll
        for (int u = 0; u < NPER; ++u) {
            const int j = lane + 64 * u;
            nv[u] = (j < N && j != jd) ? a.gamma - sqrtf(fmaxf(fmaf(-2.f, nv[u], asq_i + bq[u]), 1e-30f)) : 0.f;
        }
    }
#ifdef KGE_TL_MARKS
    KGE_TL_MARK(0);              // score row, positive score, weight have arrived
#endif
    // (the row's share of the running total: when the positive share comes from edge_fwd, B slots further on than edge_fwd's add -
    //  the same two adds as the strict step's in-launch loss rows, whose other half runs in the SAME launch)
    loss_row_regs<NPER>(a, i, nv, w, p, lane, (int)((a.skip_pos ? i + a.B : i) & (KGE_ACC_SLOTS - 1)));
#ifdef KGE_TL_MARKS
    KGE_TL_MARK(1);              // softmax, criterion, reductions done; gradient stores acknowledged
#endif
}

int launch_loss(const LossArgs &a, hipStream_t s) {
    if (a.B == 0) return KGE_OK;
    const dim3 g(blocks_for_waves(a.B)), b(KGE_BLOCK);
    const bool lean = a.genre == KGE_LOSS_LOGSIGMOID && !a.pairwise && a.skip_pos && a.clampv == 0.f && !a.neg_copy &&
                      !a.row_pos && !a.row_neg && a.diag_chunk <= 0;
    if (a.l2_raw && (!a.asq || !a.bsq || a.l2_chunk <= 0)) return KGE_ERR_ARG;
#define KGE_LOSS(N) do { if (a.l2_raw) { if (lean) hipLaunchKernelGGL((loss_kernel_reg<N, true, true>), g, b, 0, s, a); \
                                         else hipLaunchKernelGGL((loss_kernel_reg<N, false, true>), g, b, 0, s, a); } \
                         else if (lean) hipLaunchKernelGGL((loss_kernel_reg<N, true, false>), g, b, 0, s, a); \
                         else hipLaunchKernelGGL((loss_kernel_reg<N, false, false>), g, b, 0, s, a); } while (0)
    if (a.N <= 64) KGE_LOSS(1);
    else if (a.N <= 128) KGE_LOSS(2);
    else if (a.N <= 256) KGE_LOSS(4);
    else if (a.N <= 512) KGE_LOSS(8);
    else hipLaunchKernelGGL(loss_kernel, g, b, 0, s, a);
#undef KGE_LOSS
    return check_launch();
}

// deterministic final reduction of the per-row loss terms (fixed summation order)
__device__ float block_sum_det(const float *x, int n, float *sh) {
    float v = 0.f;
    if (x) for (int k = threadIdx.x; k < n; k += KGE_BLOCK) v += x[k];
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(KGE_BLOCK) void finalize_kernel(FinalizeArgs a) {
    __shared__ float sh[KGE_WAVES_PER_BLOCK];
    const float lp = block_sum_det(a.row_pos, a.B, sh);
    const float ln = block_sum_det(a.row_neg, a.B, sh);
    const float re = block_sum_det(a.reg_ent, a.counts_dev ? a.counts_dev[0] : a.UE, sh);
    const float rr = block_sum_det(a.reg_rel, a.counts_dev ? a.counts_dev[1] : a.UR, sh);
    if (threadIdx.x == 0) {
        float o[4];
        if (a.pairwise) { o[0] = NAN; o[1] = NAN; o[2] = ln; }
        else { o[0] = lp; o[1] = ln; o[2] = 0.5f * (lp + ln); }
        o[3] = re + rr;
        if (a.loss4) for (int k = 0; k < 4; ++k) a.loss4[k] = o[k];
    }
}

// reduce the running-sum slots to 4 scalars {pos_loss, neg_loss, loss, regularisation} (sums over
// all steps since the last reset; the host divides by the step count)
__global__ __launch_bounds__(KGE_BLOCK) void reduce_acc_kernel(float *acc, float *out4, int zero_after) {
    __shared__ float sh[KGE_WAVES_PER_BLOCK];
    for (int r = 0; r < 4; ++r) {
        const float v = block_sum_det(acc + r * KGE_ACC_SLOTS, KGE_ACC_SLOTS, sh);
        if (threadIdx.x == 0) out4[r] = v;
    }
    if (zero_after) {
        __syncthreads();
        for (int k = threadIdx.x; k < 4 * KGE_ACC_SLOTS; k += KGE_BLOCK) acc[k] = 0.f;
    }
}

int launch_reduce_acc(float *acc, float *out4, int zero_after, hipStream_t s) {
    hipLaunchKernelGGL(reduce_acc_kernel, dim3(1), dim3(KGE_BLOCK), 0, s, acc, out4, zero_after);
    return check_launch();
}

int launch_finalize(const FinalizeArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(KGE_BLOCK), 0, s, a);
    return check_launch();
}

// ------------------------------------------------------------------------------------------
// owner-computes row-sparse Adagrad (fused path).  One wavefront owns one table row, gathers
// every gradient contribution of that row in a fixed order, and applies the reference's
// sequential trace semantics (tensor_models.py:316-361):
//   entity row x:  trace 0 (pos-unique):  s0 = state + mean(g0^2);   x += (-lr*g0)/(sqrt(s0)+eps)
//                  trace 1 (negatives):   s1 = s0 + sum_k mean(gk^2); x += (-lr*gk)/(sqrt(s1)+eps)
//   relation row:  one trace with duplicates, same rule.
// No atomics, bit-reproducible.
// ------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(KGE_BLOCK) void update_kernel(UpdateArgs a, int nb_ent) {
    const int lane = LANE();
    const bool reg = a.reg_coef > 0.f && a.reg_norm > 0;
    if ((int)blockIdx.x < nb_ent) {
        const int64_t u = WAVE_ID();
        if (u >= (a.counts_dev ? a.counts_dev[0] : a.UE)) return;
        const int d = a.model_d_e;
        const int64_t id = a.ue_id[u];
        float *row = shard_row(a.em, a.ent, id, d);
        float *srow = shard_state(a.em, a.ent_state, id);
        const int p0 = a.ue_pos_ptr[u], p1 = a.ue_pos_ptr[u + 1];
        const int n0 = a.ue_neg_ptr[u], n1 = a.ue_neg_ptr[u + 1];
        const bool has_pos = p1 > p0, has_neg = n1 > n0;
        const int nit = d / V;
        float s0 = 0.f, s1 = 0.f, rv = 0.f;
        for (int it = lane; it < nit; it += 64) {
            const int off = it * V;
            const Pack<V> x = ld<V>(row + off);
            Pack<V> g0 = zero_pack<V>();
            if (reg) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    rv += reg_val(x.v[e], a.reg_norm);
                    g0.v[e] = reg_grad(x.v[e], a.reg_coef, a.reg_norm);
                }
            }
            if (has_pos) {
                for (int p = p0; p < p1; ++p) {
                    const int adj = a.ue_pos_adj[p];
                    const float *src = ((adj & 1) ? a.GT : a.GH) + (int64_t)(adj >> 1) * d + off;
                    const Pack<V> g = ld<V>(src);
#pragma unroll
                    for (int e = 0; e < V; ++e) g0.v[e] += g.v[e];
                }
#pragma unroll
                for (int e = 0; e < V; ++e) s0 += g0.v[e] * g0.v[e];
            }
            for (int k = n0; k < n1; ++k) {
                Pack<V> g = ld<V>(a.GN + gn_row(a, a.ue_neg_slot[k]) * d + off);
                if (a.nd_chunk && reg) {      // neg_deg_sample: the scoring kernels leave the regulariser to this kernel
#pragma unroll
                    for (int e = 0; e < V; ++e) g.v[e] += reg_grad(x.v[e], a.reg_coef, a.reg_norm);
                }
#pragma unroll
                for (int e = 0; e < V; ++e) s1 += g.v[e] * g.v[e];
            }
        }
        s0 = wave_sum(s0) / (float)d;
        s1 = wave_sum(s1) / (float)d;
        const float st0 = *srow;
        const float sA = has_pos ? st0 + s0 : st0;
        const float sB = has_neg ? sA + s1 : sA;
        const float std0 = sqrtf(sA) + a.eps, std1 = sqrtf(sB) + a.eps;
        const int64_t mu = a.emit_by_id ? id : u;       // message row: union entry, or the row id itself (cache rows, dist.py)
        for (int it = lane; it < nit; it += 64) {
            const int off = it * V;
            Pack<V> x = ld<V>(row + off);
            const Pack<V> x0 = x;             // the row as it was gathered (regulariser of the negative rows, nd mode)
            Pack<V> g0 = zero_pack<V>(), g1 = zero_pack<V>();
            if (has_pos) {
                if (reg) {
#pragma unroll
                    for (int e = 0; e < V; ++e) g0.v[e] = reg_grad(x.v[e], a.reg_coef, a.reg_norm);
                }
                for (int p = p0; p < p1; ++p) {
                    const int adj = a.ue_pos_adj[p];
                    const float *src = ((adj & 1) ? a.GT : a.GH) + (int64_t)(adj >> 1) * d + off;
                    const Pack<V> g = ld<V>(src);
#pragma unroll
                    for (int e = 0; e < V; ++e) g0.v[e] += g.v[e];
                }
#pragma unroll
                for (int e = 0; e < V; ++e) x.v[e] = fmaf(g0.v[e], -a.lr / std0, x.v[e]);
            }
            for (int k = n0; k < n1; ++k) {
                Pack<V> g = ld<V>(a.GN + gn_row(a, a.ue_neg_slot[k]) * d + off);
                if (a.nd_chunk && reg) {
#pragma unroll
                    for (int e = 0; e < V; ++e) g.v[e] += reg_grad(x0.v[e], a.reg_coef, a.reg_norm);
                }
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    x.v[e] = fmaf(g.v[e], -a.lr / std1, x.v[e]);
                    g1.v[e] += g.v[e];
                }
            }
            if (!a.emit_ent) st<V>(row + off, x);
            if (a.g0) st<V>(a.g0 + mu * (int64_t)a.ld_e + off, g0);
            if (a.g1) st<V>(a.g1 + mu * (int64_t)a.ld_e + off, g1);
        }
        if (lane == 0) {
            if (!a.emit_ent) *srow = sB;
            if (a.gs0) a.gs0[mu * (int64_t)a.ld_gs_e] = has_pos ? s0 : 0.f;
            if (a.gs1) a.gs1[mu * (int64_t)a.ld_gs_e] = has_neg ? s1 : 0.f;
        }
        if (reg && (a.reg_ent || a.acc)) {
            rv = wave_sum(rv);
            const float val = a.reg_coef * rv * (float)((has_pos ? 1 : 0) + (n1 - n0));
            if (lane == 0) {
                if (a.reg_ent) a.reg_ent[u] = val;
                if (a.acc) acc_add(&a.acc[3 * KGE_ACC_SLOTS + (int)(u & (KGE_ACC_SLOTS - 1))], val, a.UE + a.UR <= KGE_ACC_SLOTS);
            }
        } else if (a.reg_ent && lane == 0) a.reg_ent[u] = 0.f;
    } else {
        const int64_t u = ((int64_t)blockIdx.x - nb_ent) * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
        if (u >= (a.counts_dev ? a.counts_dev[1] : a.UR)) {
            if (a.rid && u < a.UR && lane == 0) { a.rid[u * (int64_t)a.ld_r] = -1; a.rid[u * (int64_t)a.ld_r + 1] = -1; }   // pad message
            return;
        }
        const int d = a.d_r;
        const int64_t id = a.ur_id[u];
        float *row = shard_row(a.rm, a.rel, id, d);
        float *srow = shard_state(a.rm, a.rel_state, id);
        const int e0 = a.ur_ptr[u], e1 = a.ur_ptr[u + 1];
        const int nit = d / V;
        float ss = 0.f, rv = 0.f;
        for (int it = lane; it < nit; it += 64) {
            const int off = it * V;
            if (reg && (a.reg_rel || a.acc)) {
                const Pack<V> x = ld<V>(row + off);
#pragma unroll
                for (int e = 0; e < V; ++e) rv += reg_val(x.v[e], a.reg_norm);
            }
            for (int k = e0; k < e1; ++k) {
                const Pack<V> g = ld<V>(a.GR + (int64_t)a.ur_edge[k] * d + off);
#pragma unroll
                for (int e = 0; e < V; ++e) ss += g.v[e] * g.v[e];
            }
        }
        ss = wave_sum(ss) / (float)d;
        const float sN = *srow + ss;
        const float sd = sqrtf(sN) + a.eps;
        for (int it = lane; it < nit; it += 64) {
            const int off = it * V;
            Pack<V> x = ld<V>(row + off);
            Pack<V> gsum = zero_pack<V>();
            for (int k = e0; k < e1; ++k) {
                const Pack<V> g = ld<V>(a.GR + (int64_t)a.ur_edge[k] * d + off);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    x.v[e] = fmaf(g.v[e], -a.lr / sd, x.v[e]);
                    gsum.v[e] += g.v[e];
                }
            }
            if (!a.emit_rel) st<V>(row + off, x);
            if (a.gr) st<V>(a.gr + u * (int64_t)a.ld_r + off, gsum);
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
}

// single-pass register-resident variant: body in kge_update_body.hpp
template <int NIT, bool SHARDED, int LEAN>
__global__ __launch_bounds__(KGE_BLOCK) void update_kernel_reg(UpdateArgs a, int nb_ent, int nbM, SmpTail st) {
    if ((int)blockIdx.x >= nbM) { sampler_tail_p3(st, (int)blockIdx.x - nbM); return; }   // (round 5) phase 3 of the sampler tail
    KGE_TL(((int)blockIdx.x < nbM - nb_ent) ? 7 : 4);      // timeline: relation workgroups come first (id 7)
    update_reg_body<NIT, SHARDED, LEAN>(a, nb_ent, (int)blockIdx.x, nbM);
}

int launch_update(const UpdateArgs &a, hipStream_t s, const SmpTail *tail) {
    const int nbE = blocks_for_waves(a.UE), nbR = blocks_for_waves(a.UR);
    if (nbE + nbR == 0) return tail && tail->phase ? KGE_ERR_ARG : KGE_OK;
    SmpTail st{};
    if (tail && tail->phase && tail->advance > 0) { st = *tail; st.phase = 3; }       // (only the group's last job finishes under its own update launch)
    const int nbM = nbE + nbR;
    const dim3 g(nbM + (st.phase ? ST_P3_WGS : 0)), b(KGE_BLOCK);
    const int dmax = a.model_d_e > a.d_r ? a.model_d_e : a.d_r;
    const bool vec = a.model_d_e % 4 == 0 && a.d_r % 4 == 0;
    const bool sharded = a.em.n != 0 || a.rm.n != 0;
    const bool inplace = !a.emit_ent && !a.emit_rel && !a.g0 && !a.g1 && !a.gs0 && !a.gs1 && !a.gr && !a.gsr && !a.rid &&
                         !a.dry;
    if ((a.Hs || a.Rs || a.Ns) && !(vec && dmax <= 1024)) return KGE_ERR_ARG;      // stale-row regulariser: register-resident kernel only
    if (a.Q && !a.transe_fast) return KGE_ERR_ARG;
    // (the LEAN instances fix the regulariser's norm at 3: plain multiplies instead of exp2 / log2 per element)
    const bool reg3 = !(a.reg_coef > 0.f && a.reg_norm > 0) || a.reg_norm == 3;
    // (4: the generic instance - gradient-emitting step, neg_deg_sample - with the norm fixed at 3 as well)
    int lean = !reg3 ? 0 : ((!inplace || a.nd_chunk) ? 4 : (a.transe_fast ? (a.Q ? 3 : 1) : 2));
    const int nit = dmax <= 256 ? 1 : (dmax <= 512 ? 2 : 4);
#ifndef UPD_NO_EMIT6
    if (lean == 4 && a.emit_ent && a.emit_rel && a.g0 && (a.g1 || a.msg_rows) && a.gr && !a.transe_fast && !a.nd_chunk && !a.Hs && !a.Ts && !a.Rs && !a.Ns && !a.dry)
        lean = 6;                    // the all-to-all engine's step for the models with per-edge gradient rows
    else if (lean == 4 && a.emit_ent && !a.emit_rel && a.g0 && (a.g1 || a.msg_rows) && !a.gr && !a.gsr && !a.rid && !a.transe_fast && !a.nd_chunk && !a.Hs &&
             !a.Ts && !a.Rs && !a.Ns && !a.dry)
        lean = 7;                    // ... under relation partitioning: entity messages out, the relation trace applied in place
#endif
    if (a.gn_parts > 0) {            // unsummed GN partials / GA parts (see UpdateArgs): the folded instance or nothing
        if (lean != 1 || sharded || !(vec && dmax <= 512) || !a.GNp || a.gn_parts > KGE_GN_MAXP || a.ga_parts < 1 || a.ga_parts > KGE_GA_MAXP ||
            a.Hs || a.Rs || a.Ns)
            return KGE_ERR_ARG;
        lean = 5;
    }
#define KGE_UPD(N, SH, LE) hipLaunchKernelGGL((update_kernel_reg<N, SH, LE>), g, b, 0, s, a, nbE, nbM, st)
#define KGE_UPD_L(N, SH)                                                         \
    do { if (lean == 1) KGE_UPD(N, SH, 1); else if (lean == 2) KGE_UPD(N, SH, 2); else if (lean == 3) KGE_UPD(N, SH, 3); \
         else if (lean == 4) KGE_UPD(N, SH, 4); else if (lean == 6) KGE_UPD(N, SH, 6); else if (lean == 7) KGE_UPD(N, SH, 7); \
         else KGE_UPD(N, SH, 0); } while (0)
#define KGE_UPD_N(N) do { if (sharded) KGE_UPD_L(N, true); else KGE_UPD_L(N, false); } while (0)
    if (lean == 5) { if (nit == 1) KGE_UPD(1, false, 5); else KGE_UPD(2, false, 5); }
    else
    if (vec && dmax <= 1024) {
        if (nit == 1) KGE_UPD_N(1); else if (nit == 2) KGE_UPD_N(2); else KGE_UPD_N(4);
    }
#undef KGE_UPD_N
#undef KGE_UPD_L
#undef KGE_UPD
    else {
        if (st.phase) return KGE_ERR_ARG;                    // (the generic kernels carry no sampler tail)
        if (vec) hipLaunchKernelGGL(update_kernel<4>, g, b, 0, s, a, nbE);
        else hipLaunchKernelGGL(update_kernel<1>, g, b, 0, s, a, nbE);
    }
    return check_launch();
}

// ------------------------------------------------------------------------------------------
// stand-alone ExternalEmbedding.update for one trace with arbitrary duplicate indices and no
// plan: lock-free float atomics, two phases (state accumulate, then apply).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(KGE_BLOCK) void adagrad_state_kernel(float *state, int dim,
                                                                  const int64_t *idx,
                                                                  const float *grad, int64_t n) {
    const int64_t k = WAVE_ID();
    if (k >= n) return;
    const int lane = LANE();
    const float *g = grad + k * (int64_t)dim;
    float ss = 0.f;
    for (int d = lane; d < dim; d += 64) ss += g[d] * g[d];
    ss = wave_sum(ss) / (float)dim;
    if (lane == 0) atomicAdd(&state[idx[k]], ss);
}
__global__ __launch_bounds__(KGE_BLOCK) void adagrad_apply_kernel(float *table, const float *state,
                                                                  int dim, const int64_t *idx,
                                                                  const float *grad, int64_t n,
                                                                  float lr, float eps) {
    const int64_t k = WAVE_ID();
    if (k >= n) return;
    const int lane = LANE();
    const int64_t id = idx[k];
    const float sd = sqrtf(state[id]) + eps;
    const float *g = grad + k * (int64_t)dim;
    float *row = table + id * (int64_t)dim;
    for (int d = lane; d < dim; d += 64) atomicAdd(&row[d], g[d] * (-lr / sd));
}

int launch_adagrad_scatter(float *table, float *state, int dim, const int64_t *idx,
                           const float *grad, int64_t n, float lr, float eps, hipStream_t s) {
    if (n == 0) return KGE_OK;
    const int nb = blocks_for_waves(n);
    hipLaunchKernelGGL(adagrad_state_kernel, dim3(nb), dim3(KGE_BLOCK), 0, s, state, dim, idx, grad, n);
    hipLaunchKernelGGL(adagrad_apply_kernel, dim3(nb), dim3(KGE_BLOCK), 0, s, table, state, dim, idx,
                       grad, n, lr, eps);
    return check_launch();
}

// owner-side apply of pushed (row gradient, Adagrad increment) pairs; unique idx per call.
template <int V>
__global__ __launch_bounds__(KGE_BLOCK) void apply_rows_kernel(float *table, float *state, int dim,
                                                               const int64_t *idx, const float *g,
                                                               const float *gs, int64_t n, float lr,
                                                               float eps) {
    const int64_t k = WAVE_ID();
    if (k >= n) return;
    const int lane = LANE();
    const int64_t id = idx[k];
    const float inc = gs[k];
    if (id < 0 || inc == 0.f) return;
    const float sN = state[id] + inc;
    const float sd = sqrtf(sN) + eps;
    float *row = table + id * (int64_t)dim;
    const float *gg = g + k * (int64_t)dim;
    for (int it = lane; it < dim / V; it += 64) {
        Pack<V> x = ld<V>(row + it * V);
        const Pack<V> gv = ld<V>(gg + it * V);
#pragma unroll
        for (int e = 0; e < V; ++e) x.v[e] = fmaf(gv.v[e], -lr / sd, x.v[e]);
        st<V>(row + it * V, x);
    }
    if (lane == 0) state[id] = sN;
}

// packed messages: T traces per row applied in order by one wavefront (see include/kge_hip.h)
template <int V>
__global__ __launch_bounds__(KGE_BLOCK) void apply_packed_kernel(float *table, float *state, int dim,
                                                                 const int64_t *idx, const float *msg, int mld,
                                                                 int64_t n, int T, float lr, float eps) {
    const int64_t k = WAVE_ID();
    if (k >= n) return;
    const int lane = LANE();
    const float *m = msg + k * (int64_t)mld;
    int64_t id;
    if (idx) id = idx[k];
    else {
        const int32_t *w = reinterpret_cast<const int32_t *>(m + (int64_t)T * dim + T);
        id = (int64_t)(uint32_t)w[0] | ((int64_t)w[1] << 32);
    }
    if (id < 0) return;
    float *row = table + id * (int64_t)dim;
    float s = state[id];
    bool any = false;
    for (int t = 0; t < T; ++t) {
        const float inc = m[(int64_t)T * dim + t];
        if (inc == 0.f) continue;
        any = true;
        s += inc;
        const float sd = sqrtf(s) + eps;
        const float *g = m + (int64_t)t * dim;
        for (int it = lane; it < dim / V; it += 64) {
            Pack<V> x = ld<V>(row + it * V);
            const Pack<V> gv = ld<V>(g + it * V);
#pragma unroll
            for (int e = 0; e < V; ++e) x.v[e] = fmaf(gv.v[e], -lr / sd, x.v[e]);
            st<V>(row + it * V, x);
        }
    }
    if (any && lane == 0) state[id] = s;
}

int launch_adagrad_apply_packed(float *table, float *state, int dim, const int64_t *idx, const float *msg,
                                int ld, int64_t n, int ntraces, float lr, float eps, hipStream_t s) {
    if (n == 0) return KGE_OK;
    const int nb = blocks_for_waves(n);
    if (dim % 4 == 0 && ld % 4 == 0)
        hipLaunchKernelGGL(apply_packed_kernel<4>, dim3(nb), dim3(KGE_BLOCK), 0, s, table, state, dim, idx, msg, ld, n, ntraces, lr, eps);
    else
        hipLaunchKernelGGL(apply_packed_kernel<1>, dim3(nb), dim3(KGE_BLOCK), 0, s, table, state, dim, idx, msg, ld, n, ntraces, lr, eps);
    return check_launch();
}

int launch_adagrad_apply_rows(float *table, float *state, int dim, const int64_t *idx,
                              const float *g, const float *gs, int64_t n, float lr, float eps,
                              hipStream_t s) {
    if (n == 0) return KGE_OK;
    const int nb = blocks_for_waves(n);
    if (dim % 4 == 0)
        hipLaunchKernelGGL(apply_rows_kernel<4>, dim3(nb), dim3(KGE_BLOCK), 0, s, table, state, dim, idx, g, gs, n, lr, eps);
    else
        hipLaunchKernelGGL(apply_rows_kernel<1>, dim3(nb), dim3(KGE_BLOCK), 0, s, table, state, dim, idx, g, gs, n, lr, eps);
    return check_launch();
}

// ------------------------------------------------------------------------------------------
// small ops of the drop-in (per-op autograd) route that used to be torch glue
// ------------------------------------------------------------------------------------------
// out[idx[k], :] += src[k, :]  - the backward of gathering rows through LOCAL ids (pos_g.ndata['emb'][head_ids],
// general_models.py:384-388, 410-414; autograd of advanced indexing = index_add): float atomics like the reference's.
__global__ __launch_bounds__(KGE_BLOCK) void scatter_add_rows_kernel(float *out, int dim, const int64_t *idx, const float *src,
                                                                     int64_t n) {
    const int64_t k = WAVE_ID();
    if (k >= n) return;
    float *o = out + idx[k] * (int64_t)dim;
    const float *g = src + k * (int64_t)dim;
    for (int d = LANE(); d < dim; d += 64) atomicAdd(o + d, g[d]);
}
int launch_scatter_add_rows(float *out, int dim, const int64_t *idx, const float *src, int64_t n, hipStream_t s) {
    if (n == 0) return KGE_OK;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(blocks_for_waves(n)), dim3(KGE_BLOCK), 0, s, out, dim, idx, src, n);
    return check_launch();
}

// x.norm(p) ** p over a [n, dim] block (tensor_models.py:54 `norm`; general_models.py:572-576): per-row partials, then a
// fixed-order single-block sum (deterministic); and its gradient gout * p * |x|^(p-1) * sign(x)
__global__ __launch_bounds__(KGE_BLOCK) void pnorm_rows_kernel(const float *x, int64_t n, int dim, int p, float *part) {
    const int64_t k = WAVE_ID();
    if (k >= n) return;
    const float *r = x + k * (int64_t)dim;
    float v = 0.f;
    for (int d = LANE(); d < dim; d += 64) v += reg_val(r[d], p);
    v = wave_sum(v);
    if (LANE() == 0) part[k] = v;
}
__global__ __launch_bounds__(KGE_BLOCK) void pnorm_final_kernel(const float *part, int64_t n, float *out) {
    __shared__ float sh[KGE_WAVES_PER_BLOCK];
    float v = 0.f;
    for (int64_t k = threadIdx.x; k < n; k += KGE_BLOCK) v += part[k];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) *out = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void pnorm_bwd_kernel(const float *x, int64_t total, int p, const float *gout, float *gx) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < total) gx[k] = reg_grad(x[k], *gout, p);
}
int launch_pnorm(const float *x, int64_t n, int dim, int p, float *part, float *out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(pnorm_rows_kernel, dim3(blocks_for_waves(n)), dim3(KGE_BLOCK), 0, s, x, n, dim, p, part);
    hipLaunchKernelGGL(pnorm_final_kernel, dim3(1), dim3(KGE_BLOCK), 0, s, part, n, out);
    return check_launch();
}
int launch_pnorm_bwd(const float *x, int64_t total, int p, const float *gout, float *gx, hipStream_t s) {
    if (total == 0) return KGE_OK;
    hipLaunchKernelGGL(pnorm_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, total, p, gout, gx);
    return check_launch();
}

// x[c, i, i] = 0 for i < chunk on a [C, chunk, Np] block, in place: the diagonal mask of --neg_deg_sample
// (general_models.py:401-402, 429-432: mask[:, 0::(neg_sample_size + 1)] = 0)
__global__ void mask_diag_kernel(float *x, int C, int chunk, int Np) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= C * chunk) return;
    const int i = k % chunk;
    if (i < Np) x[(int64_t)k * Np + i] = 0.f;
}
int launch_mask_diag(float *x, int C, int chunk, int Np, hipStream_t s) {
    if (C * chunk == 0) return KGE_OK;
    hipLaunchKernelGGL(mask_diag_kernel, dim3((C * chunk + 255) / 256), dim3(256), 0, s, x, C, chunk, Np);
    return check_launch();
}

// ranks[i] = 1 + #{j : neg[i, j] >= pos[i] and bias[i, j] != -1}   (KEModel.forward_test, general_models.py:463-478)
__global__ __launch_bounds__(KGE_BLOCK) void rank_mask_kernel(const float *neg, const float *pos, const float *bias, int64_t E,
                                                              int64_t N, int64_t *ranks) {
    const int64_t i = WAVE_ID();
    if (i >= E) return;
    const float p = pos[i];
    const float *r = neg + i * N;
    const float *b = bias ? bias + i * N : nullptr;
    float c = 0.f;                                   // < 2^24 candidates per row: exact in fp32
    for (int64_t j = LANE(); j < N; j += 64) c += (r[j] >= p && (!b || b[j] != -1.f)) ? 1.f : 0.f;
    c = wave_sum(c);
    if (LANE() == 0) ranks[i] = 1 + (int64_t)c;
}
int launch_rank_mask(const float *neg, const float *pos, const float *bias, int64_t E, int64_t N, int64_t *ranks, hipStream_t s) {
    if (E == 0) return KGE_OK;
    hipLaunchKernelGGL(rank_mask_kernel, dim3(blocks_for_waves(E)), dim3(KGE_BLOCK), 0, s, neg, pos, bias, E, N, ranks);
    return check_launch();
}

