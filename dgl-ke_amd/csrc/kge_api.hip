// kge_api.hip - the C ABI of libkge_hip.so (include/kge_hip.h): argument checking, workspace
// carving and the kernel sequence of each entry point.  No allocation, no synchronisation: every
// function only enqueues kernels on the caller's stream.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <algorithm>
#include "kge_common.hpp"
#include "kge_sampler_common.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define KGE_TRY(expr)                                                                  \
    do {                                                                               \
        const int rc_ = (expr);                                                        \
        if (rc_ != KGE_OK) return fail(rc_, "%s failed (%d) at %s:%d", #expr, rc_, __FILE__, __LINE__); \
    } while (0)

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Carver {
    char *base; size_t size, off;
    Carver(void *p, size_t n) : base((char *)p), size(n), off(0) {}
    float *f(size_t n_floats) {
        const size_t o = off;
        off = align_up(off + n_floats * sizeof(float));
        return (float *)(base + o);
    }
    bool ok() const { return off <= size; }
};

int check_model(int model, int d_e, int d_r) {
    if (model < KGE_TRANSE_L1 || model > KGE_TRANSR) return fail(KGE_ERR_ARG, "unknown model %d", model);
    if (model == KGE_TRANSR) {
        if (d_e <= 0 || d_r <= 0 || d_r > 1024) return fail(KGE_ERR_ARG, "TransR needs 0 < d_r <= 1024 (got d_e=%d d_r=%d)", d_e, d_r);
        return KGE_OK;
    }
    if (model == KGE_RESCAL) {
        if (d_e <= 0 || d_e > 1024 || (int64_t)d_r != (int64_t)d_e * d_e)
            return fail(KGE_ERR_ARG, "RESCAL needs d_r == d_e*d_e and d_e <= 1024 (got d_e=%d d_r=%d)", d_e, d_r);
        return KGE_OK;
    }
    if (d_e <= 0 || d_r <= 0) return fail(KGE_ERR_ARG, "bad dims d_e=%d d_r=%d", d_e, d_r);
    if (model == KGE_COMPLEX || model == KGE_SIMPLE) {
        if (d_e % 2 || d_r != d_e) return fail(KGE_ERR_ARG, "%s needs even d_e and d_r == d_e (got %d, %d)",
                                               model == KGE_COMPLEX ? "ComplEx" : "SimplE", d_e, d_r);
    } else if (model == KGE_ROTATE) {
        if (d_e % 2 || d_r != d_e / 2) return fail(KGE_ERR_ARG, "RotatE needs d_r == d_e/2 (got d_e=%d d_r=%d)", d_e, d_r);
    } else if (d_r != d_e) {
        return fail(KGE_ERR_ARG, "%s needs d_r == d_e (got %d, %d)", "TransE/DistMult", d_e, d_r);
    }
    return KGE_OK;
}

inline float rot_div_of(float emb_init) { return (float)((double)emb_init / M_PI); }
inline float clamp_of(int model) { return model == KGE_SIMPLE ? KGE_SIMPLE_CLAMP : 0.f; }

bool use_mfma(int model, int d_e, int N, unsigned flags) {
    return !(flags & KGE_FLAG_FORCE_PAIRWISE) && neg_mfma_supported(model, d_e, N);
}

// SimplE, modular backward: no gradient through a saturated clamp
__global__ void clamp_mask_kernel(const float *dneg, const float *score, float c, float *W, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    W[k] = fabsf(score[k]) >= c ? 0.f : dneg[k];
}

__global__ void l2_scale_kernel(const float *dneg, const float *score, float gamma, float *W, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float d = gamma - score[k];
    W[k] = d > 1e-15f ? dneg[k] / d : 0.f;
}

}  // namespace

int kge_fail(int code, const char *msg) { return fail(code, "%s", msg); }

extern "C" {

int kge_abi_version(void) { return KGE_ABI_VERSION; }
const char *kge_last_error(void) { return g_err; }

int kge_gather_rows(const float *table, int64_t n_rows, int dim, const int64_t *idx, int64_t n_idx,
                    float *out, void *stream) {
    if (!table || (!out && n_idx) || (!idx && n_idx) || dim <= 0 || n_rows < 0 || n_idx < 0)
        return fail(KGE_ERR_ARG, "kge_gather_rows: bad argument");
    KGE_TRY(launch_gather_rows(table, dim, idx, n_idx, out, (hipStream_t)stream));
    return KGE_OK;
}

int kge_score_pos(int model, const float *h, const float *r, const float *t, int64_t B, int d_e,
                  int d_r, float gamma, float emb_init, float *out, void *stream) {
    if (int rc = check_model(model, d_e, d_r)) return rc;
    if (!h || !r || !t || !out || B < 0) return fail(KGE_ERR_ARG, "kge_score_pos: bad argument");
    if (model == KGE_TRANSR) return fail(KGE_ERR_ARG, "TransR has no modular score ops: use kge_step_fused / kge_rank_eval_ex");
    if (model == KGE_RESCAL) {                       // p = h . (M t), one pass over M per edge
        RescalMatvecArgs m{};
        m.B = (int)B; m.D = d_e; m.rel = r; m.y1 = t; m.pd = h; m.p = out;
        KGE_TRY(launch_rescal_matvec(m, (hipStream_t)stream));
        return KGE_OK;
    }
    EdgeFwdArgs a{};
    a.src = EdgeSrc{h, nullptr, t, nullptr, r, nullptr};
    a.B = (int)B; a.d_e = d_e; a.d_r = d_r; a.neg_head = 0; a.model = model;
    a.gamma = gamma; a.rot_div = rot_div_of(emb_init);
    a.pos_score = out;
    KGE_TRY(launch_edge_fwd(a, (hipStream_t)stream));
    return KGE_OK;
}

int kge_score_pos_bwd(int model, const float *h, const float *r, const float *t, const float *dpos,
                      int64_t B, int d_e, int d_r, float gamma, float emb_init, float *gh, float *gr,
                      float *gt, void *stream) {
    if (int rc = check_model(model, d_e, d_r)) return rc;
    if (!h || !r || !t || !dpos || B < 0) return fail(KGE_ERR_ARG, "kge_score_pos_bwd: bad argument");
    if (model == KGE_TRANSR) return fail(KGE_ERR_ARG, "TransR has no modular score ops: use kge_step_fused / kge_rank_eval_ex");
    if (model == KGE_RESCAL) {                       // gh = dp M t, gt = dp M^T h, gr = dp h t^T
        if (!gh || !gt || !gr) return fail(KGE_ERR_ARG, "kge_score_pos_bwd: RESCAL needs all three outputs");
        hipStream_t s = (hipStream_t)stream;
        RescalMatvecArgs m{};
        m.B = (int)B; m.D = d_e; m.rel = r; m.y1 = t; m.r1 = gh; m.z1 = h; m.c1 = gt;
        KGE_TRY(launch_rescal_matvec(m, s));
        KGE_TRY(launch_rescal_axpy(dpos, gh, nullptr, (int)B, d_e, gh, s));
        KGE_TRY(launch_rescal_axpy(dpos, gt, nullptr, (int)B, d_e, gt, s));
        RescalOuterArgs o{};
        o.B = (int)B; o.D = d_e; o.c = dpos; o.u = h; o.v = t; o.G = gr;
        KGE_TRY(launch_rescal_outer(o, s));
        return KGE_OK;
    }
    EdgeBwdArgs a{};
    a.src = EdgeSrc{h, nullptr, t, nullptr, r, nullptr};
    a.B = (int)B; a.d_e = d_e; a.d_r = d_r; a.neg_head = 0; a.model = model;
    a.gamma = gamma; a.rot_div = rot_div_of(emb_init);
    a.dpos = dpos; a.GA = nullptr; a.reg_coef = 0.f; a.reg_norm = 0; a.clamp_pos = 1;
    a.GH = gh; a.GT = gt; a.GR = gr;
    KGE_TRY(launch_edge_bwd(a, (hipStream_t)stream));
    return KGE_OK;
}

size_t kge_score_neg_workspace_bytes(int model, int C, int chunk, int N, int d_e) {
    const size_t B = (size_t)C * chunk, CN = (size_t)C * N;
    size_t n = 0;
    n += align_up(B * d_e * sizeof(float));      // A
    n += align_up(B * sizeof(float));            // (reserved)
    n += align_up(CN * sizeof(float));           // (reserved)
    n += align_up(B * (size_t)N * sizeof(float)); // W (L2-scaled dneg)
    n += align_up(B * d_e * sizeof(float));      // GA
    if (neg_bwd_lc_supported(model, d_e)) n += align_up(neg_bwd_lc_partial_floats(model, C, chunk, N, d_e) * sizeof(float));
    return n;
}

// pos-side vectors A = T(pos_side, rel) for the modular negative-score entry points
static int neg_prepare(int model, int neg_head, const float *pos_side, const float *rel,
                       const float *neg, int C, int chunk, int N, int d_e, int d_r, float gamma,
                       float emb_init, bool l2g, Carver &cv, float *&A, float *&asq, float *&bsq,
                       hipStream_t s) {
    const int B = C * chunk;
    A = cv.f((size_t)B * d_e);
    asq = cv.f(B); bsq = cv.f((size_t)C * N);
    if (!cv.ok()) return fail(KGE_ERR_WORKSPACE, "workspace too small");
    if (model == KGE_RESCAL) {
        RescalMatvecArgs m{};
        m.B = B; m.D = d_e; m.rel = rel; m.y1 = pos_side; m.r1 = A;
        KGE_TRY(launch_rescal_matvec(m, s));
        return KGE_OK;
    }
    EdgeFwdArgs a{};
    a.src = EdgeSrc{pos_side, nullptr, pos_side, nullptr, rel, nullptr};
    a.B = B; a.d_e = d_e; a.d_r = d_r; a.neg_head = neg_head; a.model = model;
    a.gamma = gamma; a.rot_div = rot_div_of(emb_init);
    a.A = A;
    if (l2g) { a.asq = asq; a.nbase = neg; a.nidx = nullptr; a.n_neg = C * N; a.bsq = bsq; }
    KGE_TRY(launch_edge_fwd(a, s));
    return KGE_OK;
}

static void fill_gemm(GemmArgs &g, int model, int C, int chunk, int N, int d_e, float gamma,
                      const float *A, const float *nbase, const int64_t *nidx) {
    g = GemmArgs{};
    g.model = model; g.C = C; g.chunk = chunk; g.N = N; g.D = d_e; g.gamma = gamma;
    g.A = A; g.nbase = nbase; g.nidx = nidx; g.B = C * chunk; g.clampv = clamp_of(model);
}
static void fill_pair(NegArgs &na, int model, int C, int chunk, int N, int d_e, float gamma,
                      const float *A, const float *nbase, const int64_t *nidx) {
    na = NegArgs{};
    na.model = model; na.C = C; na.chunk = chunk; na.N = N; na.d_e = d_e; na.gamma = gamma;
    na.A = A; na.nbase = nbase; na.nidx = nidx; na.clampv = clamp_of(model);
}

int kge_score_neg_fwd(int model, int neg_head, const float *pos_side, const float *rel,
                      const float *neg, int C, int chunk, int N, int d_e, int d_r, float gamma,
                      float emb_init, float *out, void *ws, size_t ws_bytes, unsigned flags,
                      void *stream) {
    if (int rc = check_model(model, d_e, d_r)) return rc;
    if (!pos_side || !rel || !neg || !out || !ws || C < 0 || chunk <= 0 || N <= 0)
        return fail(KGE_ERR_ARG, "kge_score_neg_fwd: bad argument");
    if (model == KGE_TRANSR) return fail(KGE_ERR_ARG, "TransR has no modular score ops: use kge_step_fused / kge_rank_eval_ex");
    if (C == 0) return KGE_OK;
    hipStream_t s = (hipStream_t)stream;
    Carver cv(ws, ws_bytes);
    float *A, *asq, *bsq;
    const bool mf = use_mfma(model, d_e, N, flags);
    if (int rc = neg_prepare(model, neg_head, pos_side, rel, neg, C, chunk, N, d_e, d_r, gamma,
                             emb_init, mf && model == KGE_TRANSE_L2, cv, A, asq, bsq, s)) return rc;
    if (mf) {
        GemmArgs g; fill_gemm(g, model, C, chunk, N, d_e, gamma, A, neg, nullptr);
        g.S = out; g.asq = asq; g.bsq = bsq;
        KGE_TRY(launch_neg_fwd_gemm(g, s));
    } else {
        NegArgs na; fill_pair(na, model, C, chunk, N, d_e, gamma, A, neg, nullptr);
        na.S = out;
        KGE_TRY(launch_neg_fwd_pair(na, s));
    }
    return KGE_OK;
}

int kge_score_neg_bwd(int model, int neg_head, const float *pos_side, const float *rel,
                      const float *neg, const float *neg_score, const float *dneg, int C,
                      int chunk, int N, int d_e, int d_r, float gamma, float emb_init,
                      float *g_pos_side, float *g_rel, float *g_neg, void *ws, size_t ws_bytes,
                      unsigned flags, void *stream) {
    if (int rc = check_model(model, d_e, d_r)) return rc;
    if (!pos_side || !rel || !neg || !dneg || !g_pos_side || !g_rel || !g_neg || !ws || C < 0 ||
        chunk <= 0 || N <= 0)
        return fail(KGE_ERR_ARG, "kge_score_neg_bwd: bad argument");
    if (model == KGE_TRANSR) return fail(KGE_ERR_ARG, "TransR has no modular score ops: use kge_step_fused / kge_rank_eval_ex");
    if ((model == KGE_TRANSE_L2 || model == KGE_SIMPLE) && !neg_score)
        return fail(KGE_ERR_ARG, "kge_score_neg_bwd: TransE_l2 / SimplE need the forward scores");
    if (C == 0) return KGE_OK;
    hipStream_t s = (hipStream_t)stream;
    const int B = C * chunk;
    Carver cv(ws, ws_bytes);
    float *A, *asq, *bsq;
    if (int rc = neg_prepare(model, neg_head, pos_side, rel, neg, C, chunk, N, d_e, d_r, gamma,
                             emb_init, false, cv, A, asq, bsq, s)) return rc;
    float *W = cv.f((size_t)B * N), *GA = cv.f((size_t)B * d_e);
    float *GNp = (neg_bwd_lc_supported(model, d_e) && !(flags & KGE_FLAG_TWO_PASS_PAIR))
                     ? cv.f(neg_bwd_lc_partial_floats(model, C, chunk, N, d_e)) : nullptr;
    if (!cv.ok()) return fail(KGE_ERR_WORKSPACE, "workspace too small: need %zu bytes",
                              kge_score_neg_workspace_bytes(model, C, chunk, N, d_e));
    const float *Wuse = dneg;
    if (model == KGE_TRANSE_L2) {
        const int64_t n = (int64_t)B * N;
        hipLaunchKernelGGL(l2_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dneg,
                           neg_score, gamma, W, n);
        Wuse = W;
    } else if (model == KGE_SIMPLE) {
        const int64_t n = (int64_t)B * N;
        hipLaunchKernelGGL(clamp_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dneg,
                           neg_score, KGE_SIMPLE_CLAMP, W, n);
        Wuse = W;
    }
    if (use_mfma(model, d_e, N, flags)) {
        GemmArgs g; fill_gemm(g, model, C, chunk, N, d_e, gamma, A, neg, nullptr);
        g.W = Wuse; g.GA = GA; g.GN = g_neg;
        KGE_TRY(launch_neg_bwd_gemm(g, s));
    } else {
        NegArgs na; fill_pair(na, model, C, chunk, N, d_e, gamma, A, neg, nullptr);
        na.W = Wuse; na.GA = GA; na.GN = g_neg; na.GNp = GNp;
        KGE_TRY(launch_neg_bwd_pair(na, s));
    }
    if (model == KGE_RESCAL) {                       // a = M x:  dL/dx = M^T GA,  dL/dM = GA x^T
        RescalMatvecArgs m{};
        m.B = B; m.D = d_e; m.rel = rel; m.z1 = GA; m.c1 = g_pos_side;
        KGE_TRY(launch_rescal_matvec(m, s));
        RescalOuterArgs o{};
        o.B = B; o.D = d_e; o.u = GA; o.v = pos_side; o.G = g_rel;
        KGE_TRY(launch_rescal_outer(o, s));
        return KGE_OK;
    }
    EdgeBwdArgs e{};
    e.src = EdgeSrc{pos_side, nullptr, pos_side, nullptr, rel, nullptr};
    e.B = B; e.d_e = d_e; e.d_r = d_r; e.neg_head = neg_head; e.model = model;
    e.gamma = gamma; e.rot_div = rot_div_of(emb_init);
    e.dpos = nullptr; e.GA = GA; e.reg_coef = 0.f; e.reg_norm = 0;
    e.GH = neg_head ? nullptr : g_pos_side;
    e.GT = neg_head ? g_pos_side : nullptr;
    e.GR = g_rel;
    KGE_TRY(launch_edge_bwd(e, s));
    return KGE_OK;
}

int kge_loss_fwd_bwd(int loss_genre, int adv, float adv_temp, int pairwise, float margin,
                     const float *pos, const float *neg, const float *w, int64_t B, int N,
                     float *loss3, float *dpos, float *dneg, void *ws, size_t ws_bytes,
                     void *stream) {
    if (loss_genre < KGE_LOSS_LOGSIGMOID || loss_genre > KGE_LOSS_BCE)
        return fail(KGE_ERR_ARG, "unknown loss genre %d", loss_genre);
    if (pairwise && adv) return fail(KGE_ERR_ARG, "loss cannot be pairwise and adversarial sampled");
    if (pairwise && loss_genre != KGE_LOSS_LOGISTIC && loss_genre != KGE_LOSS_HINGE)
        return fail(KGE_ERR_ARG, "this loss cannot be applied to pairwise loss function");
    if (!pos || !neg || !dpos || !dneg || !ws || B <= 0 || N <= 0)
        return fail(KGE_ERR_ARG, "kge_loss_fwd_bwd: bad argument");
    Carver cv(ws, ws_bytes);
    float *row_pos = cv.f(B), *row_neg = cv.f(B);
    if (!cv.ok()) return fail(KGE_ERR_WORKSPACE, "workspace too small");
    LossArgs a{};
    a.B = (int)B; a.N = N; a.genre = loss_genre; a.adv = adv; a.pairwise = pairwise;
    a.adv_temp = adv_temp; a.margin = margin; a.pos = pos; a.neg = neg; a.w = w;
    a.dpos = dpos; a.dneg = dneg; a.row_pos = row_pos; a.row_neg = row_neg;
    a.l2_scale = 0; a.gamma = 0.f; a.neg_copy = nullptr; a.acc = nullptr;
    KGE_TRY(launch_loss(a, (hipStream_t)stream));
    if (loss3) {
        // finalize writes 4 floats {pos, neg, loss, reg}; loss3 has room for 3 -> stage in ws
        float *l4 = cv.f(4);
        if (!cv.ok()) return fail(KGE_ERR_WORKSPACE, "workspace too small");
        FinalizeArgs f{};
        f.B = (int)B; f.UE = 0; f.UR = 0; f.pairwise = pairwise;
        f.row_pos = row_pos; f.row_neg = row_neg; f.loss4 = l4;
        KGE_TRY(launch_finalize(f, (hipStream_t)stream));
        if (hipMemcpyAsync(loss3, l4, 3 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            return fail(KGE_ERR_LAUNCH, "hipMemcpyAsync failed");
    }
    return KGE_OK;
}

int kge_reduce_loss(float *loss_accum, float *out4, int zero_after, void *stream) {
    if (!loss_accum || !out4) return fail(KGE_ERR_ARG, "kge_reduce_loss: null argument");
    KGE_TRY(launch_reduce_acc(loss_accum, out4, zero_after, (hipStream_t)stream));
    return KGE_OK;
}

int kge_adagrad_scatter(float *table, float *state_sum, int64_t n_rows, int dim, const int64_t *idx,
                        const float *grad, int64_t n_idx, float lr, float eps, void *stream) {
    if (!table || !state_sum || (!idx && n_idx) || (!grad && n_idx) || dim <= 0 || n_rows < 0)
        return fail(KGE_ERR_ARG, "kge_adagrad_scatter: bad argument");
    KGE_TRY(launch_adagrad_scatter(table, state_sum, dim, idx, grad, n_idx, lr, eps, (hipStream_t)stream));
    return KGE_OK;
}

int kge_scatter_add_rows(float *out, int64_t n_rows, int dim, const int64_t *idx, const float *src, int64_t n_idx, void *stream) {
    if ((n_idx && (!out || !idx || !src)) || dim <= 0 || n_rows < 0 || n_idx < 0) return fail(KGE_ERR_ARG, "kge_scatter_add_rows: bad argument");
    KGE_TRY(launch_scatter_add_rows(out, dim, idx, src, n_idx, (hipStream_t)stream));
    return KGE_OK;
}

int kge_pnorm_pow(const float *x, int64_t n, int dim, int p, float *out, void *ws, size_t ws_bytes, void *stream) {
    if (!out || (n && !x) || n < 0 || dim <= 0 || p <= 0 || !ws || ws_bytes < (size_t)(n > 0 ? n : 1) * sizeof(float))
        return fail(KGE_ERR_ARG, "kge_pnorm_pow: bad argument (workspace: n floats)");
    KGE_TRY(launch_pnorm(x, n, dim, p, (float *)ws, out, (hipStream_t)stream));
    return KGE_OK;
}

int kge_pnorm_pow_bwd(const float *x, int64_t n, int dim, int p, const float *gout, float *gx, void *stream) {
    if ((n && (!x || !gx)) || !gout || n < 0 || dim <= 0 || p <= 0) return fail(KGE_ERR_ARG, "kge_pnorm_pow_bwd: bad argument");
    KGE_TRY(launch_pnorm_bwd(x, n * (int64_t)dim, p, gout, gx, (hipStream_t)stream));
    return KGE_OK;
}

int kge_mask_diag(float *x, int C, int chunk, int Np, void *stream) {
    if (!x || C < 0 || chunk < 0 || Np <= 0) return fail(KGE_ERR_ARG, "kge_mask_diag: bad argument");
    KGE_TRY(launch_mask_diag(x, C, chunk, Np, (hipStream_t)stream));
    return KGE_OK;
}

int kge_rank_from_scores(const float *neg, const float *pos, const float *bias, int64_t E, int64_t N, int64_t *ranks, void *stream) {
    if ((E && (!neg || !pos || !ranks)) || E < 0 || N < 0 || N >= (1 << 24)) return fail(KGE_ERR_ARG, "kge_rank_from_scores: bad argument");
    KGE_TRY(launch_rank_mask(neg, pos, bias, E, N, ranks, (hipStream_t)stream));
    return KGE_OK;
}

int kge_adagrad_apply_packed(float *table, float *state_sum, int64_t n_rows, int dim,
                             const int64_t *idx, const float *msg, int ld, int64_t n, int ntraces,
                             float lr, float eps, void *stream) {
    if (!table || !state_sum || (n && !msg) || dim <= 0 || n_rows < 0 || ntraces < 1 ||
        ld < ntraces * dim + ntraces + (idx ? 0 : 2))
        return fail(KGE_ERR_ARG, "kge_adagrad_apply_packed: bad argument");
    KGE_TRY(launch_adagrad_apply_packed(table, state_sum, dim, idx, msg, ld, n, ntraces, lr, eps, (hipStream_t)stream));
    return KGE_OK;
}

int kge_adagrad_apply_rows(float *table, float *state_sum, int64_t n_rows, int dim,
                           const int64_t *idx, const float *g, const float *gs, int64_t n, float lr,
                           float eps, void *stream) {
    if (!table || !state_sum || (n && (!idx || !g || !gs)) || dim <= 0 || n_rows < 0)
        return fail(KGE_ERR_ARG, "kge_adagrad_apply_rows: bad argument");
    KGE_TRY(launch_adagrad_apply_rows(table, state_sum, dim, idx, g, gs, n, lr, eps, (hipStream_t)stream));
    return KGE_OK;
}

// ------------------------------------------------------------------------------------------
// fused step
// ------------------------------------------------------------------------------------------
size_t kge_step_workspace_bytes(const kge_hparams *hp, int B, int C, int chunk, int N, int UE, int UR) {
    const bool nd = (hp->flags & KGE_FLAG_NEG_DEG_SAMPLE) != 0;
    if (nd) N += chunk;                                   // neg_deg_sample: the chunk's own positives join its negatives
    const size_t d_e = hp->d_e, d_r = hp->d_r, CN = (size_t)C * N, tj16 = (N + 15) / 16;
    size_t n = 0;
    auto add = [&](size_t floats) { n += align_up(floats * sizeof(float)); };
    add(B * d_e);        // A
    add(CN * d_e);       // Bn (pairwise-kernel path only)
    add(B); add(CN);     // asq, bsq
    add(B); add(B);      // pos score, dpos
    add((size_t)B * N);  // S / W
    add(B * tj16); add(B * tj16); add(B * tj16);   // per-(row, 16-column tile) partials: max / sum-exp / loss
    add(B * d_e * (size_t)(!(hp->flags & KGE_FLAG_TWO_PASS_PAIR) ? neg_bwd_lc_splits(hp->model, C, chunk, N, hp->d_e) : 1));   // GA (shared-pair backward: in parts)
    add(CN * d_e);       // GN
    add(B * d_e);        // P (TransE) or GH
    add(B * d_e);                 // GT
    if (hp->model == KGE_TRANSR) {   // projections, q, signs, dq, P s, P dq; sign bytes; per-edge projection gradients
        for (int k = 0; k < 5; ++k) add(B * d_r);
        add(B * d_e); add(B * d_e);
        add(((size_t)B * N * d_r + 3) / 4);
        add((size_t)B * d_e * d_r);
        add(B); add(B); add(UR); add(UR);
        add((size_t)TRANSR_GN_GROUPS_WIDE * CN * d_e);
        add((size_t)B * ((d_e + 63) / 64) * ((d_r + 63) / 64));       // sum of squares per 64 x 64 tile of the projection gradients
    }
    if (hp->model == KGE_RESCAL) {   // V = M t, M^T h, M^T GA (no [B, d_r] buffer) + update scratch
        add(B * d_e); add(B * d_e * RESCAL_RBN); add(B * d_e * RESCAL_RBN);   // V; parts of M^T h, M^T GA per row block
        add((size_t)B * RESCAL_RB); add(UR); add((size_t)UR * (RESCAL_RB > RESCAL_RBN ? RESCAL_RB : RESCAL_RBN)); add((size_t)B * RESCAL_RBN);
    }
    else add(B * d_r);            // GR
    add(B); add(B); add(UE); add(UR);           // row_pos, row_neg, reg_ent, reg_rel
    if (neg_bwd_lc_supported(hp->model, hp->d_e))   // TransE_l1 / RotatE: GN partials of the shared-pair backward
        add(neg_bwd_lc_partial_floats(hp->model, C, chunk, N, hp->d_e));
    if (hp->model == KGE_TRANSR || hp->model == KGE_RESCAL) {      // sharded entity tables: dense [h | t | negative] rows + identity ids
        add((size_t)(2 * B + CN) * d_e); add((size_t)2 * (2 * B + CN));
    }
    if (hp->model == KGE_TRANSR && nd) add((size_t)2 * CN);        // neg_deg_sample: the combined [own | sampled] id list
    return n;
}

// phases of one step.  The strict step runs all three back to back on one stream; the --async_update pipeline
// (kge_step_async) runs PREP(s) | SCORE(s) on the caller's stream and UPDATE(s-1) on its side stream in between.
enum { PH_PREP = 1, PH_FWD = 2, PH_BWD = 16, PH_SCORE = 18, PH_UPD_ENT = 4, PH_UPD_REL = 8, PH_UPDATE = 12, PH_ALL = 31,
       PH_STRICT = 32 };   // PH_STRICT: the phases belong to a strict step issued in pieces (kge_step_phase): table reads stay
                           // table reads (no dense copies), exactly the kernels of PH_ALL

static int step_impl(const kge_hparams *hp, const kge_tables *tb, const kge_batch *b,
                     const kge_step_out *out, const kge_emit *emit, void *ws, size_t ws_bytes,
                     void *stream, const kge_shards *sh = nullptr, int phases = PH_ALL,
                     UpdateArgs *build_update = nullptr,      // PH_UPD_*: fill the launch arguments instead of launching
                     const UpdateArgs *co_update = nullptr,   // PH_FWD: another step's update to run alongside the forward
                     EdgeFwdArgs *build_prep = nullptr,       // PH_PREP: fill the launch arguments instead of launching
                     const EdgeFwdArgs *co_prep = nullptr,    // PH_BWD: the NEXT step's PREP to run alongside the backward
                     const SmpTail *tail = nullptr) {         // the sampler job this step's launches carry as tail workgroups
    if (!hp || !tb || !b || !ws) return fail(KGE_ERR_ARG, "kge_step: null argument");
    kge::ShardMap em{}, rm{};
    if (sh) {
        const bool rl = sh->rel_local != nullptr;      // ABI 8: relation-side tables local to this rank (relation partitioning)
        if (sh->n_shards < 1 || sh->ent_rows_per_shard <= 0 || !sh->ent_rows || !sh->ent_state ||
            (!rl && (sh->rel_rows_per_shard <= 0 || !sh->rel_rows || !sh->rel_state)) || (rl && !sh->rel_state_local))
            return fail(KGE_ERR_ARG, "kge_step_sharded: bad shard map");
        em = kge::ShardMap{sh->ent_rows, sh->ent_state, sh->ent_rows_per_shard, sh->n_shards};
        if (!rl) rm = kge::ShardMap{sh->rel_rows, sh->rel_state, sh->rel_rows_per_shard, sh->n_shards};
    }
    if (int rc = check_model(hp->model, hp->d_e, hp->d_r)) return rc;
    // (round 6: the relation-matrix models in the gradient-emitting step when the relation side is applied IN PLACE - emit->gr NULL,
    //  the all-to-all engine under relation partitioning: their kernels run against the row cache like every model's, the relation
    //  matrices / projection rows of the batch belong to this rank and are updated here, only entity messages leave)
    if (hp->model == KGE_RESCAL && ((emit && emit->gr) || (sh && !sh->rel_local)))
        return fail(KGE_ERR_ARG, "RESCAL: the gradient-emitting step needs the relation trace in place (emit.gr NULL: relation partitioning); "
                                 "on sharded tables it needs kge_shards.rel_local (ABI 8)");
    if (b->B <= 0 || b->C <= 0 || b->chunk <= 0 || b->N <= 0 || (int64_t)b->C * b->chunk != b->B)
        return fail(KGE_ERR_ARG, "kge_step: need C*chunk == B (B=%d C=%d chunk=%d)", b->B, b->C, b->chunk);
    if (hp->loss_genre < KGE_LOSS_LOGSIGMOID || hp->loss_genre > KGE_LOSS_BCE)
        return fail(KGE_ERR_ARG, "unknown loss genre %d", hp->loss_genre);
    if (hp->pairwise && hp->adv) return fail(KGE_ERR_ARG, "loss cannot be pairwise and adversarial sampled");
    if (hp->pairwise && hp->loss_genre != KGE_LOSS_LOGISTIC && hp->loss_genre != KGE_LOSS_HINGE)
        return fail(KGE_ERR_ARG, "this loss cannot be applied to pairwise loss function");
    if ((!sh && (!tb->ent || !tb->ent_state || !tb->rel || !tb->rel_state)) || !b->h_gid || !b->t_gid ||
        !b->rel_ids || !b->neg_ids || !b->ue_id || !b->ue_pos_ptr || !b->ue_pos_adj ||
        !b->ue_neg_ptr || !b->ue_neg_slot || !b->ur_id || !b->ur_ptr || !b->ur_edge || !b->ue_rec ||
        !b->ur_rec)
        return fail(KGE_ERR_ARG, "kge_step: null table / batch pointer");
    hipStream_t s = (hipStream_t)stream;
    // neg_deg_sample: the scoring / loss / gradient kernels see N = chunk + (sampled negatives) rows per chunk; the
    // plan of the batch keeps indexing the sampled ones (b->N)
    const bool nd = (hp->flags & KGE_FLAG_NEG_DEG_SAMPLE) != 0;
    // (round 4: also in the gradient-emitting step - the in-batch rows' gradients join the row's positive-trace message g0 through
    //  edge_bwd, the sampled rows' the negative-trace message g1 through the update's slot remap, exactly as in the fused step)
    // (round 6: TransR and RESCAL too - the reference's concat-and-mask sits in front of head_neg_prepare / tail_neg_prepare,
    //  general_models.py:396-402, 417-432: model-agnostic.  Not on sharded tables / in the gradient-emitting step.)
    if (nd && (hp->model == KGE_RESCAL || hp->model == KGE_TRANSR) && (sh || emit))
        return fail(KGE_ERR_ARG, "neg_deg_sample for RESCAL / TransR: single-table step only");
    const int B = b->B, C = b->C, chunk = b->chunk, N = nd ? b->chunk + b->N : b->N, CN = C * N;
    const int d_e = hp->d_e, d_r = hp->d_r, tj16 = (N + 15) / 16;
    const bool reg = hp->reg_coef > 0.f && hp->reg_norm > 0;
    const bool pairwise = hp->pairwise != 0;
    // which kernels run
    const bool gemm = use_mfma(hp->model, d_e, N, hp->flags);       // matrix-core negative scoring
    // KGE_FLAG_FUSED_LOSS (matrix-core path, pointwise criteria): no loss kernel - the forward tiles emit the
    // factorised gradient (kge_neg_gemm.hip).  4 launches per step instead of 5, 1.6 MB less traffic, but measured
    // 41.7 vs 41.1 us per step at cfg-T (profiles/r02_fused_loss_experiment.txt), so it is opt-in.
    // (RESCAL has no edge_fwd: its positive-loss part lives in the loss kernel)
    const bool fused_loss = gemm && !pairwise && !nd && hp->model != KGE_RESCAL && (hp->flags & KGE_FLAG_FUSED_LOSS) &&
                            neg_gemm_fused_loss_supported(chunk, N);
    const bool is_l2 = hp->model == KGE_TRANSE_L2;
    const bool transe = hp->model == KGE_TRANSE_L1 || is_l2;
    const int dmax = d_e > d_r ? d_e : d_r;
    const bool transe_fast = transe && !pairwise && !nd && d_e % 4 == 0 && d_r % 4 == 0 && dmax <= 1024 &&
                             !(hp->flags & KGE_FLAG_NO_TRANSE_FAST);
    // ... behind the matrix-core backward: its GA epilogue also writes Q = GA +/- P (into the GT buffer, unused on this path)
    // and the update reads one gradient row per list entry (GemmArgs::Q)
    const bool qfuse = transe_fast && gemm;

    Carver cv(ws, ws_bytes);
    float *A = cv.f((size_t)B * d_e), *Bn = cv.f((size_t)CN * d_e);
    float *asq = cv.f(B), *bsq = cv.f(CN);
    float *P = cv.f(B), *dP = cv.f(B);
    float *S = cv.f((size_t)B * N);
    float *PM = cv.f((size_t)B * tj16), *PS = cv.f((size_t)B * tj16), *PL = cv.f((size_t)B * tj16);
    // (the shared-pair backward of RotatE / TransE_l1 leaves GA in up to 8 parts, neg_bwd_lc_splits)
    const int ga_parts = (!gemm && !(hp->flags & KGE_FLAG_TWO_PASS_PAIR)) ? neg_bwd_lc_splits(hp->model, C, chunk, N, d_e) : 1;
    float *GA = cv.f((size_t)B * d_e * ga_parts), *GN = cv.f((size_t)CN * d_e);
    const bool rescal = hp->model == KGE_RESCAL;
    const bool rescal_rel = rescal && d_e % 4 == 0;      // RESCAL's passes over M per unique relation (16-byte accesses)
    float *GH = cv.f((size_t)B * d_e), *GT = cv.f((size_t)B * d_e);
    float *RV = rescal ? cv.f((size_t)B * d_e) : nullptr, *RC1 = rescal ? cv.f((size_t)B * d_e * RESCAL_RBN) : nullptr;
    float *RC2 = rescal ? cv.f((size_t)B * d_e * RESCAL_RBN) : nullptr;
    float *Rgs = rescal ? cv.f((size_t)B * RESCAL_RB) : nullptr, *Rstd = rescal ? cv.f(b->UR) : nullptr;
    float *Rreg = rescal ? cv.f((size_t)b->UR * (RESCAL_RB > RESCAL_RBN ? RESCAL_RB : RESCAL_RBN)) : nullptr, *Rpp = rescal ? cv.f((size_t)B * RESCAL_RBN) : nullptr;
    float *GR = rescal ? nullptr : cv.f((size_t)B * d_r);
    const bool transr = hp->model == KGE_TRANSR;
    TransRArgs tr{};
    float *TR1 = nullptr, *TR2 = nullptr;
    // ---- TransR / RESCAL on peer-to-peer sharded ENTITY tables (round 6, ABI 8; the reference trains TransR on 8 GPUs with
    // --rel_part, examples/freebase/multi_gpu.sh:80-89).  Their kernels address `table + id * width`; here the batch's entity rows
    // are first gathered through the shard map into ONE dense block [h rows | t rows | negative rows] and the kernels run on that
    // block with identity index arrays - no kernel of the two families changes.  The relation-side tables (relation rows /
    // matrices, TransR's projection table) are local to the rank (kge_shards.rel_local: relation partitioning); the entity update
    // goes through the shard map by global id like every other model's.
    const bool sh_dense = sh && (transr || rescal);
    kge_tables tbd = *tb; kge_batch bd = *b;
    float *Xd = nullptr; int64_t *iota = nullptr;
    if (transr || rescal) {                       // (always carved: the workspace size does not depend on the table layout)
        Xd = cv.f((size_t)(2 * B + C * b->N) * d_e);
        iota = reinterpret_cast<int64_t *>(cv.f((size_t)2 * (2 * B + C * b->N)));
    }
    if (sh_dense) {
        tbd.ent = Xd; tbd.ent_state = nullptr; tbd.n_ent = 2 * B + C * b->N;
        tbd.rel = sh->rel_local; tbd.rel_state = sh->rel_state_local; tbd.n_rel = sh->n_rel;
        tbd.proj = sh->proj_local; tbd.proj_state = sh->proj_state_local;
        bd.h_gid = iota; bd.t_gid = iota + B; bd.neg_ids = iota + 2 * B;
    }
    // TransR with neg_deg_sample: its kernels address the negatives through ONE id list - the combined [own | sampled] ids per chunk
    int64_t *ndids = (transr && nd) ? reinterpret_cast<int64_t *>(cv.f((size_t)2 * CN)) : nullptr;
    const kge_tables *tbx = sh_dense ? &tbd : tb;  // what the TransR / RESCAL kernels read
    const kge_batch *bx = sh_dense ? &bd : b;
    if (transr) {
        if (emit && emit->gr) return fail(KGE_ERR_ARG, "TransR: the gradient-emitting step needs the relation trace in place (emit.gr NULL: relation partitioning)");
        if (sh && !sh->rel_local) return fail(KGE_ERR_ARG, "TransR on sharded tables needs kge_shards.rel_local / proj_local (ABI 8)");
        if (!tbx->proj || !tbx->proj_state) return fail(KGE_ERR_ARG, "TransR needs kge_tables.proj / proj_state");
        tr.HP = cv.f((size_t)B * d_r); tr.TP = cv.f((size_t)B * d_r); tr.Q = cv.f((size_t)B * d_r);
        tr.SG = cv.f((size_t)B * d_r); tr.DQ = cv.f((size_t)B * d_r);
        TR1 = cv.f((size_t)B * d_e); TR2 = cv.f((size_t)B * d_e);
        tr.Z = reinterpret_cast<signed char *>(cv.f(((size_t)B * N * d_r + 3) / 4));
        tr.GP = cv.f((size_t)B * d_e * d_r);
        tr.gs0 = cv.f(B); tr.gs1 = cv.f(B); tr.k0 = cv.f(b->UR); tr.k1 = cv.f(b->UR);
        tr.nG = transr_gn_groups(d_e, d_r, chunk, N);
        tr.GNp = cv.f((size_t)TRANSR_GN_GROUPS_WIDE * CN * d_e);
        tr.gs1p = cv.f((size_t)B * ((d_e + 63) / 64) * ((d_r + 63) / 64));
    }
    float *row_pos = cv.f(B), *row_neg = cv.f(B), *reg_ent = cv.f(b->UE), *reg_rel = cv.f(b->UR);
    float *GNp = (neg_bwd_lc_supported(hp->model, d_e) && !(hp->flags & KGE_FLAG_TWO_PASS_PAIR))
                     ? cv.f(neg_bwd_lc_partial_floats(hp->model, C, chunk, N, d_e)) : nullptr;
    // --async_update pipeline (phases != PH_ALL): dense copies of the h / t / r rows as PREP gathered them, for the
    // kernels that read them again after the previous step's update has started (edge_bwd)
    const bool pipelined = phases != PH_ALL && !(phases & PH_STRICT);      // the async pipeline (kge_step_async)
    const bool need_cp = pipelined && (!transe_fast || reg || (out && out->g_rel));
    float *Hc = need_cp ? cv.f((size_t)B * d_e) : nullptr, *Tc = need_cp ? cv.f((size_t)B * d_e) : nullptr;
    float *Rc = need_cp ? cv.f((size_t)B * d_r) : nullptr;
    if (!cv.ok())
        return fail(KGE_ERR_WORKSPACE, "kge_step: workspace too small (%zu < %zu)", ws_bytes,
                    kge_step_workspace_bytes(hp, B, C, chunk, b->N, b->UE, b->UR));
    float *Pg = GH;                                   // TransE fast path: P rows reuse the GH buffer
    if (out && out->pos_score) P = out->pos_score;
    if (out && out->g_neg) GN = out->g_neg;
    if (out && out->g_rel && !transe_fast) GR = out->g_rel;
    const bool want4 = out && out->loss4;
    float *acc = out ? out->loss_accum : nullptr;
    const LossParams lp{hp->loss_genre, hp->adv, hp->pairwise, hp->adv_temp, hp->margin};

    const EdgeSrc src{tb->ent, b->h_gid, tb->ent, b->t_gid, tb->rel, b->rel_ids, em, rm};
    // ids of the rows the scoring kernels treat as negatives.  neg_deg_sample: [the chunk's own corrupted-side entities |
    // the sampled ids] per chunk - edge_fwd resolves that on the fly and writes the dense copy Bn every later kernel reads
    const int64_t *nids = b->neg_ids;
    if (transr) {
        tr.B = B; tr.C = C; tr.chunk = chunk; tr.N = N; tr.De = d_e; tr.Dr = d_r; tr.neg_head = b->neg_head;
        tr.UR = b->UR; tr.reg_norm = hp->reg_norm; tr.gamma = hp->gamma; tr.lr = hp->lr; tr.eps = hp->eps;
        tr.reg_coef = reg ? hp->reg_coef : 0.f;
        tr.ent = tbx->ent; tr.h_gid = bx->h_gid; tr.t_gid = bx->t_gid; tr.neg_ids = bx->neg_ids; tr.rel_ids = b->rel_ids;
        tr.rel = tbx->rel; tr.proj = tbx->proj; tr.proj_state = tbx->proj_state;
        tr.P = P; tr.S = S; tr.GN = GN; tr.GR = GR; tr.dpos = dP;
        tr.ur_id = b->ur_id; tr.ur_ptr = b->ur_ptr; tr.ur_edge = b->ur_edge; tr.counts_dev = b->counts_dev;
        if (nd) { tr.neg_ids = ndids; tr.nd_chunk = chunk; }     // (N = chunk + sampled; the update adds the sampled rows' regulariser)
    }
    const float rot_div = rot_div_of(hp->emb_init);
    // --async_update pipeline: everything after PREP must read the rows as PREP gathered them (the previous step's
    // update is changing the tables meanwhile): negatives from the dense copy Bn, h / t / r from dense copies too
    const bool async = pipelined;
    if (async && (sh || emit || transr || rescal))
        return fail(KGE_ERR_ARG, "kge_step_async: not available for RESCAL / TransR and the sharded / gradient-emitting steps");

    const bool dense_neg = !gemm || sh || async || nd || (hp->flags & KGE_FLAG_DENSE_NEG);
    const bool l2g = gemm && is_l2;                   // GEMM form of the L2 distance needs |a|^2, |b|^2
    // rows as the scoring / gradient kernels of THIS step see them: the tables, or PREP's dense copies (async)
    const EdgeSrc src_bwd = need_cp ? EdgeSrc{Hc, nullptr, Tc, nullptr, Rc, nullptr, kge::ShardMap{}, kge::ShardMap{}} : src;

    // round 3: edge-forward and the forward GEMM of the strict step share ONE launch (kge_neg_gemm.hip, neg_fwd_edge_kernel): the
    // tiles build their pos-side fragments from the table rows themselves and emit raw products, the loss kernel applies the
    // TransE_l2 distance transform.  4 launches per TransE_l2 step instead of 5.  Not for the async pipeline (its forward shares
    // a launch with the previous update already), dense-negative modes (sharded / neg_deg_sample) and the fused-loss variant.
    const bool merged_fwd = gemm && !dense_neg && !fused_loss && !pipelined && !pairwise &&
                            (phases & (PH_PREP | PH_FWD)) == (PH_PREP | PH_FWD) && !build_prep && !co_update &&
                            !(hp->flags & KGE_FLAG_SPLIT_FWD) && neg_fwd_gemm_with_edge_supported(hp->model, d_e, d_r) &&
                            !(hp->model == KGE_COMPLEX && (hp->flags & KGE_FLAG_FWD_DIRECT));
    // ... and the pairwise family's TransE_l1: its forward tasks build the uniform rows x +/- r themselves and gather the negative
    // rows through neg_ids; the edge half writes the dense copies (A, Bn) the backward kernels read
    const bool merged_pair = !gemm && !pipelined && !pairwise && !nd && !sh && !rescal && !transr &&
                             (phases & (PH_PREP | PH_FWD)) == (PH_PREP | PH_FWD) && !build_prep && !co_update &&
                             !(hp->flags & KGE_FLAG_SPLIT_FWD) && neg_fwd_bcast_with_edge_supported(hp->model, d_e, d_r);
    // round 4, KGE_FLAG_LOSS_IN_FWD: ... and LossGenerator too - the forward tiles store final scores and the workgroup of a 16-row
    // strip that arrives last runs the strip's loss rows (kge_neg_gemm.hip, neg_fwd_loss_edge_kernel): 3 launches per TransE_l2 /
    // DistMult / ComplEx step.  Needs the caller's ticket words (kge_step_out.tickets); pointwise criteria; |a|^2 and |b|^2 of the
    // TransE_l2 distance are then computed by the tiles, not by the edge half.  Opt-in: slower than the loss launch on MI355X
    // (profiles/r04_loss_fold.txt).
    const bool fold_loss = merged_fwd && out && out->tickets && (hp->flags & KGE_FLAG_LOSS_IN_FWD) && !(hp->flags & KGE_FLAG_FWD_DIRECT) &&
                           neg_fwd_loss_fold_supported(hp->model, C, chunk, N, d_e, d_r);
    // round 5: a batch of the NEXT group is built by tail workgroups of this step's first, backward and update launches
    // (kge_sampler_tail.hpp) - the one-call strict step of the matrix-core family in its 4-launch form
    if (tail && !(merged_fwd && !fold_loss && gemm && phases == PH_ALL && !co_prep && !co_update && !build_update && !build_prep &&
                  !transr && !rescal && hp->d_e % 4 == 0 && hp->d_r % 4 == 0 && (hp->d_e > hp->d_r ? hp->d_e : hp->d_r) <= 1024))
        return fail(KGE_ERR_ARG, "kge_step_fused_sampling: this step's launches cannot carry the sampler (matrix-core models, strict "
                                 "4-launch step on local tables only)");
    EdgeFwdArgs ef{};
    if (sh_dense && (phases & PH_PREP))
        KGE_TRY(launch_gather3_sharded(em, d_e, b->h_gid, b->t_gid, b->neg_ids, B, C * b->N, Xd, iota, s));
    if (phases & PH_PREP) {
    // 1. gather + positive score + pos-side vectors (+ positive-loss part, + P rows for TransE)
    ef.src = src; ef.B = B; ef.d_e = d_e; ef.d_r = d_r; ef.neg_head = b->neg_head; ef.model = hp->model;
    ef.gamma = hp->gamma; ef.rot_div = rot_div;
    ef.pos_score = P; ef.A = A;
    ef.nbase = tb->ent; ef.nidx = nids; ef.n_neg = CN;
    if (nd) { ef.nd_own = b->neg_head ? b->h_gid : b->t_gid; ef.nd_chunk = chunk; ef.nd_Ns = b->N; }
    // sharded tables: the scoring kernels re-read the negative rows many times - they must come from
    // the dense local copy edge_fwd makes, never from the (remote, uncached) table rows
    // merged first launch, KGE_FLAG_DENSE_BWD: its edge half also writes the dense copy of the negative rows and the backward GEMM
    // reads that (no id round).  Measured equal to gathering through neg_ids (profiles/r03_merged_fwd.txt) at +1.6 MB of writes: opt-in
    const bool dense_bwd = merged_fwd && (hp->flags & KGE_FLAG_DENSE_BWD);
    ef.Bn = (dense_neg || dense_bwd) ? Bn : nullptr;  // the pairwise kernels read a dense copy
    ef.Hc = Hc; ef.Tc = Tc; ef.Rc = Rc;
    ef.asq = (l2g && !fold_loss) ? asq : nullptr; ef.bsq = (l2g && !fold_loss) ? bsq : nullptr;
    ef.do_pos_loss = pairwise ? 0 : 1; ef.lp = lp; ef.w = b->edge_w; ef.w_mean = b->edge_w_mean;
    ef.dpos = dP; ef.row_pos = want4 ? row_pos : nullptr; ef.acc = acc;
    ef.P = transe_fast ? Pg : nullptr;
    if (transr) {
        // hp = h P, tp = t P in one pass over every edge's projection matrix; then p, sign(u), q; then the
        // batched projection of the chunk's negatives with the L1 epilogue (scores + sign bytes)
        if (nd) KGE_TRY(launch_nd_ids(b->neg_head ? b->h_gid : b->t_gid, b->neg_ids, C, chunk, b->N, ndids, s));
        RescalMatvecArgs m{};
        m.B = B; m.D = d_e; m.Dc = d_r; m.rel = tbx->proj; m.ridx = b->rel_ids;
        m.z1 = tbx->ent; m.z1idx = bx->h_gid; m.c1 = tr.HP;
        m.z2 = tbx->ent; m.z2idx = bx->t_gid; m.c2 = tr.TP;
        KGE_TRY(launch_rescal_matvec(m, s));
        KGE_TRY(launch_transr_pos(tr, s));
        KGE_TRY(launch_transr_fwd(tr, s));
    } else if (rescal) {
        // V = M t (always: p = h.V), A = M x with x = h in tail mode (then a second product of the same pass)
        if (!rescal_rel) {               // d_e not a multiple of 4: one pass over M per EDGE
            RescalMatvecArgs m{};
            m.B = B; m.D = d_e; m.rel = tbx->rel; m.ridx = b->rel_ids;
            m.y1 = tbx->ent; m.y1idx = bx->t_gid; m.r1 = b->neg_head ? A : RV;
            if (!b->neg_head) { m.y2 = tbx->ent; m.y2idx = bx->h_gid; m.r2 = A; }
            m.pd = tbx->ent; m.pdidx = bx->h_gid; m.p = P;
            KGE_TRY(launch_rescal_matvec(m, s));
        } else {
        // (one pass over M per UNIQUE relation of the batch; the row blocks' parts of p are added by a one-thread-per-edge launch)
        RescalRelFwdArgs m{};
        m.B = B; m.D = d_e; m.UR = b->UR; m.rel = tbx->rel; m.ent = tbx->ent; m.hidx = bx->h_gid; m.tidx = bx->t_gid;
        m.ur_id = b->ur_id; m.ur_ptr = b->ur_ptr; m.ur_edge = b->ur_edge; m.counts_dev = b->counts_dev;
        m.V = b->neg_head ? A : RV; m.W = b->neg_head ? nullptr : A; m.ppart = Rpp; m.P = P;
        if (reg) {      // the regulariser's products ride on this pass (they live in the front of RC1, which the update's pass over
            //             the matrices overwrites only AFTER the traced rows' mean squares were taken from them)
            m.PV = RC1; m.PW = RC1 + (size_t)B * d_e; m.rho = RC1 + 2 * (size_t)B * d_e;
            m.reg_coef = hp->reg_coef; m.reg_norm = hp->reg_norm;
        }
        KGE_TRY(launch_rescal_rel_fwd(m, s));
        }
        if (dense_neg) {                 // pairwise fallback kernels read a dense copy of the negative rows
            EdgeFwdArgs nb{};
            nb.B = 0; nb.d_e = d_e; nb.d_r = d_e; nb.model = KGE_DISTMULT; nb.nbase = tbx->ent; nb.nidx = sh_dense ? bx->neg_ids : nids;
            nb.n_neg = CN; nb.Bn = Bn;
            if (nd) { nb.nd_own = b->neg_head ? b->h_gid : b->t_gid; nb.nd_chunk = chunk; nb.nd_Ns = b->N; }
            KGE_TRY(launch_edge_fwd(nb, s));
        }
    } else {
        if (build_prep) { *build_prep = ef; return KGE_OK; }
        if (!merged_fwd && !merged_pair) KGE_TRY(launch_edge_fwd(ef, s));     // merged: launched together with the forward tiles below
    }
    }   // PH_PREP

    bool fold_upd = false;                   // (TransE_l1: the update sums the backward's GN partials / GA parts, see PH_BWD)
    int fold_nrw = 0, fold_ga_parts = 1;
    if (phases & PH_SCORE) {
    // 2. chunked negative scores (+ per-16-column partial row statistics for the adversarial softmax)
    GemmArgs g{}; NegArgs na{};
    if (transr) {
        // scores already in S
    } else if (gemm) {
        if (dense_neg) fill_gemm(g, hp->model, C, chunk, N, d_e, hp->gamma, A, Bn, nullptr);
        else fill_gemm(g, hp->model, C, chunk, N, d_e, hp->gamma, A, tb->ent, nids);
        g.S = S; g.adv_temp = hp->adv_temp; g.asq = asq; g.bsq = bsq;
        g.lp = lp; g.w = b->edge_w;
        if (fused_loss) { g.PM = PM; g.PS = PS; g.PL = PL; g.Sraw = out ? out->neg_score : nullptr; }
        if (phases & PH_FWD) {
            bool fused_launch = false;
            if (co_update) {                  // async pipeline: forward GEMM(s) + update(s-1) in ONE launch
                const int rc = launch_neg_fwd_gemm_with_update(g, *co_update, s);
                if (rc == KGE_OK) fused_launch = true;
                else if (rc != KGE_ERR_ARG) return fail(rc, "launch_neg_fwd_gemm_with_update failed (%d)", rc);
                else KGE_TRY(launch_update(*co_update, s));      // no fused instantiation: one after the other
            }
            if (merged_fwd) {
                g.xbase = tb->ent; g.xidx = b->neg_head ? b->t_gid : b->h_gid; g.rbase = tb->rel; g.ridx = b->rel_ids;
                g.asign = b->neg_head ? -1.f : 1.f; g.lds_off = (hp->flags & KGE_FLAG_FWD_DIRECT) ? 1 : 0;
                if (fold_loss) {
                    LossArgs la{};
                    la.B = B; la.N = N; la.genre = hp->loss_genre; la.adv = hp->adv; la.pairwise = 0;
                    la.adv_temp = hp->adv_temp; la.margin = hp->margin;
                    la.pos = P; la.neg = S; la.w = b->edge_w; la.w_mean = b->edge_w_mean; la.dpos = dP; la.dneg = S;
                    la.row_pos = nullptr; la.row_neg = want4 ? row_neg : nullptr;      // (row_pos: written by the edge half)
                    la.acc = acc;
                    la.l2_scale = is_l2 ? 1 : 0; la.gamma = hp->gamma; la.clampv = clamp_of(hp->model);
                    la.neg_copy = out->neg_score;
                    la.skip_pos = 1;
                    KGE_TRY(launch_neg_fwd_gemm_with_edge_loss(g, ef, la, out->tickets, s));
                } else KGE_TRY(launch_neg_fwd_gemm_with_edge(g, ef, s, tail));
            } else if (!fused_launch) KGE_TRY(launch_neg_fwd_gemm(g, s));
        }
    } else {
        fill_pair(na, hp->model, C, chunk, N, d_e, hp->gamma, A, Bn, nullptr);
        na.S = S;
        if (phases & PH_FWD) {
            if (co_update) KGE_TRY(launch_update(*co_update, s));
            if (merged_pair) {
                NegArgs nf = na;                  // the forward gathers the negatives from the table (Bn is written by this launch)
                nf.nbase = tb->ent; nf.nidx = nids;
                nf.xbase = tb->ent; nf.xidx = b->neg_head ? b->t_gid : b->h_gid; nf.rbase = tb->rel; nf.ridx = b->rel_ids;
                nf.asign = b->neg_head ? -1.f : 1.f;
                KGE_TRY(launch_neg_fwd_bcast_with_edge(nf, ef, s));
            } else KGE_TRY(launch_neg_fwd_pair(na, s));
        }
    }

    // 3. stand-alone loss kernel (only when the loss is not fused into the backward GEMM)
    if (!fused_loss && !fold_loss && (phases & PH_FWD)) {
        LossArgs la{};
        la.B = B; la.N = N; la.genre = hp->loss_genre; la.adv = hp->adv; la.pairwise = hp->pairwise;
        la.adv_temp = hp->adv_temp; la.margin = hp->margin;
        la.pos = P; la.neg = S; la.w = b->edge_w; la.w_mean = b->edge_w_mean; la.dpos = dP; la.dneg = S;
        la.row_pos = want4 ? row_pos : nullptr; la.row_neg = want4 ? row_neg : nullptr;
        la.acc = acc;
        la.l2_scale = is_l2 ? 1 : 0; la.gamma = hp->gamma; la.clampv = clamp_of(hp->model);
        if (merged_fwd && is_l2) { la.l2_raw = 1; la.l2_chunk = chunk; la.asq = asq; la.bsq = bsq; }
        la.neg_copy = out ? out->neg_score : nullptr;
        la.skip_pos = (pairwise || rescal || transr) ? 0 : 1;  // RESCAL / TransR: no edge_fwd -> positive part here
        la.diag_chunk = nd ? chunk : 0;
        KGE_TRY(launch_loss(la, s));
    }

    bool fuse_gnred = false, ew_bwd = false;
    if (phases & PH_BWD) {
    // 4. gradients w.r.t. the pos-side vectors and the negative rows
    if (transr) {
        KGE_TRY(launch_transr_bwd(tr, s));       // dq, GN, per-edge projection gradients, relation-vector gradients
    } else if (gemm) {
        g.W = S; g.w = b->edge_w; g.lp = lp;      // fused loss: S holds u_ij and PM/PS/PL (set above) the partials
        if (merged_fwd && (hp->flags & KGE_FLAG_DENSE_BWD)) { g.nbase = Bn; g.nidx = nullptr; }   // dense_bwd (see PH_PREP)
        g.GA = GA; g.GN = GN;
        if (qfuse) {                              // TransE: the update reads Q = GA +/- P, GA itself only on request (g_rel output)
            g.Q = GT; g.QP = Pg; g.qc = b->neg_head ? 1.f : -1.f;
            if (!(out && out->g_rel)) g.GA = nullptr;
        }
        g.reg_coef = (reg && !nd) ? hp->reg_coef : 0.f; g.reg_norm = hp->reg_norm;   // nd: the update adds it (sampled rows only)
        g.row_neg = (fused_loss && want4) ? row_neg : nullptr;
        g.acc = fused_loss ? acc : nullptr;
        // DistMult / ComplEx / SimplE, strict step on local tables: the GA tiles write the per-edge gradient rows in their epilogue
        // (GemmArgs::ew_*) - no edge_bwd launch (5 -> 4 launches per step)
        ew_bwd = ((hp->model == KGE_DISTMULT && d_e % 4 == 0) ||
                  ((hp->model == KGE_COMPLEX || hp->model == KGE_SIMPLE) && d_e % 8 == 0)) &&
                 !pipelined && !co_prep && !nd && !sh && !fused_loss && !qfuse &&
                 d_r == d_e && !src.em.n && !src.rm.n && src.hidx && src.tidx && src.ridx &&
                 !(hp->flags & KGE_FLAG_NO_TRANSE_FAST);      // (the flag that keeps TransE on edge_bwd keeps DistMult there too)
        if (ew_bwd) {
            g.ew_ent = src.hbase; g.ew_rel = src.rbase; g.ew_h = src.hidx; g.ew_t = src.tidx; g.ew_r = src.ridx;
            g.ew_dpos = dP; g.ew_GH = GH; g.ew_GT = GT; g.ew_GR = GR;
            g.ew_reg_coef = reg ? hp->reg_coef : 0.f; g.ew_reg_norm = hp->reg_norm; g.ew_neg_head = b->neg_head;
            if (!(out && out->g_rel)) g.GA = nullptr;         // nobody reads GA itself then
        }
        bool fused_launch = false;
        if (co_prep) {                        // async pipeline: backward GEMM(s) + PREP(s+1) in ONE launch
            const int rc = launch_neg_bwd_gemm_with_prep(g, *co_prep, s);
            if (rc == KGE_OK) fused_launch = true;
            else if (rc != KGE_ERR_ARG) return fail(rc, "launch_neg_bwd_gemm_with_prep failed (%d)", rc);
        }
        if (!fused_launch) {
            KGE_TRY(launch_neg_bwd_gemm(g, s, tail));
            if (co_prep) KGE_TRY(launch_edge_fwd(*co_prep, s));  // no fused instantiation: one after the other
        }
    } else {
        na.W = S; na.GA = GA; na.GN = GN;
        na.reg_coef = (reg && !nd) ? hp->reg_coef : 0.f; na.reg_norm = hp->reg_norm;
        na.GNp = gemm ? nullptr : GNp;
        // shared-pair backward followed by edge_bwd: the sum of its GN partials shares the edge_bwd launch (same inputs' producer,
        // independent jobs).  Not with neg_deg_sample (edge_bwd reads GN) and not on the TransE fast path (no edge_bwd)
        // (round 4: on peer-to-peer sharded tables too - edge_bwd resolves its rows through the shard map, the reduction is local)
        fuse_gnred = na.GNp && !nd && !transe_fast && !rescal && !transr && d_e % 4 == 0 && na.N % 4 == 0 &&
                     (hp->model == KGE_ROTATE || hp->model == KGE_TRANSE_L1) && !(hp->flags & KGE_FLAG_TWO_PASS_PAIR) &&
                     !(hp->flags & KGE_FLAG_SPLIT_FWD);
        // TransE_l1 (fast path: no edge_bwd launch to share): the update kernel sums the GN partials and GA parts itself - the
        // stand-alone reduction launch (5.2 us + a boundary) leaves the strict step.  One-call strict step on local tables, no
        // gradient outputs (they read the summed buffers).
        fold_nrw = neg_bwd_lc_nrw(hp->model, C, chunk, d_e);
        fold_upd = phases == PH_ALL && na.GNp && transe_fast && hp->model == KGE_TRANSE_L1 && !nd && !sh && !emit && !build_update &&
                   !co_update && !co_prep && !(out && (out->g_neg || out->g_rel || out->g_pos_ent)) && d_e % 4 == 0 && d_e <= 512 &&
                   na.N % 4 == 0 && neg_bwd_lc_supported(hp->model, d_e) && fold_nrw <= 6 && ga_parts <= 4 &&
                   !(hp->flags & (KGE_FLAG_TWO_PASS_PAIR | KGE_FLAG_SPLIT_FWD)) && (!reg || hp->reg_norm == 3);
        na.defer_reduce = (fuse_gnred || fold_upd) ? 1 : 0;
        // GA in parts only where the shared-pair kernel runs: its stand-alone partial reduction adds them up in place, the
        // edge_bwd launch that carries the reduction (fuse_gnred) adds them while it reads the row
        const bool ga_split = ga_parts > 1 && na.GNp && !na.nidx && na.N % 4 == 0 && !rescal && !transr &&
                              neg_bwd_lc_supported(hp->model, d_e);
        na.ga_parts = ga_split ? ga_parts : 1; na.ga_stride = (int64_t)B * d_e;
        fold_upd = fold_upd && !na.nidx;         // (gathered negatives run the two-pass kernels: no partials)
        if (!fold_upd && !fuse_gnred) na.defer_reduce = 0;
        fold_ga_parts = na.ga_parts;
        KGE_TRY(launch_neg_bwd_pair(na, s));
        if (co_prep) KGE_TRY(launch_edge_fwd(*co_prep, s));
    }

    // 5. per-edge gradients of head / tail / relation rows (TransE rebuilds them in the update)
    const float *Vr = rescal ? (b->neg_head ? A : RV) : nullptr;           // RESCAL: M t
    if (transr) {
        // entity gradients through the projections: GH = P (-dp s) (+ P dq, tail mode), GT = P (+dp s) (+ P dq, head mode)
        RescalMatvecArgs m{};
        m.B = B; m.D = d_e; m.Dc = d_r; m.rel = tbx->proj; m.ridx = b->rel_ids;
        m.y1 = tr.SG; m.r1 = TR1; m.y2 = tr.DQ; m.r2 = TR2;
        KGE_TRY(launch_rescal_matvec(m, s));
        KGE_TRY(launch_rescal_axpy(dP, TR1, b->neg_head ? nullptr : TR2, B, d_e, GH, s, -1.f));
        KGE_TRY(launch_rescal_axpy(dP, TR1, b->neg_head ? TR2 : nullptr, B, d_e, GT, s, 1.f));
        // neg_deg_sample: the in-batch negative rows are slices of the positive trace - their gradient joins GH (head mode) / GT
        if (nd) KGE_TRY(launch_nd_fold(GN, b->neg_head ? GH : GT, B, chunk, N, d_e, s));
        // projection table first: the entity update below changes the h / t rows its rank-1 trace reads
        KGE_TRY(launch_transr_proj_update(tr, s));
    } else if (rescal) {
        // M^T h and M^T GA come out of the relation update's own pass over M (one per unique relation, row-block parts);
        // GH = dp M t (+ M^T GA, tail mode), GT = dp M^T h (+ M^T GA, head mode) are combined after it;  the relation
        // gradient stays factored.  (d_e not a multiple of 4: a pass over M per edge for the two products, then the update's own)
        if (!rescal_rel) {
            RescalMatvecArgs m{};
            m.B = B; m.D = d_e; m.rel = tbx->rel; m.ridx = b->rel_ids;
            m.z1 = tbx->ent; m.z1idx = bx->h_gid; m.c1 = RC1;
            m.z2 = GA; m.c2 = RC2;
            KGE_TRY(launch_rescal_matvec(m, s));
            KGE_TRY(launch_rescal_axpy(dP, Vr, b->neg_head ? nullptr : RC2, B, d_e, GH, s));
            KGE_TRY(launch_rescal_axpy(dP, RC1, b->neg_head ? RC2 : nullptr, B, d_e, GT, s));
            if (nd) KGE_TRY(launch_nd_fold(GN, b->neg_head ? GH : GT, B, chunk, N, d_e, s));
        }
        if (out && out->g_rel) {        // test / debugging output: materialise dp h t^T + GA x^T + regulariser
            RescalOuterArgs o{};
            o.B = B; o.D = d_e; o.c = dP; o.u = tbx->ent; o.uidx = bx->h_gid; o.v = tbx->ent; o.vidx = bx->t_gid;
            o.G = out->g_rel;
            KGE_TRY(launch_rescal_outer(o, s));
            o.c = nullptr; o.u = GA; o.uidx = nullptr; o.vidx = b->neg_head ? bx->t_gid : bx->h_gid; o.accumulate = 1;
            if (reg) { o.rel = tbx->rel; o.ridx = b->rel_ids; o.reg_coef = hp->reg_coef; o.reg_norm = hp->reg_norm; }
            KGE_TRY(launch_rescal_outer(o, s));
        }
        // relation matrices first: the entity update below changes the h / t rows this kernel reads
        RescalUpdateArgs ru{};
        ru.B = B; ru.D = d_e; ru.UE = b->UE; ru.UR = b->UR; ru.neg_head = b->neg_head; ru.reg_norm = hp->reg_norm;
        ru.rel_ids = b->rel_ids; ru.gs = Rgs; ru.inv_std = Rstd; ru.reg_part = (want4 || (reg && acc)) ? Rreg : nullptr;
        ru.lr = hp->lr; ru.eps = hp->eps; ru.reg_coef = reg ? hp->reg_coef : 0.f;
        ru.rel = tbx->rel; ru.rel_state = tbx->rel_state; ru.ent = tbx->ent; ru.hidx = bx->h_gid; ru.tidx = bx->t_gid;
        ru.dpos = dP; ru.GA = GA; ru.ur_id = b->ur_id; ru.ur_ptr = b->ur_ptr; ru.ur_edge = b->ur_edge;
        ru.counts_dev = b->counts_dev; ru.reg_rel = want4 ? reg_rel : nullptr; ru.acc = acc;
        if (rescal_rel) { ru.c1p = RC1; ru.c2p = RC2; }
        if (rescal_rel && reg) { ru.PV = RC1; ru.PW = RC1 + (size_t)B * d_e; ru.rho = RC1 + 2 * (size_t)B * d_e; }
        KGE_TRY(launch_rescal_update_rel(ru, s));
        if (rescal_rel) {
            RescalCombineArgs cb{};
            cb.B = B; cb.D = d_e; cb.neg_head = b->neg_head; cb.dpos = dP; cb.V = Vr; cb.c1p = RC1; cb.c2p = RC2; cb.GH = GH; cb.GT = GT;
            KGE_TRY(launch_rescal_combine(cb, s));
            if (nd) KGE_TRY(launch_nd_fold(GN, b->neg_head ? GH : GT, B, chunk, N, d_e, s));
        }
    } else if (!transe_fast && !ew_bwd) {
        EdgeBwdArgs eb{};
        eb.src = src_bwd; eb.B = B; eb.d_e = d_e; eb.d_r = d_r; eb.neg_head = b->neg_head; eb.model = hp->model;
        eb.gamma = hp->gamma; eb.rot_div = rot_div;
        eb.dpos = dP; eb.GA = GA; eb.reg_coef = reg ? hp->reg_coef : 0.f; eb.reg_norm = hp->reg_norm;
        eb.GH = GH; eb.GT = GT; eb.GR = GR;
        eb.ga_parts = fuse_gnred ? na.ga_parts : 1; eb.ga_stride = na.ga_stride;
        if (nd) { eb.GNd = GN; eb.nd_chunk = chunk; eb.nd_Np = N; }   // in-batch negative rows -> positive trace
        if (fuse_gnred) {
            const int rc = launch_edge_bwd_with_gn_reduce(eb, na, neg_bwd_lc_nrw(hp->model, C, chunk, d_e), s);
            if (rc != KGE_OK) return fail(rc, "launch_edge_bwd_with_gn_reduce failed (%d)", rc);
        } else KGE_TRY(launch_edge_bwd(eb, s));
    }

    if (transe_fast && out && out->g_rel) {
        // the per-edge relation gradient is not materialised on the fast path; rebuild it for the
        // caller with the generic kernel BEFORE the tables change (test / debugging output only)
        EdgeBwdArgs eb{};
        eb.src = src_bwd; eb.B = B; eb.d_e = d_e; eb.d_r = d_r; eb.neg_head = b->neg_head; eb.model = hp->model;
        eb.gamma = hp->gamma; eb.rot_div = rot_div;
        eb.dpos = dP; eb.GA = GA; eb.reg_coef = reg ? hp->reg_coef : 0.f; eb.reg_norm = hp->reg_norm;
        eb.GH = nullptr; eb.GT = nullptr; eb.GR = out->g_rel;
        KGE_TRY(launch_edge_bwd(eb, s));
    }
    }   // PH_BWD
    }   // PH_SCORE

    if (phases & PH_UPDATE) {

    // 6. owner-computes Adagrad on both tables (or gradient emission for sharded training)
    UpdateArgs ua{};
    ua.model_d_e = d_e; ua.d_r = rescal ? d_e : d_r; ua.reg_norm = hp->reg_norm;
    ua.UE = (phases & PH_UPD_ENT) ? b->UE : 0;            // the async pipeline applies the two tables' traces separately
    ua.UR = (rescal || !(phases & PH_UPD_REL)) ? 0 : b->UR;
    ua.lr = hp->lr; ua.eps = hp->eps; ua.reg_coef = reg ? hp->reg_coef : 0.f;
    ua.ent = tb->ent; ua.ent_state = tb->ent_state; ua.rel = tb->rel; ua.rel_state = tb->rel_state;
    ua.em = em; ua.rm = rm;
    if (sh && sh->rel_local) { ua.rel = sh->rel_local; ua.rel_state = sh->rel_state_local; }    // (rm.n == 0: rank-local relation table)
    ua.ue_id = b->ue_id; ua.ue_pos_ptr = b->ue_pos_ptr; ua.ue_pos_adj = b->ue_pos_adj;
    ua.ue_neg_ptr = b->ue_neg_ptr; ua.ue_neg_slot = b->ue_neg_slot;
    ua.ur_id = b->ur_id; ua.ur_ptr = b->ur_ptr; ua.ur_edge = b->ur_edge;
    ua.ue_rec = b->ue_rec; ua.ur_rec = b->ur_rec; ua.counts_dev = b->counts_dev;
    ua.GH = GH; ua.GT = GT; ua.GN = GN; ua.GR = GR;
    ua.transe_fast = transe_fast ? 1 : 0; ua.neg_head = b->neg_head; ua.P = Pg; ua.GA = GA;
    ua.Q = qfuse ? GT : nullptr;
    ua.reg_ent = want4 ? reg_ent : nullptr; ua.reg_rel = want4 ? reg_rel : nullptr;
    ua.acc = acc;
    ua.ld_e = d_e; ua.ld_r = d_r; ua.ld_gs_e = 1; ua.ld_gs_r = 1;
    if (nd) { ua.nd_chunk = chunk; ua.nd_Ns = b->N; ua.nd_Np = N; }
    if (fold_upd) {
        ua.GNp = GNp; ua.gn_parts = fold_nrw; ua.gn_stride = (int64_t)CN * d_e;
        ua.gn_reg_coef = reg ? hp->reg_coef : 0.f; ua.gn_reg_norm = hp->reg_norm;
        ua.ga_parts = fold_ga_parts; ua.ga_stride = (int64_t)B * d_e;
    }
    if (async && reg) {
        // the regulariser of the positive-trace rows is part of the gradient the reference computes in forward, from the
        // rows as gathered: evaluate it on PREP's copies.  (The relation trace is only deferred with KGE_FLAG_ASYNC_REL;
        // otherwise it lands before anything else touches the relation table and the current row IS the gathered row.)
        ua.Hs = Hc; ua.Ts = Tc;
        if (nd) ua.Ns = Bn;            // the update adds the regulariser of the sampled negative rows in this mode
        ua.Rs = ((hp->flags & KGE_FLAG_ASYNC_REL) && transe_fast) ? Rc : nullptr;
    }
    if (emit) {
        ua.g0 = emit->g0; ua.gs0 = emit->gs0; ua.g1 = emit->g1; ua.gs1 = emit->gs1;
        ua.gr = emit->gr; ua.gsr = emit->gsr; ua.rid = emit->rid;
        ua.emit_ent = 1; ua.emit_rel = emit->gr ? 1 : 0;
        ua.emit_by_id = emit->ent_by_id ? 1 : 0;
        if (emit->msg_rows) {                     // packed single-trace entity messages (ABI 8)
            if (!emit->g0 || emit->ld_e < d_e + 4 || emit->msg_cap < 1 || emit->msg_cap_extra < 1)
                return fail(KGE_ERR_ARG, "kge_step_grads: packed messages need g0, ld_e >= d_e + 4 and the bucket geometry");
            ua.msg_rows = emit->msg_rows; ua.msg_cap = emit->msg_cap; ua.msg_capT = emit->msg_cap + emit->msg_cap_extra;
            ua.g1 = nullptr; ua.gs0 = nullptr; ua.gs1 = nullptr;
        }
        if (emit->ld_e > 0) { ua.ld_e = emit->ld_e; ua.ld_gs_e = emit->ld_e; }
        if (emit->ld_r > 0) { ua.ld_r = emit->ld_r; ua.ld_gs_r = emit->ld_r; }
    }
    if (out && out->g_pos_ent) {
        if (emit && emit->ld_e > 0) return fail(KGE_ERR_ARG, "g_pos_ent output cannot be combined with a strided emit");
        ua.g0 = out->g_pos_ent;
    }
    if (build_update) { *build_update = ua; return KGE_OK; }
    KGE_TRY(launch_update(ua, s, tail));
    // 7. deterministic reduction of this step's loss terms (only when the caller wants the
    //    per-step values; running sums are accumulated by the kernels above without it)
    if (want4 && (phases & PH_UPD_ENT)) {
        FinalizeArgs f{};
        f.B = B; f.UE = b->UE; f.UR = b->UR; f.pairwise = hp->pairwise;
        f.row_pos = row_pos; f.row_neg = row_neg; f.reg_ent = reg_ent; f.reg_rel = reg_rel;
        f.counts_dev = b->counts_dev;
        f.loss4 = out->loss4;
        KGE_TRY(launch_finalize(f, s));
    }
    }   // PH_UPDATE
    return KGE_OK;
}

int kge_step_fused(const kge_hparams *hp, const kge_tables *tb, const kge_batch *b,
                   const kge_step_out *out, void *ws, size_t ws_bytes, void *stream) {
    return step_impl(hp, tb, b, out, nullptr, ws, ws_bytes, stream);
}

// ---- round 5: the strict step that also BUILDS a batch of the next group (sampler tail workgroups, kge_sampler_tail.hpp) ----
size_t kge_sampler_tail_scratch_bytes(int B, int C, int N, int64_t n_ent) {
    if (B <= 0 || C <= 0 || N <= 0) return 0;
    return (size_t)tail_scratch(B, C * N, n_ent > ((int64_t)1 << (32 - SP_CODE_BITS))).total;
}

int kge_step_fused_sampling(const kge_hparams *hp, const kge_tables *tb, const kge_batch *b, const kge_step_out *out, void *ws,
                            size_t ws_bytes, const kge_sampler_job *job, void *stream) {
    if (!job) return step_impl(hp, tb, b, out, nullptr, ws, ws_bytes, stream);
    if (!job->heads || !job->rels || !job->tails || !job->state || !job->slot || !job->scratch || job->n_train <= 0 || job->n_ent <= 0 ||
        job->B <= 0 || job->C <= 0 || job->chunk <= 0 || job->N <= 0 || job->C * job->chunk != job->B || job->k < 0 || job->advance < 0)
        return fail(KGE_ERR_ARG, "kge_step_fused_sampling: bad sampler job");
    if (job->n_train < job->B) return fail(KGE_ERR_ARG, "kge_step_fused_sampling: fewer training triples than one batch (n_train < batch)");
    if (2 * job->B + job->C * job->N > SP_MAXE || job->B > SP_MAXE / 2 || job->n_ent >= ((int64_t)1 << 51))
        return fail(KGE_ERR_ARG, "kge_step_fused_sampling: 2 * batch + chunks * neg <= 4096 (larger batches: build the plan on the host)");
    if (job->scratch_bytes < kge_sampler_tail_scratch_bytes(job->B, job->C, job->N, job->n_ent))
        return fail(KGE_ERR_WORKSPACE, "kge_step_fused_sampling: scratch too small (kge_sampler_tail_scratch_bytes)");
    SmpTail t{};
    t.a.H = job->heads; t.a.R = job->rels; t.a.T = job->tails; t.a.perm = job->perm; t.a.n_train = job->n_train; t.a.n_ent = job->n_ent;
    t.a.B = job->B; t.a.C = job->C; t.a.chunk = job->chunk; t.a.N = job->N; t.a.seed = job->seed; t.a.state = job->state;
    t.a.slots = (char *)job->slot; t.a.slot_bytes = 0; t.a.preperm = job->pre_permuted ? 1 : 0; t.slot3 = (char *)job->prev_slot;
    t.scratch = (char *)job->scratch; t.k = job->k; t.advance = job->advance; t.phase = 1;
    return step_impl(hp, tb, b, out, nullptr, ws, ws_bytes, stream, nullptr, PH_ALL, nullptr, nullptr, nullptr, nullptr, &t);
}

// ------------------------------------------------------------------------------------------
// --async_update: the update of step s-1 overlaps the scoring of step s (two HIP streams)
// ------------------------------------------------------------------------------------------
struct kge_pipe {
    bool pending;        // a step has been scored but its (entity) update is not enqueued yet
    int parity;          // workspace half of the NEXT step
    int upd_phases;      // what the pending update covers (entity trace, or entity + relation with KGE_FLAG_ASYNC_REL)
    kge_hparams hp; kge_tables tb; kge_batch b; kge_step_out out; bool has_out;
    void *ws; size_t ws_bytes;     // workspace half of the pending step
    bool prepped;        // PREP of the next step already ran (inside the previous call's backward launch) ...
    const void *prep_key;          // ... for the batch whose h_gid array is this,
    kge_hparams prep_hp;           // ... with these hyper-parameters (it is reused only by a call that asks for the same plain step)
};

int kge_pipe_create(kge_pipe **pipe) {
    if (!pipe) return fail(KGE_ERR_ARG, "kge_pipe_create: null argument");
    kge_pipe *p = new (std::nothrow) kge_pipe();
    if (!p) return fail(KGE_ERR_LAUNCH, "kge_pipe_create: out of host memory");
    *pipe = p;
    return KGE_OK;
}

int kge_pipe_destroy(kge_pipe *p) {
    delete p;
    return KGE_OK;
}

size_t kge_step_async_workspace_bytes(const kge_hparams *hp, int B, int C, int chunk, int N, int UE, int UR) {
    // two halves (the update of step s-1 reads its gradients while step s fills the other half), each with room
    // for the dense h / t / r copies
    const size_t one = kge_step_workspace_bytes(hp, B, C, chunk, N, UE, UR) +
                       2 * align_up((size_t)B * hp->d_e * sizeof(float)) + align_up((size_t)B * hp->d_r * sizeof(float));
    return 2 * align_up(one);
}

int kge_step_async(kge_pipe *p, const kge_hparams *hp, const kge_tables *tb, const kge_batch *b, const kge_batch *b_next,
                   const kge_step_out *out, void *ws, size_t ws_bytes, void *stream) {
    if (!p || !hp || !tb || !b || !ws) return fail(KGE_ERR_ARG, "kge_step_async: null argument");
    const size_t half = (ws_bytes / 2) & ~(size_t)255;
    void *wsp = (char *)ws + (size_t)p->parity * half;
    void *wsn = (char *)ws + (size_t)(p->parity ^ 1) * half;
    const bool defer_rel = (hp->flags & KGE_FLAG_ASYNC_REL) != 0;
    // PREP(s): gathers the rows (update s-2 has landed: stream order) and makes the dense copies the rest of the step
    // reads - unless the previous call already ran it inside its backward launch
    // (a PREP that ran ahead was built for a plain step - no per-step outputs - with the previous call's hyper-parameters: it
    //  serves this call only if this call wants exactly that; otherwise PREP runs again)
    const bool plain_now = !out || (!out->loss4 && !out->pos_score && !out->neg_score && !out->g_pos_ent && !out->g_neg && !out->g_rel);
    const bool reuse = p->prepped && p->prep_key == (const void *)b->h_gid && plain_now &&
                       memcmp(&p->prep_hp, hp, sizeof(kge_hparams)) == 0;
    if (!reuse)
        if (int rc = step_impl(hp, tb, b, out, nullptr, wsp, half, stream, nullptr, PH_PREP)) return rc;
    p->prepped = false;
    // launch A: forward(s) || UPDATE(s-1)
    UpdateArgs co{};
    bool have_co = false;
    if (p->pending) {
        p->pending = false;
        const kge_step_out *po = p->has_out ? &p->out : nullptr;
        if (po && po->loss4) {         // per-step loss wanted: the reduction kernel follows the update - keep them together
            if (int rc = step_impl(&p->hp, &p->tb, &p->b, po, nullptr, p->ws, p->ws_bytes, stream, nullptr, p->upd_phases)) return rc;
        } else {
            if (int rc = step_impl(&p->hp, &p->tb, &p->b, po, nullptr, p->ws, p->ws_bytes, stream, nullptr, p->upd_phases, &co)) return rc;
            have_co = true;
        }
    }
    if (int rc = step_impl(hp, tb, b, out, nullptr, wsp, half, stream, nullptr, PH_FWD, nullptr, have_co ? &co : nullptr)) return rc;
    // launch C: backward(s) || PREP(s+1).  Only when nothing touches the tables between this backward and the next PREP
    // (relation trace deferred too), for the plain step (no per-step outputs)
    EdgeFwdArgs ef{};
    bool have_prep = false;
    const bool plain_out = !out || (!out->loss4 && !out->pos_score && !out->neg_score && !out->g_pos_ent && !out->g_neg && !out->g_rel);
    if (b_next && defer_rel && plain_out) {
        if (int rc = step_impl(hp, tb, b_next, out, nullptr, wsn, half, stream, nullptr, PH_PREP, nullptr, nullptr, &ef)) return rc;
        have_prep = true;
    }
    if (int rc = step_impl(hp, tb, b, out, nullptr, wsp, half, stream, nullptr, PH_BWD, nullptr, nullptr, nullptr,
                           have_prep ? &ef : nullptr)) return rc;
    if (have_prep) { p->prepped = true; p->prep_key = (const void *)b_next->h_gid; p->prep_hp = *hp; }
    // the reference defers the ENTITY table only (general_models.py:639-647: create_async_update on entity_emb;
    // relation_emb.update stays in the training loop): relation trace now
    if (!defer_rel)
        if (int rc = step_impl(hp, tb, b, out, nullptr, wsp, half, stream, nullptr, PH_UPD_REL)) return rc;
    p->upd_phases = defer_rel ? PH_UPDATE : PH_UPD_ENT;
    p->hp = *hp; p->tb = *tb; p->b = *b; p->has_out = out != nullptr;
    if (out) p->out = *out;
    p->ws = wsp; p->ws_bytes = half;
    p->pending = true;
    p->parity ^= 1;
    return KGE_OK;
}

int kge_step_async_flush(kge_pipe *p, void *stream) {
    if (!p) return fail(KGE_ERR_ARG, "kge_step_async_flush: null argument");
    if (p->pending) {
        p->pending = false;
        if (int rc = step_impl(&p->hp, &p->tb, &p->b, p->has_out ? &p->out : nullptr, nullptr, p->ws, p->ws_bytes, stream, nullptr,
                               p->upd_phases)) return rc;
    }
    p->prepped = false;         // a PREP that ran ahead gathered rows without this update: it must not be reused
    return KGE_OK;
}

int kge_step_phase(const kge_hparams *hp, const kge_tables *tb, const kge_batch *b, const kge_step_out *out,
                   void *ws, size_t ws_bytes, int phases, void *stream) {
    // the strict step, one phase group at a time (same kernels, same order: calling the four groups in sequence IS
    // kge_step_fused); exists so that the caller can put events between the groups (the reference's per-phase timers)
    if (phases <= 0 || phases > (KGE_PHASE_GATHER | KGE_PHASE_FORWARD | KGE_PHASE_BACKWARD | KGE_PHASE_UPDATE))
        return fail(KGE_ERR_ARG, "kge_step_phase: bad phase mask %d", phases);
    int ph = 0;
    if (phases & KGE_PHASE_GATHER) ph |= PH_PREP;
    if (phases & KGE_PHASE_FORWARD) ph |= PH_FWD;
    if (phases & KGE_PHASE_BACKWARD) ph |= PH_BWD;
    if (phases & KGE_PHASE_UPDATE) ph |= PH_UPDATE;
    return step_impl(hp, tb, b, out, nullptr, ws, ws_bytes, stream, nullptr, ph | PH_STRICT);
}

int kge_step_grads(const kge_hparams *hp, const kge_tables *tb, const kge_batch *b,
                   const kge_step_out *out, const kge_emit *emit, void *ws, size_t ws_bytes,
                   void *stream) {
    if (!emit || !emit->g0 || (!emit->msg_rows && (!emit->gs0 || !emit->g1 || !emit->gs1)))
        return fail(KGE_ERR_ARG, "kge_step_grads: emit buffers g0/gs0/g1/gs1 are required (packed messages: g0 + msg_rows)");
    if ((emit->gr == nullptr) != (emit->gsr == nullptr))
        return fail(KGE_ERR_ARG, "kge_step_grads: gr and gsr must be given together");
    return step_impl(hp, tb, b, out, emit, ws, ws_bytes, stream);
}

size_t kge_rank_workspace_bytes(int Eb, int64_t n_cand, int d_e) {
    size_t n = 0;
    n += align_up((size_t)Eb * d_e * sizeof(float));        // A
    n += 2 * align_up((size_t)Eb * sizeof(float));          // asq, P
    n += align_up((size_t)n_cand * sizeof(float));          // bsq
    n += align_up(std::max((size_t)Eb * (size_t)n_cand * sizeof(float), rank_gemm_mask_bytes(Eb, n_cand)));   // S / the comparison mask
    n += align_up((size_t)Eb * d_e * sizeof(float));        // V = M t (RESCAL)
    n += 4 * align_up((size_t)Eb * 1024 * sizeof(float));   // TransR: hp, tp, q, sign rows (d_r <= 1024)
    return n;
}

int kge_rank_eval(int model, int neg_head, const float *ent, int64_t n_ent, const float *rel,
                  int64_t n_rel, const int64_t *h, const int64_t *r, const int64_t *t, int64_t E,
                  int d_e, int d_r, float gamma, float emb_init, const int64_t *cand, int64_t n_cand,
                  const int64_t *filt_ptr, const int64_t *filt_ids, int Eb, int32_t *ranks,
                  float *pos_score_out, void *ws, size_t ws_bytes, unsigned flags, void *stream) {
    return kge_rank_eval_ex(model, neg_head, ent, n_ent, rel, n_rel, nullptr, h, r, t, E, d_e, d_r, gamma, emb_init, cand,
                            n_cand, filt_ptr, filt_ids, Eb, ranks, pos_score_out, ws, ws_bytes, flags, stream);
}

int kge_rank_eval_ex(int model, int neg_head, const float *ent, int64_t n_ent, const float *rel,
                     int64_t n_rel, const float *proj, const int64_t *h, const int64_t *r, const int64_t *t,
                     int64_t E, int d_e, int d_r, float gamma, float emb_init, const int64_t *cand,
                     int64_t n_cand, const int64_t *filt_ptr, const int64_t *filt_ids, int Eb, int32_t *ranks,
                     float *pos_score_out, void *ws, size_t ws_bytes, unsigned flags, void *stream) {
    if (int rc = check_model(model, d_e, d_r)) return rc;
    if (model == KGE_TRANSR && (!proj || !cand))
        return fail(KGE_ERR_ARG, "kge_rank_eval_ex: TransR needs the projection table and an explicit candidate list");
    if (!ent || !rel || n_ent <= 0 || n_rel <= 0 || E < 0 || (E && (!h || !r || !t || !ranks)) || !ws || Eb <= 0)
        return fail(KGE_ERR_ARG, "kge_rank_eval: bad argument");
    if ((filt_ptr == nullptr) != (filt_ids == nullptr))
        return fail(KGE_ERR_ARG, "kge_rank_eval: filt_ptr and filt_ids must be given together");
    const int64_t N = cand ? n_cand : n_ent;
    if (N <= 0 || N > 0x7fffffff) return fail(KGE_ERR_ARG, "kge_rank_eval: bad candidate count %lld", (long long)N);
    if (E == 0) return KGE_OK;
    hipStream_t s = (hipStream_t)stream;
    Carver cv(ws, ws_bytes);
    float *A = cv.f((size_t)Eb * d_e), *asq = cv.f(Eb), *P = cv.f(Eb), *bsq = cv.f((size_t)N);
    float *S = cv.f(std::max((size_t)Eb * (size_t)N, rank_gemm_mask_bytes(Eb, N) / sizeof(float)));
    float *RV = cv.f((size_t)Eb * d_e);
    float *THP = cv.f((size_t)Eb * 1024), *TTP = cv.f((size_t)Eb * 1024), *TQ = cv.f((size_t)Eb * 1024), *TSG = cv.f((size_t)Eb * 1024);
    if (!cv.ok()) return fail(KGE_ERR_WORKSPACE, "kge_rank_eval: workspace too small (%zu < %zu)", ws_bytes,
                              kge_rank_workspace_bytes(Eb, N, d_e));
    const bool gemm = use_mfma(model, d_e, (int)N, flags);
    const bool l2g = gemm && model == KGE_TRANSE_L2;
    const float rot_div = rot_div_of(emb_init);
    if (l2g) {       // |b|^2 of every candidate row, once
        EdgeFwdArgs nb{};
        nb.B = 0; nb.d_e = d_e; nb.d_r = d_r; nb.model = model; nb.nbase = ent; nb.nidx = cand; nb.n_neg = (int)N;
        nb.bsq = bsq;
        KGE_TRY(launch_edge_fwd(nb, s));
    }
    for (int64_t e0 = 0; e0 < E; e0 += Eb) {
        const int rows = (int)((E - e0) < Eb ? (E - e0) : Eb);
        EdgeFwdArgs ef{};
        ef.src = EdgeSrc{ent, h + e0, ent, t + e0, rel, r + e0};
        ef.B = rows; ef.d_e = d_e; ef.d_r = d_r; ef.neg_head = neg_head; ef.model = model;
        ef.gamma = gamma; ef.rot_div = rot_div;
        ef.pos_score = pos_score_out ? pos_score_out + e0 : P; ef.A = A; ef.asq = l2g ? asq : nullptr;
        if (model == KGE_TRANSR) {
            // the training kernels with one chunk = this batch of test triples and the candidates as negatives
            RescalMatvecArgs m{};
            m.B = rows; m.D = d_e; m.Dc = d_r; m.rel = proj; m.ridx = r + e0;
            m.z1 = ent; m.z1idx = h + e0; m.c1 = THP; m.z2 = ent; m.z2idx = t + e0; m.c2 = TTP;
            KGE_TRY(launch_rescal_matvec(m, s));
            TransRArgs tr{};
            tr.B = rows; tr.C = 1; tr.chunk = rows; tr.N = (int)N; tr.De = d_e; tr.Dr = d_r; tr.neg_head = neg_head;
            tr.gamma = gamma; tr.ent = ent; tr.h_gid = h + e0; tr.t_gid = t + e0; tr.neg_ids = cand; tr.rel_ids = r + e0;
            tr.rel = rel; tr.proj = const_cast<float *>(proj);
            tr.HP = THP; tr.TP = TTP; tr.Q = TQ; tr.SG = TSG; tr.P = ef.pos_score; tr.S = S; tr.Z = nullptr;
            KGE_TRY(launch_transr_pos(tr, s));
            KGE_TRY(launch_transr_fwd(tr, s));
        } else if (model == KGE_RESCAL) {
            RescalMatvecArgs m{};
            m.B = rows; m.D = d_e; m.rel = rel; m.ridx = r + e0;
            m.y1 = ent; m.y1idx = t + e0; m.r1 = neg_head ? A : RV;
            if (!neg_head) { m.y2 = ent; m.y2idx = h + e0; m.r2 = A; }
            m.pd = ent; m.pdidx = h + e0; m.p = ef.pos_score;
            KGE_TRY(launch_rescal_matvec(m, s));
        } else {
            KGE_TRY(launch_edge_fwd(ef, s));
        }
        if (model == KGE_TRANSR) {
            // scores already in S
        } else if (gemm && rank_gemm_supported(model, d_e)) {
            // one tiled GEMM per batch whose epilogue keeps the comparison bits; ranks from the mask (kge_rank_gemm.hip)
            KGE_TRY(launch_rank_gemm(model, A, rows, ent, cand, N, d_e, gamma, clamp_of(model), asq, bsq,
                                     pos_score_out ? pos_score_out + e0 : P, S, filt_ptr, filt_ids, e0, ranks, s));
            continue;
        } else if (gemm) {
            GemmArgs g; fill_gemm(g, model, 1, rows, (int)N, d_e, gamma, A, ent, cand);
            g.S = S; g.asq = asq; g.bsq = bsq;
            KGE_TRY(launch_neg_fwd_gemm(g, s));
        } else {
            NegArgs na; fill_pair(na, model, 1, rows, (int)N, d_e, gamma, A, ent, cand);
            na.S = S;
            KGE_TRY(launch_neg_fwd_pair(na, s));
        }
        KGE_TRY(launch_rank_count(S, pos_score_out ? pos_score_out + e0 : P, rows, N, filt_ptr, filt_ids, e0, ranks, s));
    }
    return KGE_OK;
}

int kge_step_sharded(const kge_hparams *hp, const kge_shards *sh, const kge_batch *b,
                     const kge_step_out *out, void *ws, size_t ws_bytes, void *stream) {
    if (!sh) return fail(KGE_ERR_ARG, "kge_step_sharded: null shard map");
    // (RESCAL's relation "row" is a d_e x d_e matrix streamed by its own kernels: only the entity width is bounded)
    if (hp && (hp->d_e % 4 || hp->d_e > 1024 || (hp->model != KGE_RESCAL && (hp->d_r % 4 || hp->d_r > 1024))))
        return fail(KGE_ERR_ARG, "kge_step_sharded: row widths must be multiples of 4 and <= 1024 floats");
    kge_tables tb{};
    tb.n_ent = sh->n_ent; tb.n_rel = sh->n_rel;
    if (sh->rel_local) {              // ABI 8: relation-side tables local to this rank (relation ids resolve to them directly)
        tb.rel = sh->rel_local; tb.rel_state = sh->rel_state_local;
        tb.proj = sh->proj_local; tb.proj_state = sh->proj_state_local;
    }
    return step_impl(hp, &tb, b, out, nullptr, ws, ws_bytes, stream, sh);
}

int kge_gather_rows_sharded(float *const *shard_rows, int n_shards, int64_t rows_per_shard, int dim,
                            const int64_t *idx, int64_t n_idx, float *out, void *stream) {
    if (!shard_rows || n_shards < 1 || rows_per_shard <= 0 || dim <= 0 || n_idx < 0 || (n_idx && (!idx || !out)))
        return fail(KGE_ERR_ARG, "kge_gather_rows_sharded: bad argument");
    KGE_TRY(launch_gather_rows_sharded(shard_rows, n_shards, rows_per_shard, dim, idx, n_idx, out, (hipStream_t)stream));
    return KGE_OK;
}

int kge_ipc_export(const void *dev_ptr, void *handle64, int64_t *offset) {
    static_assert(sizeof(hipIpcMemHandle_t) == KGE_IPC_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    if (!dev_ptr || !handle64 || !offset) return fail(KGE_ERR_ARG, "kge_ipc_export: null argument");
    hipDeviceptr_t base = nullptr; size_t size = 0;
    hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)dev_ptr);
    if (e != hipSuccess) return fail(KGE_ERR_LAUNCH, "hipMemGetAddressRange: %s", hipGetErrorString(e));
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, base);
    if (e != hipSuccess) return fail(KGE_ERR_LAUNCH, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
    memcpy(handle64, &h, sizeof(h));
    *offset = (int64_t)((const char *)dev_ptr - (const char *)base);
    return KGE_OK;
}

int kge_ipc_open(const void *handle64, void **base) {
    if (!handle64 || !base) return fail(KGE_ERR_ARG, "kge_ipc_open: null argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    hipError_t e = hipIpcOpenMemHandle(base, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(KGE_ERR_LAUNCH, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    return KGE_OK;
}

int kge_ipc_close(void *base) {
    if (!base) return KGE_OK;
    hipError_t e = hipIpcCloseMemHandle(base);
    if (e != hipSuccess) return fail(KGE_ERR_LAUNCH, "hipIpcCloseMemHandle: %s", hipGetErrorString(e));
    return KGE_OK;
}

}  // extern "C"
