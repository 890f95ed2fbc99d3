/*
 * kge_hip.h - C ABI of libkge_hip.so, the MI355X (gfx950) implementation of the DGL-KE training
 * hot path: gather -> positive score -> chunked negative score -> loss -> analytic gradients ->
 * row-sparse Adagrad.
 *
 * The reference (awslabs/dgl-ke) has no FFI boundary for this path: it is Python calling torch
 * ops.  Each entry point below replaces one reference function (cited as file:line relative to
 * python/dglke/) and is what a maintainer would bind from those call sites (see INTEGRATION.md
 * for the ctypes stubs).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (a torch tensor); fp32 data, int64 ids
 *    (the reference's dtypes), int32 for the batch-plan CSR arrays this library defines;
 *  - no allocation, no host synchronisation and no global mutable state inside; every call takes
 *    an explicit stream (hipStream_t passed as void*) and only enqueues kernels on it, so calls
 *    can be captured into a hipGraph (the one object the library creates is the kge_pipe of the
 *    --async_update pipeline: a small host-side struct, explicitly created / destroyed by the caller);
 *  - return value 0 = success, negative = kge_status; kge_last_error() returns the message of
 *    the last failure on the calling thread;
 *  - embedding rows of ComplEx / RotatE are [re | im] halves (models/pytorch/score_fun.py:298-300).
 */
#ifndef KGE_HIP_H
#define KGE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KGE_ABI_VERSION 8

/* score functions (models/general_models.py:248-268 model_name strings) */
enum kge_model {
    KGE_TRANSE_L1 = 0, /* score_fun.py:40  TransEScore(dist_func='l1') */
    KGE_TRANSE_L2 = 1, /* score_fun.py:40  TransEScore(dist_func='l2') */
    KGE_DISTMULT  = 2, /* score_fun.py:222 DistMultScore */
    KGE_COMPLEX   = 3, /* score_fun.py:289 ComplExScore */
    KGE_ROTATE    = 4, /* score_fun.py:451 RotatEScore */
    KGE_TRANSR    = 7, /* score_fun.py:110 TransRScore: third table of per-relation projection matrices
                          [d_e x d_r] (kge_tables.proj); fused step and ranking only (no modular autograd ops) */
    KGE_RESCAL    = 6, /* score_fun.py:378 RESCALScore: relation row = [d_e x d_e] matrix M (d_r = d_e*d_e),
                          p = h.(M t); BOTH corruption modes use the pos-side vector M x (:428-447) */
    KGE_SIMPLE    = 5  /* score_fun.py:556 SimplEScore: rows = [x_i | x_j] halves, relation = [r | r_inv];
                          scores are clamped to [-20, 20] like the reference (:568, :622, :641) */
};

/* loss criteria (models/pytorch/loss.py:44-59) */
enum kge_loss {
    KGE_LOSS_LOGSIGMOID = 0,
    KGE_LOSS_LOGISTIC   = 1,
    KGE_LOSS_HINGE      = 2,
    KGE_LOSS_BCE        = 3
};

enum kge_status {
    KGE_OK            = 0,
    KGE_ERR_ARG       = -1, /* bad argument (shape / model / null pointer) */
    KGE_ERR_WORKSPACE = -2, /* workspace too small */
    KGE_ERR_LAUNCH    = -3  /* HIP launch error */
};

/* flags for kge_score_neg_* / kge_step_* : force the VALU pairwise kernels instead of the MFMA
 * GEMM kernels for the models that have a GEMM form (validation aid; TransE_l1 / RotatE always
 * use the pairwise kernels). */
#define KGE_FLAG_FORCE_PAIRWISE 1u
/* keep the generic edge-gradient kernel: TransE - instead of rebuilding the per-edge gradients inside the
 * update kernel; DistMult - instead of writing them from the epilogue of the backward GEMM's GA tiles
 * (validation aid) */
#define KGE_FLAG_NO_TRANSE_FAST 2u
/* matrix-core path: read the negative rows from a dense per-step copy instead of gathering them
 * from the entity table through neg_ids (tuning aid) */
#define KGE_FLAG_DENSE_NEG 4u
/* matrix-core path, pointwise criteria, N <= 256: no stand-alone loss kernel - the forward tiles emit the
 * factorised loss gradient u_ij + per-(row, 16-column tile) softmax partials and the backward wavefronts
 * combine them (4 launches per TransE step instead of 5; same results within rounding).  Opt-in: on MI355X
 * the work it moves into the two GEMM kernels costs what the loss launch cost
 * (profiles/r02_fused_loss_experiment.txt). */
#define KGE_FLAG_FUSED_LOSS 8u
/* TransE_l1 / RotatE: keep the two-pass pairwise backward (GA and GN evaluated separately) instead
 * of the kernel that evaluates every (positive, negative) pair once for both products (validation aid) */
#define KGE_FLAG_TWO_PASS_PAIR 16u
/* kge_step_fused / kge_step_sharded: --neg_deg_sample of the reference (general_models.py:396-402,
 * 424-432).  The corrupted-side entities of a chunk's OWN positives are prepended to the chunk's
 * negatives: every positive is scored against N' = chunk + N rows, the chunk x chunk diagonal (the
 * positive edge itself) is masked to score 0 (it stays in the loss, its gradient vanishes) and the
 * gradients of the in-batch rows join the POSITIVE trace of their entity.  kge_batch is unchanged
 * (N and the plan describe the SAMPLED negatives); kge_step_workspace_bytes accounts for N';
 * kge_step_out.neg_score is [B, N'], g_neg [C*N', d_e] (rows c*N' + chunk.. are the sampled ones).
 * Not available for RESCAL / TransR.  kge_step_grads (ABI 6): the in-batch rows' gradients are part of the positive-trace
 * message g0 of their entity, the sampled rows' of g1 - the traces the reference pushes. */
#define KGE_FLAG_NEG_DEG_SAMPLE 32u
/* kge_step_async: defer the relation-table update by one step as well (the reference's --async_update defers the
 * entity table only, general_models.py:639-647; with this flag no update at all sits between two steps' scoring) */
#define KGE_FLAG_ASYNC_REL 64u

/* strict step, TransE_l2 / DistMult on the matrix-core path: keep edge-forward and the forward GEMM as two launches
 * (5 launches per TransE_l2 step) instead of the merged first launch of round 3 whose tiles build the pos-side
 * fragments from the table rows themselves (4 launches).  Validation / A-B aid: same results within rounding. */
#define KGE_FLAG_SPLIT_FWD 128u

/* merged first launch: the edge-forward half also writes a dense copy of the negative rows and the backward GEMM reads it
 * instead of gathering the rows from the entity table through neg_ids (tuning / A-B aid: same speed, +1.6 MB of writes) */
#define KGE_FLAG_DENSE_BWD 256u

/* merged first launch: forward tiles with direct fragment loads of x, r and b (three gathered loads per k-step) instead of the
 * pos-side tile built once per workgroup in LDS (tuning / A-B aid) */
#define KGE_FLAG_FWD_DIRECT 512u

/* strict step, TransE_l2 / DistMult / ComplEx on the merged first launch, kge_step_out.tickets given: run LossGenerator INSIDE
 * that launch (3 launches per step): the forward tiles store final scores write-through, the workgroups of a 16-row strip draw
 * an arrival ticket and the last one runs the strip's loss rows.  Opt-in: measured SLOWER on MI355X (34.6 vs 30.6 us per cfg-T
 * step, profiles/r04_loss_fold.txt - the last arrivers' four wavefronts run 16 loss rows where the loss launch runs one row per
 * wavefront, and drain + ticket + re-read cost what the launch boundary costs).  DistMult / ComplEx results are bit-identical to
 * the 4-launch step, TransE_l2 equal within the rounding of |a|^2, |b|^2 (summed by the tiles instead of by edge_fwd). */
#define KGE_FLAG_LOSS_IN_FWD 1024u

int         kge_abi_version(void);
const char *kge_last_error(void);

/* ---- A2: ExternalEmbedding.__call__  (models/pytorch/tensor_models.py:270-302, line 292) ----
 * out[k,:] = table[idx[k],:] */
int kge_gather_rows(const float *table, int64_t n_rows, int dim, const int64_t *idx,
                    int64_t n_idx, float *out, void *stream);

/* ---- A3: score_func.edge_func  (score_fun.py:54-59, 229-235, 297-307, 460-472) ----
 * out[i] = positive score of (h[i], r[i], t[i]); h,t: [B,d_e]  r: [B,d_r] dense rows. */
int kge_score_pos(int model, const float *h, const float *r, const float *t, int64_t B, int d_e,
                  int d_r, float gamma, float emb_init, float *out, void *stream);
/* autograd of edge_func: (gh, gr, gt) = dpos[i] * d score/d (h, r, t) */
int kge_score_pos_bwd(int model, const float *h, const float *r, const float *t,
                      const float *dpos, int64_t B, int d_e, int d_r, float gamma,
                      float emb_init, float *gh, float *gr, float *gt, void *stream);

/* ---- A4/A5: score_func.create_neg(neg_head) closures (score_fun.py:91-108, 268-286, 345-376,
 * 512-554) called from KEModel.predict_neg_score (general_models.py:405, 426) ----
 * pos_side: the uncorrupted entity rows [C*chunk, d_e] (tails if neg_head else heads);
 * rel: [C*chunk, d_r]; neg: corrupt entity rows [C*N, d_e]; out: [C, chunk, N].
 * ws: scratch of at least kge_score_neg_workspace_bytes(). */
size_t kge_score_neg_workspace_bytes(int model, int C, int chunk, int N, int d_e);
int kge_score_neg_fwd(int model, int neg_head, const float *pos_side, const float *rel,
                      const float *neg, int C, int chunk, int N, int d_e, int d_r, float gamma,
                      float emb_init, float *out, void *ws, size_t ws_bytes, unsigned flags,
                      void *stream);
/* autograd of the closure: given dneg [C,chunk,N] -> g_pos_side [C*chunk,d_e], g_rel
 * [C*chunk,d_r], g_neg [C*N,d_e]. */
int kge_score_neg_bwd(int model, int neg_head, const float *pos_side, const float *rel,
                      const float *neg, const float *neg_score, const float *dneg, int C,
                      int chunk, int N, int d_e, int d_r, float gamma, float emb_init,
                      float *g_pos_side, float *g_rel, float *g_neg, void *ws, size_t ws_bytes,
                      unsigned flags, void *stream);

/* ---- A6: LossGenerator.get_total_loss (models/pytorch/loss.py:69-98) + its gradient ----
 * pos [B], neg [B,N], w = edge importance [B] or NULL.  loss3 = {pos_loss, neg_loss, loss}
 * (pairwise: {nan, nan, loss}); dpos [B], dneg [B,N] = d loss / d score.
 * ws: scratch of >= 2*B floats. */
int kge_loss_fwd_bwd(int loss_genre, int adv, float adv_temp, int pairwise, float margin,
                     const float *pos, const float *neg, const float *w, int64_t B, int N,
                     float *loss3, float *dpos, float *dneg, void *ws, size_t ws_bytes,
                     void *stream);

/* ---- A9: ExternalEmbedding.update for ONE trace (tensor_models.py:304-362) ----
 * state[idx[k]] += mean_d(grad[k,d]^2) for all k (duplicates accumulate, :352), then
 * table[idx[k],:] += -lr*grad[k,:]/(sqrt(state[idx[k]])+eps) using the state AFTER all adds
 * (:353-361).  Lock-free (float atomics), two kernels. */
int kge_adagrad_scatter(float *table, float *state_sum, int64_t n_rows, int dim,
                        const int64_t *idx, const float *grad, int64_t n_idx, float lr, float eps,
                        void *stream);

/* ---- small ops of the drop-in route (KEModel.forward / forward_test with per-op autograd) ----
 * kge_scatter_add_rows: out[idx[k],:] += src[k,:] - autograd of the LOCAL-id gathers pos_g.ndata['emb'][head_ids]
 *   (general_models.py:384-388, 410-414), i.e. index_add; float atomics, out is accumulated into (caller zeroes).
 * kge_pnorm_pow(_bwd): x.norm(p)**p of a [n, dim] block (tensor_models.py:54, used by the regulariser
 *   general_models.py:572-576) - deterministic two-stage sum, ws = n floats - and its gradient
 *   gx = gout[0] * p * |x|^(p-1) * sign(x).
 * kge_mask_diag: x[c,i,i] = 0 in place on [C, chunk, Np] - the --neg_deg_sample mask (general_models.py:401-402).
 * kge_rank_from_scores: ranks[i] = 1 + #{j: neg[i,j] >= pos[i] and bias[i,j] != -1} (bias NULL: no filter) -
 *   the ranking of KEModel.forward_test (general_models.py:463-478). */
int kge_scatter_add_rows(float *out, int64_t n_rows, int dim, const int64_t *idx, const float *src, int64_t n_idx,
                         void *stream);
int kge_pnorm_pow(const float *x, int64_t n, int dim, int p, float *out, void *ws, size_t ws_bytes, void *stream);
int kge_pnorm_pow_bwd(const float *x, int64_t n, int dim, int p, const float *gout, float *gx, void *stream);
int kge_mask_diag(float *x, int C, int chunk, int Np, void *stream);
int kge_rank_from_scores(const float *neg, const float *pos, const float *bias, int64_t E, int64_t N, int64_t *ranks,
                         void *stream);

/* ---- fused step: KEModel.forward + loss.backward() + KEModel.update
 * (train_pytorch.py:141-152; general_models.py:529-588) for one batch ----
 *
 * The batch is the content of the reference's (pos_g, neg_g) pair (general_models.py:376-427,
 * dataloader/sampler.py:421-457) as flat id arrays, plus a "plan": the grouping of duplicate rows
 * that lets every table row be updated by exactly one wavefront, deterministically and without
 * atomics, with the reference's trace order (entity pos-unique trace, then entity negative trace,
 * then relation trace; tensor_models.py:316). */
typedef struct kge_batch {
    int32_t B;      /* positive edges                                                       */
    int32_t C;      /* neg_g.num_chunks                                                     */
    int32_t chunk;  /* neg_g.chunk_size  (C*chunk == B)                                     */
    int32_t N;      /* neg_g.neg_sample_size                                                */
    int32_t neg_head;
    int32_t U;      /* number of unique positive entities (len(pos_g.ndata['id']))          */
    int32_t UE;     /* unique entities in (pos nodes U negative nodes)                      */
    int32_t UR;     /* unique relations                                                     */
    const int64_t *h_gid;   /* [B]   global head entity id  = nid[h_local]                   */
    const int64_t *t_gid;   /* [B]   global tail entity id                                   */
    const int64_t *rel_ids; /* [B]   pos_g.edata['id']                                       */
    const int64_t *neg_ids; /* [C*N] neg_g.ndata['id'][head_nid|tail_nid]                    */
    const float   *edge_w;  /* [B] pos_g.edata['impts'] or NULL                              */
    /* plan */
    const int64_t *ue_id;       /* [UE]   global entity id of union entry u                  */
    const int32_t *ue_pos_ptr;  /* [UE+1] CSR into ue_pos_adj (empty list: not a pos node)   */
    const int32_t *ue_pos_adj;  /* [2B]   edge*2 + side (0 = head end, 1 = tail end)         */
    const int32_t *ue_neg_ptr;  /* [UE+1] CSR into ue_neg_slot                               */
    const int32_t *ue_neg_slot; /* [C*N]  positions in neg_ids holding this entity           */
    const int64_t *ur_id;       /* [UR]   unique relation id                                 */
    const int32_t *ur_ptr;      /* [UR+1] CSR into ur_edge                                   */
    const int32_t *ur_edge;     /* [B]    edges carrying this relation                       */
    /* packed records, 32-byte aligned, one per unique row: everything the update kernel needs  */
    /* to request the row and its first gradient rows in ONE dependent round                    */
    const int32_t *ue_rec;      /* [UE][8] {id_lo, id_hi, pos_begin, pos_end, neg_begin,     */
                                /*          neg_end, ue_pos_adj[pos_begin]|-1, ue_neg_slot[neg_begin]|-1} */
    const int32_t *ur_rec;      /* [UR][8] {id_lo, id_hi, edge_begin, edge_end, ur_edge[edge_begin], 0,0,0} */
    /* batches built ON THE DEVICE (kge_sample_batches): UE / UR above are upper bounds and the  */
    /* kernels read the actual counts from this device array of FOUR int32 {UE, UR, corrupt-head */
    /* flag of the step, edges of the batch's most frequent relation}; NULL for host-built plans */
    const int32_t *counts_dev;
    /* ABI 8: mean of edge_w[0..B) in fp32, computed ONCE per batch by whoever builds it (dglke_amd/plan.py).  The reference weights */
    /* every positive edge by the batch's MEAN importance (loss.py:75,82: a [B] * [B,1] broadcast); <= 0 (or edge_w NULL): the     */
    /* kernels sum the weights themselves, every row's wavefront for itself (what ABI <= 7 always did).                             */
    float edge_w_mean;
    int32_t reserved_;
} kge_batch;

typedef struct kge_hparams {
    int32_t model;       /* enum kge_model                                                   */
    int32_t d_e, d_r;    /* entity / relation row widths in floats                           */
    int32_t loss_genre;  /* enum kge_loss                                                    */
    int32_t adv;         /* args.neg_adversarial_sampling                                    */
    int32_t pairwise;
    int32_t reg_norm;    /* args.regularization_norm                                         */
    uint32_t flags;
    float gamma, emb_init, lr, adv_temp, margin, reg_coef, eps /* 1e-10 */;
} kge_hparams;

typedef struct kge_tables {
    float *ent;        /* [n_ent, d_e] entity_emb.emb                                        */
    float *ent_state;  /* [n_ent]      entity_emb.state_sum                                  */
    float *rel;        /* [n_rel, d_r]                                                       */
    float *rel_state;  /* [n_rel]                                                            */
    int64_t n_ent, n_rel;
    float *proj;       /* TransR only: [n_rel, d_e*d_r] projection_emb.emb (score_fun.py:114-118), else NULL */
    float *proj_state; /* TransR only: [n_rel] projection_emb.state_sum                      */
} kge_tables;

/* optional outputs of a step (any may be NULL) */
typedef struct kge_step_out {
    float *loss4;      /* {pos_loss, neg_loss, loss (without reg), regularization} of THIS   */
                       /* step (costs one extra single-block reduction kernel)              */
    float *loss_accum; /* [4][KGE_ACC_SLOTS] running sums over steps of the same four        */
                       /* quantities, spread over slots so that no per-step reduction is     */
                       /* needed; reduce with kge_reduce_loss() when the log is wanted       */
                       /* (replaces the 3-4 .item() host syncs per step of loss.py:95-97)    */
    float *pos_score;  /* [B]                                                                */
    float *neg_score;  /* [C,chunk,N] scores (copied before they are overwritten)            */
    float *g_pos_ent;  /* [UE, d_e] trace-0 gradient per union entry (0 if not a pos node)   */
    float *g_neg;      /* [C*N, d_e]  trace-1 gradient                                       */
    float *g_rel;      /* [B, d_r]    relation-trace gradient                                */
    int32_t *tickets;  /* [KGE_TICKET_INTS] hand-off words of KGE_FLAG_LOSS_IN_FWD (ABI 6): ZERO when    */
                       /* first handed over, returned to zero by every step; one array per stream of */
                       /* steps (concurrent steps must not share it).  NULL: the flag is ignored.    */
                       /* A step that finds a word outside [0, workgroups per strip) traps           */
                       /* (hipErrorLaunchFailure at the next synchronisation).                        */
} kge_step_out;

#define KGE_ACC_SLOTS 4096
#define KGE_TICKET_INTS 4096
/* out4 = per-quantity sums of the running-sum slots (divide by the number of steps for the
 * averages the reference prints, train_pytorch.py:165-167); optionally zero the slots. */
int kge_reduce_loss(float *loss_accum, float *out4, int zero_after, void *stream);

size_t kge_step_workspace_bytes(const kge_hparams *hp, int B, int C, int chunk, int N, int UE,
                                int UR);
/* forward + backward + update of one batch, enqueued on `stream`. */
int kge_step_fused(const kge_hparams *hp, const kge_tables *tb, const kge_batch *b,
                   const kge_step_out *out, void *ws, size_t ws_bytes, void *stream);

/* ---- --async_update (models/pytorch/tensor_models.py:136-175 async_update, :364-375 create/finish_async_update;
 * models/general_models.py:639-647; train_pytorch.py:120-121, 194-195; docs/source/train.rst:217-259) ----
 * The reference hands the entity gradients of step s to a helper PROCESS and starts step s+1 while they are
 * applied: the gather of step s+1 may miss the update of step s (<= 1 step of staleness), racily.  Here the
 * staleness is exact and race-free, and the "helper" is the other half of a horizontally fused launch:
 *
 *     [ forward(s) || UPDATE(s-1) ]  ->  loss(s)  ->  [ backward(s) || PREP(s+1) ]  ->  [ forward(s+1) || UPDATE(s) ] ...
 *
 *   PREP(s)   = gather + positive scores + pos-side vectors + DENSE COPIES of every row the step reads again
 *               (negative rows; h / t / r rows for the per-edge gradient kernel and the regulariser);
 *   forward / loss / backward(s) read only PREP's copies;
 *   UPDATE(s-1) = row-sparse Adagrad with the gradients of step s-1, applied to the rows as they are then, in the
 *               SAME launch as the forward matrix-core tiles of step s (different workspace half).
 * UPDATE(s-1) is launched after PREP(s) and completes before PREP(s+1): step s+1 gathers entity rows that contain
 * every update up to s-1 and not the update of s - bit-reproducible, one stream, no events.
 * By default only the ENTITY table is deferred, like the reference (general_models.py:639-647 defers entity_emb
 * only): the relation trace of step s is applied right after backward(s) and PREP(s+1) is its own launch.
 * KGE_FLAG_ASYNC_REL defers the relation trace too; then nothing touches the tables between backward(s) and
 * PREP(s+1), and - when the caller names the next batch (b_next) - PREP(s+1) shares the backward launch of step s
 * (three launches per step on the critical path).  b_next may be NULL (PREP runs at the start of the next call).
 * The PREP that ran ahead is used by the next call only if that call passes the same batch arrays (b->h_gid), the same
 * hyper-parameters and asks for no per-step outputs; the arrays of b_next must not be rebuilt in between (a sampler slot
 * is re-sampled only after kge_step_async_flush).
 * kge_step_async_flush() applies the last pending update (call it before reading the tables and at the end of a
 * captured group of steps; the step after a flush gathers fully updated rows).  Gradients, including the
 * regulariser, are those of the rows as gathered.  Not available for RESCAL / TransR.  The workspace holds two halves (kge_step_async_workspace_bytes).  kge_pipe is host-side state only. */
typedef struct kge_pipe kge_pipe;
int kge_pipe_create(kge_pipe **pipe);
int kge_pipe_destroy(kge_pipe *pipe);
size_t kge_step_async_workspace_bytes(const kge_hparams *hp, int B, int C, int chunk, int N, int UE, int UR);
int kge_step_async(kge_pipe *pipe, const kge_hparams *hp, const kge_tables *tb, const kge_batch *b,
                   const kge_batch *b_next, const kge_step_out *out, void *ws, size_t ws_bytes, void *stream);
int kge_step_async_flush(kge_pipe *pipe, void *stream);

/* ---- the strict step in four pieces (the reference's per-phase timers, train_pytorch.py:127-177: sample / forward /
 * backward / update): calling the four phase groups in this order on one stream IS kge_step_fused - same kernels,
 * same workspace; the caller puts events between them.  TransR's forward projections sit in GATHER. */
#define KGE_PHASE_GATHER   1   /* ids -> rows, positive scores, pos-side vectors, positive-loss part */
#define KGE_PHASE_FORWARD  2   /* chunked negative scores + loss */
#define KGE_PHASE_BACKWARD 4   /* gradients w.r.t. the pos-side vectors, the negative rows and the per-edge rows */
#define KGE_PHASE_UPDATE   8   /* row-sparse Adagrad on both tables */
int kge_step_phase(const kge_hparams *hp, const kge_tables *tb, const kge_batch *b, const kge_step_out *out,
                   void *ws, size_t ws_bytes, int phases, void *stream);

/* ---- range-sharded training (one process per GPU) ----
 * kge_step_grads: same as kge_step_fused but instead of updating the entity table it EMITS, per
 * union entry u, the two trace gradients and their Adagrad increments so that the owner rank can
 * apply them: g0[u,:] (trace 0), gs0[u] = mean(g0^2); g1[u,:] = sum over duplicates (trace 1),
 * gs1[u] = sum_k mean(g_k^2).  The relation table is still updated locally unless rel_emit. */
typedef struct kge_emit {
    float *g0;  float *gs0;  /* [UE,d_e], [UE] */
    float *g1;  float *gs1;  /* [UE,d_e], [UE] */
    float *gr;  float *gsr;  /* [UR,d_r], [UR] summed relation gradient (NULL: update locally) */
    /* row strides in floats (0 = dense: d_e / 1 / d_r / 1).  With strides the six arrays can be
     * interleaved into ONE message per row - e.g. [g0 | g1 | gs0 gs1 pad pad] with ld_e = 2*d_e+4 -
     * so that a single all-to-all pushes both traces (see kge_adagrad_apply_packed). */
    int32_t ld_e;            /* stride of g0, g1, gs0, gs1 */
    int32_t ld_r;            /* stride of gr, gsr */
    int32_t *rid;            /* optional [UR*ld_r] view: relation id written as two int32 words at */
                             /* rid[u*ld_r], rid[u*ld_r+1] (lo, hi) next to the gradient, or NULL;  */
                             /* on device-built plans rows u in [unique relations, UR) get id -1    */
    int32_t ent_by_id;       /* != 0: the entity messages of union entry u are written at row       */
                             /* ue_id[u] instead of row u (batches re-addressed to cache rows by     */
                             /* kge_route_build: the message array is then laid out like the cache) */
    int32_t reserved;
    /* ABI 8 - PACKED single-trace entity messages (the push all-to-all moves one trace per row instead of two, of which one was   */
    /* almost always empty): msg_rows != NULL -> union entry u writes ONE message [g (d_e) | gs | link | . .] (ld_e = d_e + 4) at   */
    /* row msg_rows[2u] of g0 - its positive trace if it has one, else its negative trace - and, only when the row is in BOTH      */
    /* traces (msg_rows[2u+1] = p >= 0), the negative trace as a second message at position p of the same owner bucket's extra     */
    /* region (`link` = p in the first one's header, else -1).  msg_rows comes from kge_route_build (ue_msg); g1 / gs* are unused.  */
    const int32_t *msg_rows;
    int32_t msg_cap, msg_cap_extra;   /* bucket geometry of the packed messages: cap + cap_extra rows per owner, extra region from cap */
} kge_emit;
int kge_step_grads(const kge_hparams *hp, const kge_tables *tb, const kge_batch *b,
                   const kge_step_out *out, const kge_emit *emit, void *ws, size_t ws_bytes,
                   void *stream);
/* owner side: for k < n: s = state[idx[k]] + gs[k]; state = s; table[idx[k],:] += -lr*g[k,:] /
 * (sqrt(s)+eps).  idx must be unique within one call (one wavefront owns one row). A row with
 * gs[k] == 0 and an all-zero gradient is left untouched. */
int kge_adagrad_apply_rows(float *table, float *state_sum, int64_t n_rows, int dim,
                           const int64_t *idx, const float *g, const float *gs, int64_t n,
                           float lr, float eps, void *stream);

/* ---- ranking evaluation: KEModel.forward_test (models/general_models.py:436-485) over the
 * batches an EvalSampler would produce (dataloader/sampler.py:514-597; test loop
 * train_pytorch.py:199-253), for E test triples at once ----
 * For triple i the corrupted side (head if neg_head else tail) is replaced by every candidate
 * (cand == NULL: all n_ent entities, column j = entity j; else the n_cand ids in cand) and
 *   ranks[i] = 1 + #{j : score(i,j) >= score(i)} - #{j in filt_i : score(i,j) >= score(i)}
 * filt_i = filt_ids[filt_ptr[2i] .. filt_ptr[2i+1]) are candidate COLUMNS whose corrupted triple
 * exists in the graph (the reference's bias == -1 mask, :463-475; unique within a list; filt_ptr
 * is [E][2] begin/end so that triples with the same (h,r) or (r,t) share one list; NULL =
 * --no_eval_filter, where the true triple itself counts exactly as in the reference).  One pass per
 * Eb triples.  Matrix-form models (TransE_l2, DistMult, ComplEx, SimplE, RESCAL): a tiled fp32-MFMA
 * GEMM whose epilogue keeps the comparison bit of every (triple, candidate) pair, ranks from the
 * bit mask (csrc/kge_rank_gemm.hip); KGE_FLAG_FORCE_PAIRWISE, the pairwise models and TransR: the
 * training kernels' chunked negative scores as a block + a counting kernel.
 * pos_score_out (optional) receives the E true-triple scores. */
size_t kge_rank_workspace_bytes(int Eb, int64_t n_cand, int d_e);
int kge_rank_eval(int model, int neg_head, const float *ent, int64_t n_ent, const float *rel,
                  int64_t n_rel, const int64_t *h, const int64_t *r, const int64_t *t, int64_t E,
                  int d_e, int d_r, float gamma, float emb_init, const int64_t *cand, int64_t n_cand,
                  const int64_t *filt_ptr, const int64_t *filt_ids, int Eb, int32_t *ranks,
                  float *pos_score_out, void *ws, size_t ws_bytes, unsigned flags, void *stream);

/* same with the TransR projection table (proj = NULL for every other model); for TransR `cand` must be given
 * (pass the identity list for "all entities") */
int kge_rank_eval_ex(int model, int neg_head, const float *ent, int64_t n_ent, const float *rel,
                     int64_t n_rel, const float *proj, const int64_t *h, const int64_t *r, const int64_t *t,
                     int64_t E, int d_e, int d_r, float gamma, float emb_init, const int64_t *cand,
                     int64_t n_cand, const int64_t *filt_ptr, const int64_t *filt_ids, int Eb, int32_t *ranks,
                     float *pos_score_out, void *ws, size_t ws_bytes, unsigned flags, void *stream);

/* ---- peer-to-peer sharded step (xGMI direct; the Hogwild multi-GPU mode) ----
 * The reference's multi-GPU trainer keeps ONE entity table in shared host memory and lets every
 * trainer process gather from it and update it without locks (train.py:298-317 --num_proc,
 * tensor_models.py:233 share_memory, :304-362 update; relation table likewise unless
 * --rel_part).  Here the shared table is the union of the 8 GPUs' HBM: rank k owns the rows
 * [k*rows_per_shard, (k+1)*rows_per_shard) of both tables (+ Adagrad state), every rank maps all
 * peers' shards into its address space (kge_ipc_export / kge_ipc_open = hipIpc handles) and
 * kge_step_sharded runs the SAME kernels as kge_step_fused with row addresses resolved through
 * the shard map: remote rows are read and read-modify-written directly over xGMI, no collective
 * and no host work in the step.  Within a rank the update is still owner-computes and
 * deterministic; across ranks it is Hogwild, exactly like the reference's shared table.
 * The four pointer arrays are DEVICE arrays [n_shards] of bases valid in the calling process. */
typedef struct kge_shards {
    int32_t n_shards, reserved;
    int64_t ent_rows_per_shard, rel_rows_per_shard;
    float *const *ent_rows;  float *const *ent_state;
    float *const *rel_rows;  float *const *rel_state;
    int64_t n_ent, n_rel;    /* global row counts (ids in the batch are global) */
    /* ABI 8 - TransR / RESCAL on sharded ENTITY tables (the reference trains TransR on 8 GPUs with --rel_part,                    */
    /* examples/freebase/multi_gpu.sh:80-89: every trainer holds the rows of ITS relations on its own GPU,                         */
    /* general_models.py:590-637).  rel_local != NULL: the relation-side tables are LOCAL to the calling rank - [n_rel, d_r] rows   */
    /* (RESCAL: d_e x d_e matrices) + state, and for TransR the projection table [n_rel, d_e * d_r] + state                        */
    /* (score_fun.py:114-118) - and rel_rows / rel_state are ignored.  The step gathers the batch's entity rows through the shard   */
    /* map into dense copies, runs the TransR / RESCAL kernels on those, updates relation-side rows in place and entity rows        */
    /* through the shard map.  Required for KGE_TRANSR / KGE_RESCAL, optional (relation table replicated per rank) otherwise.       */
    float *rel_local, *rel_state_local;
    float *proj_local, *proj_state_local;
} kge_shards;
int kge_step_sharded(const kge_hparams *hp, const kge_shards *sh, const kge_batch *b,
                     const kge_step_out *out, void *ws, size_t ws_bytes, void *stream);
/* rows of a sharded table by global id (evaluation / checkpoint helper): out[k,:] = row idx[k] */
int kge_gather_rows_sharded(float *const *shard_rows, int n_shards, int64_t rows_per_shard, int dim,
                            const int64_t *idx, int64_t n_idx, float *out, void *stream);
/* hipIpc plumbing for the shard map.  export: handle64 = 64-byte hipIpcMemHandle_t of the
 * ALLOCATION that contains dev_ptr, *offset = dev_ptr - allocation base.  open: maps a peer's
 * allocation into this process (current device) and returns its base; close unmaps it. */
#define KGE_IPC_HANDLE_BYTES 64
int kge_ipc_export(const void *dev_ptr, void *handle64, int64_t *offset);
int kge_ipc_open(const void *handle64, void **base);
int kge_ipc_close(void *base);

/* ---- on-device sampler + plan builder (replaces the DGL EdgeSampler wrappers of
 * dataloader/sampler.py:376-419, 823-876 and the host plan of dglke_amd/plan.py) ----
 * heads/rels/tails: the training triples in HBM; perm: base edge permutation or NULL (sequential order);
 * state: device int64[4] = {(legacy) position, step number (1-based), ticket counter (0 between launches), unused}, advanced
 * by the launch itself (its last workgroup); builds n_slots
 * consecutive batches (batch k = step state[1]+k: odd steps corrupt tails, even steps heads;
 * C*N uniform negatives with replacement from a counter-based RNG keyed by (seed, step)).  Epochs consist of
 * floor(n_train / B) WHOLE batches (the trailing partial batch is dropped, dataloader/sampler.py:503-504) and
 * every epoch has its own edge order (shuffle=True): epoch 0 = perm, epoch e > 0 = perm composed with an affine
 * bijection of [0, n_train) keyed by (seed, e).  n_train >= B.
 * Limits: 2B + C*N <= 8192 (one workgroup sorts a batch in LDS: 4 keys per thread up to 4096 elements, the
 * wide instance 8 keys per thread - the reference's batch-2048 recipes are 6144). */
size_t kge_sampler_slot_bytes(int B, int C, int N);
int kge_sample_batches(const int64_t *heads, const int64_t *rels, const int64_t *tails,
                       const int64_t *perm, int64_t n_train, int64_t n_ent, int B, int C, int chunk,
                       int N, uint64_t seed, int64_t *state, void *slots, size_t slot_bytes,
                       int n_slots, void *stream);
/* ---- ABI 7 (round 5): the sampler as TAIL WORKGROUPS of the training step's own launches (csrc/kge_sampler_tail.hpp) ----
 * kge_step_fused_sampling = kge_step_fused of batch `b` whose first, backward and update launches each carry a few extra
 * 256-thread workgroups that build ONE other batch - the one kge_sample_batches would build as batch `k` of its next launch
 * (step state[1] + k: same ids from the same counter RNG / epoch permutation, same plan, bit for bit) - into `slot`, a phase per
 * launch (ids + bucketed keys | per-bucket sort + scans | relocation to the final lists and records - that last phase on the NEXT
 * step's first launch, `prev_slot`; the group's last job: on its own update launch).  Batch k is complete one launch into the
 * step that carries job k + 1 (the last one: when its step ends); nothing is launched for it.
 * The reference's counterpart: the EdgeSampler worker threads that build the next batches while a step runs
 * (dataloader/sampler.py:823-876).  `state` is NOT advanced by a job unless `advance` > 0 (the group's last job: advance by that
 * many batches, after every job of the group has read it).  `scratch`: kge_sampler_tail_scratch_bytes(), one buffer shared by
 * all jobs of a stream.  Steps whose launches cannot carry the tail (everything but the 4-launch strict step of TransE_l2 /
 * DistMult / ComplEx on local tables) return KGE_ERR_ARG: use kge_sample_batches then.  job == NULL: plain kge_step_fused. */
typedef struct kge_sampler_job {
    const int64_t *heads, *rels, *tails, *perm;   /* as kge_sample_batches */
    int64_t  n_train, n_ent;
    int      B, C, chunk, N;
    uint64_t seed;
    int64_t *state;             /* int64[8]: {position, step (1-based), ticket, -} as kge_sample_batches uses them + {epoch, mul, add, reciprocal}: the epoch constants the jobs cache (initialise word 4 to -1) */
    void    *slot;              /* output slot of THIS batch (kge_sampler_slot_bytes) */
    void    *scratch; size_t scratch_bytes;
    int      k;                 /* index of the batch within the group under construction */
    int      advance;           /* > 0: last job of the group - advance `state` by this many batches */
    int      pre_permuted;      /* != 0: heads / rels / tails are already in base-permutation order (x[perm[i]] at i): the job skips
                                 * the perm lookup - one dependent memory round less in its first phase; perm != NULL still means
                                 * "a new edge order every epoch" */
    int      reserved;
    void    *prev_slot;         /* slot of the job the PREVIOUS step carried (k - 1), whose last phase rides on this step's first launch; NULL for k = 0.
                                 * The group's last job (advance > 0) finishes under its own step's update launch. */
} kge_sampler_job;
size_t kge_sampler_tail_scratch_bytes(int B, int C, int N, int64_t n_ent);
int kge_step_fused_sampling(const kge_hparams *hp, const kge_tables *tb, const kge_batch *b, const kge_step_out *out,
                            void *ws, size_t ws_bytes, const kge_sampler_job *job, void *stream);

/* host-side: point a kge_batch at slot `slot` (pure pointer arithmetic, no device access) */
int kge_batch_from_slot(void *slots, size_t slot_bytes, int slot, int B, int C, int chunk, int N,
                        int neg_head, kge_batch *out);

/* owner side, packed messages (one row per message, `ld` floats apart):
 *   msg = [ g_0 (dim) | ... | g_{T-1} (dim) | gs_0 .. gs_{T-1} | (id_lo id_hi as int32 bits when idx == NULL) ]
 * applies the T traces of every row in order (trace t: s += gs_t; row += -lr*g_t/(sqrt(s)+eps)).
 * Row ids come from idx[k] or, when idx is NULL, from the two int32 words stored after the gs
 * values (a negative id skips the row).  Ids must be unique within one call. */
int kge_adagrad_apply_packed(float *table, float *state_sum, int64_t n_rows, int dim,
                             const int64_t *idx, const float *msg, int ld, int64_t n, int ntraces,
                             float lr, float eps, void *stream);

/* ---- TransR, per-op (drop-in) route: the projections of TransRScore.prepare / create_neg_prepare
 * (models/pytorch/score_fun.py:131-166, th.matmul in the reference) and their autograd, on DENSE gathered operands ----
 * proj: [B, d_e * d_r] gathered projection rows, row i viewed as P_i [d_e, d_r] (projection_emb(rel_id), :133, :142)
 * kge_transr_project:          out[i,:] = x[i,:] P_i                                   (:135-136, :144-145)
 * kge_transr_project_bwd:      gx[i,:] = P_i gy[i,:];  gproj[i] (+)= x[i,:] (x) gy[i,:]  (either output may be NULL)
 * kge_transr_project_neg:      Y[c,i,j,:] = neg[c,j,:] P_{c,i}   [C, chunk, N, d_r]     (:147-148: every negative of a chunk
 *                              through every positive's matrix)
 * kge_transr_project_neg_bwd:  gneg[c,j,:] = sum_i gY[c,i,j,:] P_{c,i}^T;  gproj[c,i] (+)= neg_c^T gY[c,i]
 * fp32-MFMA tile routine of kge_transr.hip (64 x 64 tiles through LDS); accumulate != 0 adds into gproj. */
int kge_transr_project(const float *proj, const float *x, int64_t B, int d_e, int d_r, float *out, void *stream);
int kge_transr_project_bwd(const float *proj, const float *x, const float *gy, int64_t B, int d_e, int d_r, float *gx,
                           float *gproj, int accumulate, void *stream);
int kge_transr_project_neg(const float *proj, const float *neg, int C, int chunk, int N, int d_e, int d_r, float *Y,
                           void *stream);
int kge_transr_project_neg_bwd(const float *proj, const float *neg, const float *gY, int C, int chunk, int N, int d_e, int d_r,
                               float *gneg, float *gproj, int accumulate, void *stream);

/* ---- device-side routing of the range-sharded step (entity table sharded by id range over the ranks,
 * pull -> compute -> push with owner-side Adagrad, KEModel.pull_model / push_gradient, models/general_models.py:650-680,
 * kvserver.py:41-51) - no host round trip, fixed-size messages, so that the collectives between these calls are plain
 * equal-split all-to-alls (dglke_amd/dist.py) ----
 * kge_route_build: the batch's sorted unique entity ids (b->ue_id, count in b->counts_dev[0] or b->UE) are cut into `world` owner
 *   buckets (owner = id / rows_per_shard) of capacity `cap`:
 *     req_ids[o*cap + p]  = p-th id requested from owner o, -1 beyond the bucket's fill;
 *     cache row of an entity = o*cap + p - the position its row will have in the concatenated replies;
 *     h_loc / t_loc / neg_loc = the batch's edge ends and negative slots as cache rows, ue_loc / ue_rec_loc = the plan's id
 *     array and packed records with cache rows as ids: kge_batch_localized() points a kge_batch at them.
 *   Entries that do not fit their bucket are counted in *overflow (must be checked by the caller at its log interval: the
 *   step treats them as row world*cap, a dump row of the cache that is never sent); one workgroup, ~3 us.
 * kge_gather_rows_req: owner side of the pull - out[k,:] = table[ids[k] - id_offset,:], entries with ids[k] < 0 (pads) or
 *   outside [id_offset, id_offset + n_rows) are skipped.
 * kge_adagrad_apply_merged: owner side of the push for nsrc sources x cap messages (layout [g_0 | .. | g_{T-1} | gs_0 ..
 *   gs_{T-1} | ..], ld floats apart); the id of message k is the int32 pair id_words[k*id_stride_words], [.. + 1] (a separate
 *   int64 array: stride 2; ids inside the messages: stride ld) minus id_offset; negative ids are pads.  Within a source the
 *   ids are ascending with the pads at the end and unique; the same row may come from several sources: the wavefront of its
 *   first source applies every occurrence in source order (trace order inside a message) - deterministic, one launch. */
/* kge_route_fill: *max_fill = max(*max_fill, the largest owner-bucket fill) over n_batches batches - b0 and, for n_batches > 1, the
 *   batches whose ue_id / counts_dev arrays lie k * stride_bytes behind b0's (consecutive slots of kge_sample_batches: stride =
 *   the slot size).  One launch per GROUP of batches; the caller reads the word before the group's steps run and grows `cap`
 *   when needed, so that no entry ever overflows its bucket (dglke_amd/dist.py DistEngine.ensure_capacity). */
/* ABI 8: max_fill is int32[2] - [1] = the largest number of entries of ONE owner bucket that are in BOTH traces of a batch (the
 *   extra-region capacity `cap2` the packed messages need; 0 when the batches carry no plan records). */
int kge_route_fill(const kge_batch *b0, int n_batches, size_t stride_bytes, int world, int64_t rows_per_shard, int32_t *max_fill,
                   void *stream);
/* ABI 8, packed entity messages: ue_msg != NULL -> also writes, per union entry u, ue_msg[2u] = its message row
 *   owner * (cap + cap2) + position (the dump row world * (cap + cap2) for an entry that does not fit) and ue_msg[2u+1] = the
 *   position of its SECOND message in the bucket's extra region (row owner * (cap + cap2) + cap + p, p = rank of u among the
 *   both-trace entries of its bucket), or -1 when u is in one trace only; both-trace entries beyond cap2 are counted in
 *   *overflow.  ue_msg NULL (cap2 ignored): two-trace messages. */
int kge_route_build(const kge_batch *b, int world, int64_t rows_per_shard, int cap, int64_t *req_ids, int64_t *h_loc,
                    int64_t *t_loc, int64_t *neg_loc, int64_t *ue_loc, int32_t *ue_rec_loc, int32_t *overflow, int cap2,
                    int32_t *ue_msg, void *stream);
/* kge_route_build for a whole GROUP of batches in one launch (round 4): batch k's plan arrays lie k * in_stride_bytes behind b0's
 * (consecutive slots of kge_sample_batches), its six outputs k * out_stride_bytes behind the pointers given (one pool per group).
 * Same results per batch as kge_route_build.  Called once per sampled group, after the bucket capacity was checked (kge_route_fill):
 * the step itself then starts with the id exchange. */
int kge_route_build_group(const kge_batch *b0, int n_batches, size_t in_stride_bytes, int world, int64_t rows_per_shard, int cap,
                          int64_t *req_ids, int64_t *h_loc, int64_t *t_loc, int64_t *neg_loc, int64_t *ue_loc, int32_t *ue_rec_loc,
                          size_t out_stride_bytes, int32_t *overflow, int cap2, int32_t *ue_msg, void *stream);
int kge_batch_localized(const kge_batch *b, const int64_t *h_loc, const int64_t *t_loc, const int64_t *neg_loc,
                        const int64_t *ue_loc, const int32_t *ue_rec_loc, kge_batch *out);
int kge_gather_rows_req(const float *table, int64_t n_rows, int dim, const int64_t *ids, int64_t id_offset, int64_t n_ids,
                        float *out, void *stream);
/* ABI 8: cap_extra > 0 = PACKED messages (see kge_emit.msg_rows): source s's bucket is cap + cap_extra message rows, message
 *   (s, pos) = [g | gs | link | . .] (ntraces must be 1, ld >= dim + 4); link >= 0 names a second message of the same row at
 *   bucket position cap + link, applied behind the first (positive trace, then negative trace: the reference's order). */
int kge_adagrad_apply_merged(float *table, float *state_sum, int64_t n_rows, int dim, int nsrc, int cap, const int32_t *id_words,
                             int64_t id_stride_words, int64_t id_offset, const float *msg, int ld, int ntraces, int cap_extra,
                             float lr, float eps, void *stream);

/* two kge_adagrad_apply_merged jobs - the entity-shard apply and the relation-replica apply of one sharded step - as ONE launch
 * (independent tables and messages; same arithmetic and order per job as two separate calls: bit-identical results) */
typedef struct kge_merge_job {
    float *table, *state_sum;
    int64_t n_rows;
    int32_t dim, nsrc, cap, ld, ntraces, cap_extra;   /* cap_extra > 0: packed messages (ABI 8; 0: [g_0 | .. | gs_0 ..] messages) */
    const int32_t *id_words; int64_t id_stride_words, id_offset;
    const float *msg;
} kge_merge_job;
int kge_adagrad_apply_merged_pair(const kge_merge_job *a, const kge_merge_job *b, float lr, float eps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KGE_HIP_H */
