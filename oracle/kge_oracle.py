"""CPU oracle for the DGL-KE training hot path (gather -> score -> negative score -> loss ->
analytic gradients -> row-sparse Adagrad).

*** TEST INFRASTRUCTURE ONLY. ***  Nothing under dgl-ke_amd/ (the product) may import this
module.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
the checker / the CPU baseline, never as the thing measured or shipped.

This is a restatement in numpy of the reference's algorithm (awslabs/dgl-ke, all citations are
relative to /root/reference/python/dglke/).  The reference computes gradients with torch autograd;
here they are written out analytically because that is exactly what the HIP
kernels implement.  Parity status: PINNED - tests/test_oracle_golden.py checks every function
below against tests/golden/*.npz, which were produced by running the unmodified reference
(`KEModel.forward -> loss.backward() -> KEModel.update`) in the build container
(tests/golden/gen_golden.py).

All functions take/return numpy arrays; `dtype` selects float32 (reference precision) or float64
(tight checking).  Embedding rows of the complex models are stored as [re | im] halves
(models/pytorch/score_fun.py:298-300, 461-462).
"""
import numpy as np

MODELS = ("TransE_l1", "TransE_l2", "DistMult", "ComplEx", "RotatE", "SimplE", "RESCAL", "TransR")
SIMPLE_CLAMP = 20.0          # th.clamp(score, -20, 20): score_fun.py:568, 622, 641
LOSSES = ("Logsigmoid", "Logistic", "Hinge", "BCE")


def _canon(model):
    return "TransE_l2" if model == "TransE" else model


# --------------------------------------------------------------------------------------------
# small numerics helpers
# --------------------------------------------------------------------------------------------
def _logsigmoid(x):
    # torch.nn.functional.logsigmoid: min(x,0) - log1p(exp(-|x|))
    return np.minimum(x, 0) - np.log1p(np.exp(-np.abs(x)))


def _sigmoid(x):
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    e = np.exp(x[~pos])
    out[~pos] = e / (1.0 + e)
    return out


def _softplus(x):
    # torch softplus(beta=1, threshold=20)
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20))))


def _halves(x):
    d = x.shape[-1] // 2
    return x[..., :d], x[..., d:]


# --------------------------------------------------------------------------------------------
# A2: ExternalEmbedding.__call__ (models/pytorch/tensor_models.py:270-302): row gather
# --------------------------------------------------------------------------------------------
def gather_rows(table, idx):
    return table[idx].copy()


# --------------------------------------------------------------------------------------------
# A3: positive score, edge_func of each score function
# --------------------------------------------------------------------------------------------
def score_pos(model, h, r, t, gamma, emb_init=None):
    """TransE: score_fun.py:54-59; DistMult: :229-235; ComplEx: :297-307; RotatE: :460-472."""
    model = _canon(model)
    if model == "TransE_l1":
        return gamma - np.abs(h + r - t).sum(-1)
    if model == "TransE_l2":
        return gamma - np.sqrt(((h + r - t) ** 2).sum(-1))
    if model == "DistMult":
        return (h * r * t).sum(-1)
    if model == "ComplEx":
        rh, ih = _halves(h)
        rt, it = _halves(t)
        rr, ir = _halves(r)
        return (rh * rt * rr + ih * it * rr + rh * it * ir - ih * rt * ir).sum(-1)
    if model == "RotatE":
        rh, ih = _halves(h)
        rt, it = _halves(t)
        phase = r / (h.dtype.type(emb_init) / h.dtype.type(np.pi))
        c, s = np.cos(phase), np.sin(phase)
        re = rh * c - ih * s - rt
        im = rh * s + ih * c - it
        return gamma - np.sqrt(re * re + im * im).sum(-1)
    if model == "RESCAL":      # score_fun.py:387-394: relation row = [rel_dim, ent_dim] matrix, score = h . (M t)
        M = r.reshape(r.shape[0], h.shape[-1], -1)
        return (h * np.einsum("bij,bj->bi", M, t)).sum(-1)
    if model == "SimplE":      # score_fun.py:562-569
        hi, hj = _halves(h)
        ti, tj = _halves(t)
        rel, rinv = _halves(r)
        raw = 0.5 * (hi * rel * tj + ti * rinv * hj).sum(-1)
        return np.clip(raw, -SIMPLE_CLAMP, SIMPLE_CLAMP)
    raise ValueError(model)


def score_pos_bwd(model, h, r, t, dp, gamma, emb_init=None):
    """d(sum_i dp_i * p_i)/d(h,r,t): analytic form of autograd through edge_func."""
    model = _canon(model)
    dp = dp[:, None]
    if model in ("TransE_l1", "TransE_l2"):
        u = h + r - t
        if model == "TransE_l1":
            g = np.sign(u)
        else:
            nrm = np.sqrt((u * u).sum(-1, keepdims=True))
            g = np.where(nrm > 0, u / np.where(nrm > 0, nrm, 1), 0)
        return -dp * g, -dp * g, dp * g
    if model == "DistMult":
        return dp * r * t, dp * h * t, dp * h * r
    if model == "ComplEx":
        rh, ih = _halves(h)
        rt, it = _halves(t)
        rr, ir = _halves(r)
        gh = np.concatenate([rt * rr + it * ir, it * rr - rt * ir], -1)
        gt = np.concatenate([rh * rr - ih * ir, ih * rr + rh * ir], -1)
        gr = np.concatenate([rh * rt + ih * it, rh * it - ih * rt], -1)
        return dp * gh, dp * gr, dp * gt
    if model == "RotatE":
        rh, ih = _halves(h)
        rt, it = _halves(t)
        scale = h.dtype.type(np.pi) / h.dtype.type(emb_init)
        phase = r * scale
        c, s = np.cos(phase), np.sin(phase)
        re = rh * c - ih * s - rt
        im = rh * s + ih * c - it
        m = np.sqrt(re * re + im * im)
        inv = np.where(m > 0, 1.0 / np.where(m > 0, m, 1), 0)
        # p = gamma - sum m ; upstream on (re, im) of the rotated head is -dp * (re,im)/m
        gre = -dp * re * inv
        gim = -dp * im * inv
        gh = np.concatenate([gre * c + gim * s, -gre * s + gim * c], -1)
        gt = np.concatenate([-gre, -gim], -1)
        gphi = gre * (-rh * s - ih * c) + gim * (rh * c - ih * s)
        return gh, gphi * scale, gt
    if model == "RESCAL":
        M = r.reshape(r.shape[0], h.shape[-1], -1)
        gh = dp * np.einsum("bij,bj->bi", M, t)
        gt = dp * np.einsum("bij,bi->bj", M, h)
        gr = (dp[:, :, None] * h[:, :, None] * t[:, None, :]).reshape(r.shape)
        return gh, gr, gt
    if model == "SimplE":      # th.clamp passes the gradient where -20 <= raw <= 20
        hi, hj = _halves(h)
        ti, tj = _halves(t)
        rel, rinv = _halves(r)
        raw = 0.5 * (hi * rel * tj + ti * rinv * hj).sum(-1, keepdims=True)
        d = 0.5 * dp * (np.abs(raw) <= SIMPLE_CLAMP)
        gh = np.concatenate([d * rel * tj, d * ti * rinv], -1)
        gt = np.concatenate([d * rinv * hj, d * hi * rel], -1)
        gr = np.concatenate([d * hi * tj, d * ti * hj], -1)
        return gh, gr, gt
    raise ValueError(model)


# --------------------------------------------------------------------------------------------
# A4/A5: negative score = create_neg closures.  Split as: pos_side() builds the per-positive
# vector a_i from the uncorrupted entity and the relation, score_neg() scores a_i against the
# chunk's corrupt entities.
# --------------------------------------------------------------------------------------------
def pos_side(model, neg_head, x, r, emb_init=None):
    """x = tail rows if neg_head else head rows (general_models.py:384-388 / :410-414).
    TransE score_fun.py:94-107; DistMult :270-284; ComplEx :347-371; RotatE :516-545."""
    model = _canon(model)
    if model in ("TransE_l1", "TransE_l2"):
        return x - r if neg_head else x + r
    if model == "DistMult":
        return x * r
    if model == "RESCAL":
        # BOTH modes use M x (score_fun.py:428-447): head mode scores h'.(M t) like the positive score, tail
        # mode scores (M h).t' - NOT h.(M t') - exactly as the reference does
        M = r.reshape(r.shape[0], x.shape[-1], -1)
        return np.einsum("bij,bj->bi", M, x)
    if model == "SimplE":
        # a . neg = the un-halved chunked score: head mode (score_fun.py:611-620) pairs rel*t_j with head_i
        # and rel_inv*t_i with head_j; tail mode (:626-639) pairs rel_inv*h_j with tail_i and h_i*rel with tail_j
        xi, xj = _halves(x)
        rel, rinv = _halves(r)
        if neg_head:
            return np.concatenate([rel * xj, rinv * xi], -1)
        return np.concatenate([rinv * xj, xi * rel], -1)
    if model in ("ComplEx", "RotatE"):
        rx, ix = _halves(x)
        if model == "ComplEx":
            rr, ir = _halves(r)
        else:
            phase = r / (x.dtype.type(emb_init) / x.dtype.type(np.pi))
            rr, ir = np.cos(phase), np.sin(phase)
        if neg_head:
            return np.concatenate([rx * rr + ix * ir, -rx * ir + ix * rr], -1)
        return np.concatenate([rx * rr - ix * ir, rx * ir + ix * rr], -1)
    raise ValueError(model)


def score_neg(model, a, neg, C, chunk, N, gamma):
    """[C,chunk,N] scores.  L2 uses the reference's expansion with clamp
    (score_fun.py:26-34 batched_l2_dist); L1 = cdist p=1 (:36-38); DistMult/ComplEx = bmm
    (:275,284,359,375); RotatE = complex modulus of the broadcast difference (:526-531)."""
    model = _canon(model)
    D = a.shape[-1]
    A = a.reshape(C, chunk, D)
    Bn = neg.reshape(C, N, D)
    if model == "TransE_l2":
        asq = np.sqrt((A * A).sum(-1)) ** 2
        bsq = np.sqrt((Bn * Bn).sum(-1)) ** 2
        sq = bsq[:, None, :] - 2 * np.einsum("cik,cjk->cij", A, Bn) + asq[:, :, None]
        return gamma - np.sqrt(np.maximum(sq, a.dtype.type(1e-30)))
    if model == "TransE_l1":
        return gamma - np.abs(A[:, :, None, :] - Bn[:, None, :, :]).sum(-1)
    if model in ("DistMult", "ComplEx", "RESCAL"):
        return np.einsum("cik,cjk->cij", A, Bn)
    if model == "SimplE":
        return np.clip(0.5 * np.einsum("cik,cjk->cij", A, Bn), -SIMPLE_CLAMP, SIMPLE_CLAMP)
    if model == "RotatE":
        d = A[:, :, None, :] - Bn[:, None, :, :]
        re, im = _halves(d)
        return gamma - np.sqrt(re * re + im * im).sum(-1)
    raise ValueError(model)


def score_neg_bwd(model, a, neg, dneg, C, chunk, N, gamma):
    """(dL/da [B,D], dL/dneg [C*N,D]) given dL/dn [C,chunk,N]."""
    model = _canon(model)
    D = a.shape[-1]
    A = a.reshape(C, chunk, D)
    Bn = neg.reshape(C, N, D)
    G = dneg.reshape(C, chunk, N)
    if model == "TransE_l2":
        asq = np.sqrt((A * A).sum(-1)) ** 2
        bsq = np.sqrt((Bn * Bn).sum(-1)) ** 2
        sq = bsq[:, None, :] - 2 * np.einsum("cik,cjk->cij", A, Bn) + asq[:, :, None]
        ok = sq >= 1e-30
        dist = np.sqrt(np.maximum(sq, a.dtype.type(1e-30)))
        Cw = np.where(ok, G / dist, 0)
        ga = -A * Cw.sum(2)[:, :, None] + np.einsum("cij,cjk->cik", Cw, Bn)
        gb = np.einsum("cij,cik->cjk", Cw, A) - Bn * Cw.sum(1)[:, :, None]
    elif model == "TransE_l1":
        sg = np.sign(A[:, :, None, :] - Bn[:, None, :, :])
        ga = -(G[..., None] * sg).sum(2)
        gb = (G[..., None] * sg).sum(1)
    elif model in ("DistMult", "ComplEx", "RESCAL"):
        ga = np.einsum("cij,cjk->cik", G, Bn)
        gb = np.einsum("cij,cik->cjk", G, A)
    elif model == "SimplE":
        raw = 0.5 * np.einsum("cik,cjk->cij", A, Bn)
        Gm = 0.5 * G * (np.abs(raw) <= SIMPLE_CLAMP)
        ga = np.einsum("cij,cjk->cik", Gm, Bn)
        gb = np.einsum("cij,cik->cjk", Gm, A)
    elif model == "RotatE":
        d = A[:, :, None, :] - Bn[:, None, :, :]
        re, im = _halves(d)
        m = np.sqrt(re * re + im * im)
        inv = np.where(m > 0, 1.0 / np.where(m > 0, m, 1), 0)
        wre = G[..., None] * re * inv
        wim = G[..., None] * im * inv
        w = np.concatenate([wre, wim], -1)
        ga = -w.sum(2)
        gb = w.sum(1)
    else:
        raise ValueError(model)
    return ga.reshape(-1, D), gb.reshape(-1, D)


def pos_side_bwd(model, neg_head, x, r, ga, emb_init=None):
    """chain dL/da -> (dL/dx, dL/dr)."""
    model = _canon(model)
    if model in ("TransE_l1", "TransE_l2"):
        return ga, (-ga if neg_head else ga)
    if model == "DistMult":
        return ga * r, ga * x
    if model == "RESCAL":      # a = M x
        M = r.reshape(r.shape[0], x.shape[-1], -1)
        return np.einsum("bij,bi->bj", M, ga), (ga[:, :, None] * x[:, None, :]).reshape(r.shape)
    if model == "SimplE":
        xi, xj = _halves(x)
        rel, rinv = _halves(r)
        g1, g2 = _halves(ga)
        if neg_head:          # a = [rel * x_j | rinv * x_i]
            return np.concatenate([g2 * rinv, g1 * rel], -1), np.concatenate([g1 * xj, g2 * xi], -1)
        return np.concatenate([g2 * rel, g1 * rinv], -1), np.concatenate([g2 * xi, g1 * xj], -1)   # a = [rinv*x_j | x_i*rel]
    rx, ix = _halves(x)
    gre, gim = _halves(ga)
    if model == "ComplEx":
        rr, ir = _halves(r)
        if neg_head:
            gx = np.concatenate([gre * rr - gim * ir, gre * ir + gim * rr], -1)
            gr = np.concatenate([gre * rx + gim * ix, gre * ix - gim * rx], -1)
        else:
            gx = np.concatenate([gre * rr + gim * ir, -gre * ir + gim * rr], -1)
            gr = np.concatenate([gre * rx + gim * ix, -gre * ix + gim * rx], -1)
        return gx, gr
    if model == "RotatE":
        scale = x.dtype.type(np.pi) / x.dtype.type(emb_init)
        phase = r * scale
        c, s = np.cos(phase), np.sin(phase)
        if neg_head:
            gx = np.concatenate([gre * c - gim * s, gre * s + gim * c], -1)
            gphi = gre * (-rx * s + ix * c) + gim * (-rx * c - ix * s)
        else:
            gx = np.concatenate([gre * c + gim * s, -gre * s + gim * c], -1)
            gphi = gre * (-rx * s - ix * c) + gim * (rx * c - ix * s)
        return gx, gphi * scale
    raise ValueError(model)


# --------------------------------------------------------------------------------------------
# A6: LossGenerator.get_total_loss (models/pytorch/loss.py:69-98) and its gradient
# --------------------------------------------------------------------------------------------
def _criterion(genre, score, label, margin):
    """loss.py:10-38 ; returns (loss, dloss/dscore)"""
    if genre in ("Logsigmoid", "Logistic"):
        # -logsigmoid(l*s) (loss.py:37-38) == softplus(-l*s) (loss.py:23-24)
        z = label * score
        return -_logsigmoid(z), -label * _sigmoid(-z)
    if genre == "Hinge":
        v = margin - label * score
        return np.where(v < 0, 0, v), np.where(v < 0, 0, -label).astype(score.dtype)
    if genre == "BCE":
        # -(l*log(sig(s)) + (1-l)*log(1-sig(s)))  (loss.py:30-31); labels 1 / 0 (loss.py:54-56)
        sg = _sigmoid(score)
        val = -(label * np.log(sg) + (1 - label) * np.log(1 - sg))
        return val, (sg - label).astype(score.dtype)
    raise ValueError(genre)


def loss_fwd_bwd(pos, neg, w=None, genre="Logsigmoid", adv=False, adv_temp=1.0, pairwise=False,
                 margin=1.0):
    """pos [B], neg [B,N] (reshaped at general_models.py:560), w = edge importance [B] or None.
    Returns (pos_loss, neg_loss, loss), dL/dpos [B], dL/dneg [B,N]."""
    dt = pos.dtype
    B, N = neg.shape
    wcol = np.ones((B, 1), dt) if w is None else w.reshape(B, 1).astype(dt)
    if pairwise:
        # loss.py:76-80
        diff = pos[:, None] - neg
        val, dval = _criterion(genre, diff, 1, margin)
        loss = (val * wcol).mean()
        dd = dval * wcol / dt.type(B * N)
        return (np.nan, np.nan, loss), dd.sum(1), -dd
    neg_label = 0 if genre == "BCE" else -1
    pl, dpl = _criterion(genre, pos, 1, margin)
    nl, dnl = _criterion(genre, neg, neg_label, margin)
    # loss.py:82: `pos_loss [B] * edge_weight.view(-1, 1) [B, 1]` broadcasts to [B, B]; its mean (:92) is mean(pl) * mean(w) - every
    # positive edge carries the batch's MEAN importance (the negative part, [B, N] * [B, 1], is per edge).  Pinned by
    # goldens/transe_l2_impts (round 4: the earlier per-edge form was 0.5 % off in pos_loss, hidden by saturated positives)
    wmean = wcol.mean()
    pl = pl * wmean
    nl = nl * wcol
    if adv:
        # loss.py:87-88 ; softmax is detached
        z = neg * dt.type(adv_temp)
        z = z - z.max(1, keepdims=True)
        e = np.exp(z)
        A = e / e.sum(1, keepdims=True)
        neg_i = (A * nl).sum(1)
    else:
        A = np.full((B, N), 1.0 / N, dt)
        neg_i = nl.mean(1)
    neg_loss = neg_i.mean()
    pos_loss = pl.mean()
    loss = (neg_loss + pos_loss) / 2
    dpos = dpl * wmean / dt.type(2 * B)
    dneg = dnl * wcol * A / dt.type(2 * B)
    return (pos_loss, neg_loss, loss), dpos.astype(dt), dneg.astype(dt)


# --------------------------------------------------------------------------------------------
# A7: regularisation (general_models.py:572-576; norm = x.norm(p)**p, tensor_models.py:54)
# --------------------------------------------------------------------------------------------
def reg_value(rows_list, coef, p):
    return coef * sum((np.abs(x) ** p).sum() for x in rows_list)


def reg_grad(x, coef, p):
    return coef * p * np.abs(x) ** (p - 1) * np.sign(x)


# --------------------------------------------------------------------------------------------
# A9: ExternalEmbedding.update (tensor_models.py:304-362): row-sparse Adagrad for one trace
# --------------------------------------------------------------------------------------------
def adagrad_update(table, state, idx, grad, lr, eps=1e-10):
    """in place.  state += index_add(mean(g^2)) with duplicates accumulating (:352); std is read
    AFTER all adds (:353-356); emb.index_add_(-lr*g/std) (:357-361)."""
    gs = (grad * grad).mean(1)
    np.add.at(state, idx, gs.astype(state.dtype))
    std = np.sqrt(state[idx]) + state.dtype.type(eps)
    np.add.at(table, idx, (-lr * grad / std[:, None]).astype(table.dtype))


# --------------------------------------------------------------------------------------------
# one whole training step: KEModel.forward (general_models.py:529-578) + loss.backward()
# (train_pytorch.py:145) + KEModel.update (general_models.py:580-588)
# --------------------------------------------------------------------------------------------
class Config(object):
    def __init__(self, model, gamma, hidden, lr, adv=False, adv_temp=1.0, reg_coef=0.0,
                 reg_norm=3, loss_genre="Logsigmoid", pairwise=False, margin=1.0,
                 double_ent=False, double_rel=False, neg_deg=False):
        self.model = _canon(model)
        self.neg_deg = bool(neg_deg)                  # --neg_deg_sample (general_models.py:396-402, 424-432)
        self.gamma = gamma
        self.hidden = hidden
        self.emb_init = (gamma + 2.0) / hidden        # general_models.py:217-218, EMB_INIT_EPS=2
        self.lr = lr
        self.adv, self.adv_temp = adv, adv_temp
        self.reg_coef, self.reg_norm = reg_coef, reg_norm
        self.loss_genre, self.pairwise, self.margin = loss_genre, pairwise, margin
        self.ent_dim = 2 * hidden if double_ent else hidden
        self.rel_dim = 2 * hidden if double_rel else hidden


def forward_backward(cfg, ent, rel, nid, h_local, t_local, rel_ids, neg_ids, neg_head, chunk, N,
                     w=None):
    """Returns dict with pos_score, neg_score [C,chunk,N], log (pos_loss,neg_loss,loss,reg),
    g_pos_ent [U,D], g_rel [B,D], g_neg [C*N,D] - the three trace gradients of the reference."""
    dt = ent.dtype
    B = h_local.shape[0]
    C = B // chunk
    pos_emb = gather_rows(ent, nid)                    # trace 0 of entity_emb
    r = gather_rows(rel, rel_ids)                      # trace 0 of relation_emb
    neg = gather_rows(ent, neg_ids)                    # trace 1 of entity_emb
    h = pos_emb[h_local]
    t = pos_emb[t_local]
    gamma = dt.type(cfg.gamma)
    p = score_pos(cfg.model, h, r, t, gamma, cfg.emb_init)
    x = t if neg_head else h
    a = pos_side(cfg.model, neg_head, x, r, cfg.emb_init)
    nd = getattr(cfg, "neg_deg", False)
    if nd:
        # neg_deg_sample: the corrupted-side entities of the chunk's OWN positives are prepended to the chunk's
        # negatives (general_models.py:396-400 heads, :424-427 tails); the chunk x chunk diagonal - the positive
        # edge itself - is multiplied by 0 (mask[:, 0::(N'+1)] = 0, :401, :429-432): its SCORE becomes 0 and stays
        # in the loss, its gradient vanishes.  These rows come out of pos_g.ndata['emb'] (no new trace): their
        # gradients join the positive trace.
        y = (h if neg_head else t).reshape(C, chunk, -1)
        Np = chunk + N
        neg_all = np.concatenate([y, neg.reshape(C, N, -1)], axis=1).reshape(C * Np, -1)
        mask = np.ones((C, chunk, Np), dtype=dt)
        mask[:, np.arange(chunk), np.arange(chunk)] = 0
        n = score_neg(cfg.model, a, neg_all, C, chunk, Np, gamma) * mask
        N_sampled, N = N, Np
    else:
        n = score_neg(cfg.model, a, neg, C, chunk, N, gamma)
    (pl, nl, loss), dpos, dneg = loss_fwd_bwd(p, n.reshape(B, N), w, cfg.loss_genre, cfg.adv,
                                              cfg.adv_temp, cfg.pairwise, cfg.margin)
    reg = 0.0
    use_reg = cfg.reg_coef > 0.0 and cfg.reg_norm > 0
    if use_reg:
        reg = reg_value([pos_emb, neg], cfg.reg_coef, cfg.reg_norm) + \
            reg_value([r], cfg.reg_coef, cfg.reg_norm)
    gh, gr, gt = score_pos_bwd(cfg.model, h, r, t, dpos, gamma, cfg.emb_init)
    if nd:
        ga, g_all = score_neg_bwd(cfg.model, a, neg_all, dneg.reshape(C, chunk, N) * mask, C, chunk, N, gamma)
        g_all = g_all.reshape(C, N, -1)
        g_inb = g_all[:, :chunk].reshape(B, -1)          # gradient w.r.t. the in-batch negative rows
        g_neg = g_all[:, chunk:].reshape(C * N_sampled, -1)
    else:
        ga, g_neg = score_neg_bwd(cfg.model, a, neg, dneg.reshape(C, chunk, N), C, chunk, N, gamma)
    gx, gr2 = pos_side_bwd(cfg.model, neg_head, x, r, ga, cfg.emb_init)
    gr = gr + gr2
    if neg_head:
        gt = gt + gx
        if nd:
            gh = gh + g_inb
    else:
        gh = gh + gx
        if nd:
            gt = gt + g_inb
    g_pos = np.zeros_like(pos_emb)
    np.add.at(g_pos, h_local, gh.astype(dt))
    np.add.at(g_pos, t_local, gt.astype(dt))
    if use_reg:
        g_pos += reg_grad(pos_emb, cfg.reg_coef, cfg.reg_norm).astype(dt)
        g_neg = g_neg + reg_grad(neg, cfg.reg_coef, cfg.reg_norm)
        gr = gr + reg_grad(r, cfg.reg_coef, cfg.reg_norm)
    return dict(pos_score=p, neg_score=n, log=(pl, nl, loss, reg), loss_total=loss + reg,
                g_pos_ent=g_pos.astype(dt), g_rel=gr.astype(dt), g_neg=g_neg.astype(dt))


def train_step(cfg, ent, ent_state, rel, rel_state, nid, h_local, t_local, rel_ids, neg_ids,
               neg_head, chunk, N, w=None):
    """forward + backward + update, tables modified in place.  Update order:
    entity trace 0 (pos-unique rows) then entity trace 1 (negative rows), then the relation
    trace (general_models.py:586-588 ; tensor_models.py:316)."""
    out = forward_backward(cfg, ent, rel, nid, h_local, t_local, rel_ids, neg_ids, neg_head,
                           chunk, N, w)
    adagrad_update(ent, ent_state, nid, out["g_pos_ent"], cfg.lr)
    adagrad_update(ent, ent_state, neg_ids, out["g_neg"], cfg.lr)
    adagrad_update(rel, rel_state, rel_ids, out["g_rel"], cfg.lr)
    return out


def train_steps_async(cfg, ent, ent_state, rel, rel_state, batches, defer_rel=False):
    """--async_update (tensor_models.py:136-175 async_update process, :325-328 queue hand-off, :364-375;
    general_models.py:639-647; train_pytorch.py:120-121, 194-195) for a GROUP of consecutive steps, restated
    deterministically: the reference hands the entity gradients of step s to a helper process through a
    1-slot queue and starts step s+1 at once, so the gather of step s+1 may or may not see update s (racy,
    <= 1 step of staleness).  This restatement fixes the race at its bound: step s gathers rows that contain
    every update up to s-2 and never update s-1, which lands while step s is being scored; the last update
    is applied at the end of the group (finish_async_update).  Gradients (including the regulariser) are
    those of the rows as gathered, applied to the rows as they are when the update lands, exactly like the
    helper process does.  Only the ENTITY table is deferred (create_async_update is called on entity_emb
    only); defer_rel=True defers the relation trace too (KGE_FLAG_ASYNC_REL).
    `batches`: dicts with nid, h_local, t_local, r, neg, neg_head, chunk, N.  Tables modified in place."""
    outs, pending = [], None

    def land(p):
        adagrad_update(ent, ent_state, p["nid"], p["g_pos_ent"], cfg.lr)
        adagrad_update(ent, ent_state, p["neg"], p["g_neg"], cfg.lr)
        if defer_rel:
            adagrad_update(rel, rel_state, p["r"], p["g_rel"], cfg.lr)

    for bt in batches:
        out = forward_backward(cfg, ent, rel, bt["nid"], bt["h_local"], bt["t_local"], bt["r"], bt["neg"],
                               bt["neg_head"], bt["chunk"], bt["N"], bt.get("w"))
        if pending is not None:
            land(pending)
        if not defer_rel:
            adagrad_update(rel, rel_state, bt["r"], out["g_rel"], cfg.lr)
        pending = dict(nid=bt["nid"], neg=bt["neg"], r=bt["r"], g_pos_ent=out["g_pos_ent"], g_neg=out["g_neg"],
                       g_rel=out["g_rel"])
        outs.append(out)
    if pending is not None:
        land(pending)
    return outs


# ---------------------------------------------------------------------------------------------
# TransR (score_fun.py:110-220): a THIRD table, projection_emb [n_rel, ent_dim * rel_dim], owned by the score
# function.  prepare() (:131-136) projects head and tail of every positive edge with its relation's matrix
# (projection trace 0); the neg-prepare closure (:138-166) gathers the matrices AGAIN (trace 1), projects the
# uncorrupted entity and projects EVERY negative of the chunk with EVERY positive's matrix; scores are L1
# distances in relation space (:122-127, :199-219).  The regulariser covers entity and relation traces only
# (general_models.py:572-576); projection_emb is initialised U(-1, 1) (:171).
# ---------------------------------------------------------------------------------------------
def transr_forward_backward(cfg, ent, rel, proj, nid, h_local, t_local, rel_ids, neg_ids, neg_head, chunk, N, w=None):
    dt = ent.dtype
    B, C = h_local.shape[0], h_local.shape[0] // chunk
    De, Dr = ent.shape[1], rel.shape[1]
    pos_emb, r, neg = gather_rows(ent, nid), gather_rows(rel, rel_ids), gather_rows(ent, neg_ids)
    P = gather_rows(proj, rel_ids).reshape(B, De, Dr)
    h, t = pos_emb[h_local], pos_emb[t_local]
    gamma = dt.type(cfg.gamma)
    hp, tp = np.einsum("ab,abc->ac", h, P), np.einsum("ab,abc->ac", t, P)
    u = hp + r - tp
    p = gamma - np.abs(u).sum(-1)
    x = t if neg_head else h
    xp = np.einsum("ab,abc->ac", x, P)
    # NOTE: BOTH closures subtract the relation (score_fun.py:203-204 head mode: tails - relations; :212-213
    # tail mode: heads - relations - not heads + relations); kept as the reference computes it
    q = xp - r                                                           # [B, Dr]
    nd = getattr(cfg, "neg_deg", False)
    N_sampled, neg_s = N, neg
    if nd:
        # --neg_deg_sample is model-agnostic in the reference (general_models.py:396-402, 417-423): the corrupted-side entities of the
        # chunk's own positives are concatenated IN FRONT of the sampled negatives BEFORE head_neg_prepare / tail_neg_prepare, so
        # TransR projects them like every other negative; the chunk x chunk diagonal is multiplied by 0 (:401, :429-432).  The
        # in-batch rows are slices of pos_g.ndata['emb']: their gradient joins the positive trace, the regulariser does not see them.
        y_own = (h if neg_head else t).reshape(C, chunk, De)
        neg = np.concatenate([y_own, neg.reshape(C, N, De)], axis=1).reshape(C * (chunk + N), De)
        N = chunk + N
        mask = np.ones((C, chunk, N), dtype=dt)
        mask[:, np.arange(chunk), np.arange(chunk)] = 0
    Y = np.einsum("cjd,cide->cije", neg.reshape(C, N, De), P.reshape(C, chunk, De, Dr))   # [C, chunk, N, Dr]
    D = Y - q.reshape(C, chunk, 1, Dr)                                   # head mode: heads - tails; tail mode: its negative
    n = gamma - np.abs(D).sum(-1)
    if nd:
        n = n * mask
    (pl, nl, loss), dpos, dneg = loss_fwd_bwd(p, n.reshape(B, N), w, cfg.loss_genre, cfg.adv, cfg.adv_temp,
                                              cfg.pairwise, cfg.margin)
    reg = 0.0
    use_reg = cfg.reg_coef > 0.0 and cfg.reg_norm > 0
    if use_reg:
        reg = reg_value([pos_emb, neg_s], cfg.reg_coef, cfg.reg_norm) + reg_value([r], cfg.reg_coef, cfg.reg_norm)
    # positive score: p = gamma - |hp + r - tp|_1
    s = np.sign(u)
    ghp, gtp = -dpos[:, None] * s, dpos[:, None] * s
    gr = -dpos[:, None] * s
    g_proj0 = h[:, :, None] * ghp[:, None, :] + t[:, :, None] * gtp[:, None, :]
    gh, gt = np.einsum("abc,ac->ab", P, ghp), np.einsum("abc,ac->ab", P, gtp)
    # negative scores: n = gamma - sum |Y - q|  (either sign convention): dY = -W sign(Y - q), dq = -sum_j dY
    if nd:
        dneg = (dneg.reshape(C, chunk, N) * mask).reshape(B, N)
    dY = -dneg.reshape(C, chunk, N, 1) * np.sign(D)
    dq = -dY.sum(2).reshape(B, Dr)
    g_neg = np.einsum("cije,cide->cjd", dY, P.reshape(C, chunk, De, Dr)).reshape(C * N, De)
    g_proj1 = np.einsum("cjd,cije->cide", neg.reshape(C, N, De), dY).reshape(B, De, Dr) + x[:, :, None] * dq[:, None, :]
    gx = np.einsum("abc,ac->ab", P, dq)
    gr = gr - dq
    if neg_head:
        gt = gt + gx
    else:
        gh = gh + gx
    if nd:
        g_all = g_neg.reshape(C, N, De)
        g_inb = g_all[:, :chunk].reshape(B, De)          # gradient w.r.t. the in-batch negative rows -> positive trace
        g_neg = g_all[:, chunk:].reshape(C * N_sampled, De)
        if neg_head:
            gh = gh + g_inb
        else:
            gt = gt + g_inb
    g_pos = np.zeros_like(pos_emb)
    np.add.at(g_pos, h_local, gh)
    np.add.at(g_pos, t_local, gt)
    g_neg = g_neg.astype(dt)
    if use_reg:
        g_pos += reg_grad(pos_emb, cfg.reg_coef, cfg.reg_norm)
        g_neg = g_neg + reg_grad(neg_s, cfg.reg_coef, cfg.reg_norm)
        gr = gr + reg_grad(r, cfg.reg_coef, cfg.reg_norm)
    return dict(pos_score=p, neg_score=n, log=(pl, nl, loss, reg), loss_total=loss + reg,
                g_pos_ent=g_pos.astype(dt), g_rel=gr.astype(dt), g_neg=g_neg.astype(dt),
                g_proj0=g_proj0.reshape(B, De * Dr).astype(dt), g_proj1=g_proj1.reshape(B, De * Dr).astype(dt))


def transr_train_step(cfg, ent, ent_state, rel, rel_state, proj, proj_state, nid, h_local, t_local, rel_ids, neg_ids,
                      neg_head, chunk, N, w=None):
    """update order: entity traces, relation trace (KEModel.update, general_models.py:580-588), then
    score_func.update() = projection trace 0 (prepare) and trace 1 (neg-prepare), score_fun.py:173-174."""
    out = transr_forward_backward(cfg, ent, rel, proj, nid, h_local, t_local, rel_ids, neg_ids, neg_head, chunk, N, w)
    adagrad_update(ent, ent_state, nid, out["g_pos_ent"], cfg.lr)
    adagrad_update(ent, ent_state, neg_ids, out["g_neg"], cfg.lr)
    adagrad_update(rel, rel_state, rel_ids, out["g_rel"], cfg.lr)
    adagrad_update(proj, proj_state, rel_ids, out["g_proj0"], cfg.lr)
    adagrad_update(proj, proj_state, rel_ids, out["g_proj1"], cfg.lr)
    return out


def synth_batch(rng, n_ent, n_rel, B, N, chunk, step):
    """Seeded synthetic id batch (same generator as tests/golden/gen_golden.py:make_batch):
    uniform h,t,r; C*N uniform negatives with replacement, positives not excluded
    (dataloader/sampler.py:376-419 call-site semantics); odd steps corrupt tails, even steps
    corrupt heads (dataloader/sampler.py:853-859)."""
    C = B // chunk
    h = rng.randint(0, n_ent, size=B).astype(np.int64)
    t = rng.randint(0, n_ent, size=B).astype(np.int64)
    r = rng.randint(0, n_rel, size=B).astype(np.int64)
    neg = rng.randint(0, n_ent, size=C * N).astype(np.int64)
    nid, inv = np.unique(np.concatenate([h, t]), return_inverse=True)
    return dict(h=h, t=t, r=r, neg=neg, neg_head=(step % 2 == 0), nid=nid.astype(np.int64),
                h_local=inv[:B].astype(np.int64), t_local=inv[B:].astype(np.int64), C=C)


# ---------------------------------------------------------------------------------------------
# ranking evaluation: KEModel.forward_test (general_models.py:436-485) for E test triples scored
# against ALL entities in one chunk (EvalSampler with neg_sample_size_eval = -1 collapses to one
# chunk, dataloader/sampler.py:492-495); false negatives = corrupted triples that exist in the
# graph (`bias == -1`, sampler.py:586-587; general_models.py:463-475)
# ---------------------------------------------------------------------------------------------
def false_negative_mask(known, h, r, t, neg_head, n_ent):
    """[E, n_ent] bool: True where replacing the head (neg_head) / tail of test triple i by entity e
    gives a known triple."""
    ks = set(map(tuple, np.asarray(known, np.int64).tolist()))
    E = len(h)
    m = np.zeros((E, n_ent), bool)
    for i in range(E):
        for e in range(n_ent):
            c = (e, int(r[i]), int(t[i])) if neg_head else (int(h[i]), int(r[i]), e)
            m[i, e] = c in ks
    return m


def rank_eval(model, ent, rel, h, r, t, neg_head, gamma, emb_init, false_neg=None, tol=0.0, proj=None):
    """returns (ranks [E], pos_score [E], neg_score [E, n_ent]); with tol > 0 `ranks` is a pair
    (lowest, highest) rank consistent with scores perturbed by at most tol (tie tolerance for fp32
    implementations whose rounding differs from the reference's)."""
    E, n_ent = len(h), ent.shape[0]
    hs, rs, ts = ent[h], rel[r], ent[t]
    if _canon(model) == "TransR":          # projections of both ends and of every candidate (score_fun.py:131-166)
        P = proj[r].reshape(E, ent.shape[1], rel.shape[1])
        hp, tp = np.einsum("ab,abc->ac", hs, P), np.einsum("ab,abc->ac", ts, P)
        p = gamma - np.abs(hp + rs - tp).sum(-1)
        q = (tp if neg_head else hp) - rs
        S = gamma - np.abs(np.einsum("jd,ide->ije", ent, P) - q[:, None, :]).sum(-1)
    else:
        p = score_pos(model, hs, rs, ts, gamma, emb_init)
        a = pos_side(model, neg_head, ts if neg_head else hs, rs, emb_init)
        S = score_neg(model, a, ent, 1, E, n_ent, gamma)[0]
    keep = np.ones_like(S, bool) if false_neg is None else ~false_neg
    if tol == 0.0:
        return ((S >= p[:, None]) & keep).sum(1) + 1, p, S
    lo = ((S >= p[:, None] + tol) & keep).sum(1) + 1
    hi = ((S >= p[:, None] - tol) & keep).sum(1) + 1
    return (lo, hi), p, S
