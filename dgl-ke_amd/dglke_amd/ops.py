"""torch-facing wrappers of the modular C-ABI ops (one per reference op, include/kge_hip.h) and the
autograd Functions that let `loss.backward()` (train_pytorch.py:145) run the analytic HIP
backward kernels instead of torch autograd through eager ops.

torch is plumbing here: device memory (tensors), the current stream, and the autograd graph
between the ops.  All arithmetic happens in libkge_hip.so.
"""
import torch

from . import _lib
from ._lib import check, lib, model_id, ptr, stream_ptr


def _f32(t):
    if t.dtype != torch.float32:
        raise _lib.KgeError("expected float32, got %s" % t.dtype)
    return t.contiguous()


def _i64(t):
    if t.dtype != torch.int64:
        raise _lib.KgeError("expected int64 ids, got %s" % t.dtype)
    return t.contiguous()


def gather_rows(table, idx):
    """ExternalEmbedding.__call__ row gather (tensor_models.py:292)."""
    table, idx = _f32(table), _i64(idx)
    out = torch.empty((idx.shape[0], table.shape[1]), dtype=torch.float32, device=table.device)
    check(lib().kge_gather_rows(ptr(table), table.shape[0], table.shape[1], ptr(idx), idx.shape[0],
                                ptr(out), stream_ptr()))
    return out


def adagrad_scatter(table, state_sum, idx, grad, lr, eps=1e-10):
    """ExternalEmbedding.update for one trace (tensor_models.py:304-362), in place."""
    table, state_sum, idx, grad = _f32(table), _f32(state_sum), _i64(idx), _f32(grad)
    check(lib().kge_adagrad_scatter(ptr(table), ptr(state_sum), table.shape[0], table.shape[1],
                                    ptr(idx), ptr(grad), idx.shape[0], float(lr), float(eps),
                                    stream_ptr()))


def adagrad_apply_rows(table, state_sum, idx, g, gs, lr, eps=1e-10):
    table, state_sum, idx, g, gs = _f32(table), _f32(state_sum), _i64(idx), _f32(g), _f32(gs)
    check(lib().kge_adagrad_apply_rows(ptr(table), ptr(state_sum), table.shape[0], table.shape[1],
                                       ptr(idx), ptr(g), ptr(gs), idx.shape[0], float(lr),
                                       float(eps), stream_ptr()))


class _PosScore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, r, t, model, gamma, emb_init):
        h, r, t = _f32(h), _f32(r), _f32(t)
        out = torch.empty(h.shape[0], dtype=torch.float32, device=h.device)
        check(lib().kge_score_pos(model, ptr(h), ptr(r), ptr(t), h.shape[0], h.shape[1], r.shape[1],
                                  gamma, emb_init, ptr(out), stream_ptr()))
        ctx.save_for_backward(h, r, t)
        ctx.meta = (model, gamma, emb_init)
        return out

    @staticmethod
    def backward(ctx, dpos):
        h, r, t = ctx.saved_tensors
        model, gamma, emb_init = ctx.meta
        dpos = _f32(dpos)
        gh, gr, gt = torch.empty_like(h), torch.empty_like(r), torch.empty_like(t)
        check(lib().kge_score_pos_bwd(model, ptr(h), ptr(r), ptr(t), ptr(dpos), h.shape[0],
                                      h.shape[1], r.shape[1], gamma, emb_init, ptr(gh), ptr(gr),
                                      ptr(gt), stream_ptr()))
        return gh, gr, gt, None, None, None


def score_pos(model_name, h, r, t, gamma, emb_init=1.0):
    """score_func.edge_func (score_fun.py:54-59, 229-235, 297-307, 460-472)."""
    return _PosScore.apply(h, r, t, model_id(model_name), float(gamma), float(emb_init))


class _NegScore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos_side, rel, neg, model, neg_head, C, chunk, N, gamma, emb_init, flags):
        pos_side, rel, neg = _f32(pos_side), _f32(rel), _f32(neg)
        d_e, d_r = pos_side.shape[1], rel.shape[1]
        if pos_side.shape[0] != C * chunk or neg.shape[0] != C * N:
            raise _lib.KgeError("score_neg: shapes do not match num_chunks/chunk_size/neg_sample_size")
        out = torch.empty((C, chunk, N), dtype=torch.float32, device=pos_side.device)
        wsb = lib().kge_score_neg_workspace_bytes(model, C, chunk, N, d_e)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=pos_side.device)
        check(lib().kge_score_neg_fwd(model, int(neg_head), ptr(pos_side), ptr(rel), ptr(neg), C,
                                      chunk, N, d_e, d_r, gamma, emb_init, ptr(out), ptr(ws), wsb,
                                      flags, stream_ptr()))
        ctx.save_for_backward(pos_side, rel, neg, out)
        ctx.meta = (model, int(neg_head), C, chunk, N, gamma, emb_init, flags)
        return out

    @staticmethod
    def backward(ctx, dneg):
        pos_side, rel, neg, out = ctx.saved_tensors
        model, neg_head, C, chunk, N, gamma, emb_init, flags = ctx.meta
        dneg = _f32(dneg)
        d_e, d_r = pos_side.shape[1], rel.shape[1]
        g_x, g_r, g_n = torch.empty_like(pos_side), torch.empty_like(rel), torch.empty_like(neg)
        wsb = lib().kge_score_neg_workspace_bytes(model, C, chunk, N, d_e)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=pos_side.device)
        check(lib().kge_score_neg_bwd(model, neg_head, ptr(pos_side), ptr(rel), ptr(neg), ptr(out),
                                      ptr(dneg), C, chunk, N, d_e, d_r, gamma, emb_init, ptr(g_x),
                                      ptr(g_r), ptr(g_n), ptr(ws), wsb, flags, stream_ptr()))
        return (g_x, g_r, g_n) + (None,) * 8


def score_neg(model_name, neg_head, pos_side, rel, neg, num_chunks, chunk_size, neg_sample_size,
              gamma, emb_init=1.0, flags=0):
    """create_neg(neg_head) closure (score_fun.py:91-108, 268-286, 345-376, 512-554)."""
    return _NegScore.apply(pos_side, rel, neg, model_id(model_name), bool(neg_head),
                           int(num_chunks), int(chunk_size), int(neg_sample_size), float(gamma),
                           float(emb_init), int(flags))


class _TransRProject(torch.autograd.Function):
    """y_i = x_i P_i for B edges (TransRScore.prepare, score_fun.py:131-136; the reference: th.matmul)."""
    @staticmethod
    def forward(ctx, x, proj, d_e, d_r):
        x, proj = _f32(x), _f32(proj)
        if x.shape[1] != d_e or proj.shape != (x.shape[0], d_e * d_r):
            raise _lib.KgeError("transr_project: x [B, d_e], proj [B, d_e * d_r] expected")
        out = torch.empty((x.shape[0], d_r), dtype=torch.float32, device=x.device)
        check(lib().kge_transr_project(ptr(proj), ptr(x), x.shape[0], d_e, d_r, ptr(out), stream_ptr()))
        ctx.save_for_backward(x, proj)
        ctx.meta = (d_e, d_r)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, proj = ctx.saved_tensors
        d_e, d_r = ctx.meta
        gy = _f32(gy)
        gx, gp = torch.empty_like(x), torch.empty_like(proj)
        check(lib().kge_transr_project_bwd(ptr(proj), ptr(x), ptr(gy), x.shape[0], d_e, d_r, ptr(gx), ptr(gp), 0, stream_ptr()))
        return gx, gp, None, None


def transr_project(x, proj, d_e, d_r):
    return _TransRProject.apply(x, proj, int(d_e), int(d_r))


class _TransRProjectNeg(torch.autograd.Function):
    """Y[c,i,j,:] = neg[c,j,:] P_{c,i}: every negative of a chunk through every positive's matrix (the neg-prepare closure,
    score_fun.py:138-166) - B batched [N x d_e] x [d_e x d_r] products on the fp32-MFMA tile routine of kge_transr.hip."""
    @staticmethod
    def forward(ctx, neg, proj, C, chunk, N, d_e, d_r):
        neg, proj = _f32(neg), _f32(proj)
        if neg.shape != (C * N, d_e) or proj.shape != (C * chunk, d_e * d_r):
            raise _lib.KgeError("transr_project_neg: neg [C*N, d_e], proj [C*chunk, d_e * d_r] expected")
        Y = torch.empty((C, chunk, N, d_r), dtype=torch.float32, device=neg.device)
        check(lib().kge_transr_project_neg(ptr(proj), ptr(neg), C, chunk, N, d_e, d_r, ptr(Y), stream_ptr()))
        ctx.save_for_backward(neg, proj)
        ctx.meta = (C, chunk, N, d_e, d_r)
        return Y

    @staticmethod
    def backward(ctx, gY):
        neg, proj = ctx.saved_tensors
        C, chunk, N, d_e, d_r = ctx.meta
        gY = _f32(gY)
        gn, gp = torch.empty_like(neg), torch.empty_like(proj)
        check(lib().kge_transr_project_neg_bwd(ptr(proj), ptr(neg), ptr(gY), C, chunk, N, d_e, d_r, ptr(gn), ptr(gp), 0,
                                               stream_ptr()))
        return gn, gp, None, None, None, None, None


def transr_project_neg(neg, proj, num_chunks, chunk_size, neg_sample_size, d_e, d_r):
    return _TransRProjectNeg.apply(neg, proj, int(num_chunks), int(chunk_size), int(neg_sample_size), int(d_e), int(d_r))


class _Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, neg, w, genre, adv, adv_temp, pairwise, margin):
        pos, neg = _f32(pos), _f32(neg)
        B, N = neg.shape
        loss3 = torch.empty(3, dtype=torch.float32, device=pos.device)
        dpos, dneg = torch.empty_like(pos), torch.empty_like(neg)
        wsb = (2 * B + 64) * 4 + 1024
        ws = torch.empty(wsb, dtype=torch.uint8, device=pos.device)
        check(lib().kge_loss_fwd_bwd(genre, int(adv), adv_temp, int(pairwise), margin, ptr(pos),
                                     ptr(neg), ptr(_f32(w)) if w is not None else None, B, N,
                                     ptr(loss3), ptr(dpos), ptr(dneg), ptr(ws), wsb, stream_ptr()))
        ctx.save_for_backward(dpos, dneg)
        ctx.mark_non_differentiable(loss3)
        return loss3[2], loss3

    @staticmethod
    def backward(ctx, gl, _g3):
        dpos, dneg = ctx.saved_tensors
        return dpos * gl, dneg * gl, None, None, None, None, None, None


def loss_fwd_bwd(pos, neg, w, genre, adv, adv_temp, pairwise, margin):
    """LossGenerator.get_total_loss (loss.py:69-98): returns (loss scalar with grad_fn,
    loss3 = [pos_loss, neg_loss, loss])."""
    return _Loss.apply(pos, neg, w, _lib.LOSS_IDS[genre], bool(adv), float(adv_temp),
                       bool(pairwise), float(margin))


class _GatherLocal(torch.autograd.Function):
    """rows of a [U, D] block through LOCAL ids (pos_g.ndata['emb'][head_ids], general_models.py:384-388, 410-414):
    forward = kge_gather_rows, backward = kge_scatter_add_rows (index_add with duplicates, like torch's autograd)."""
    @staticmethod
    def forward(ctx, block, idx):
        block, idx = _f32(block), _i64(idx)
        out = torch.empty((idx.shape[0], block.shape[1]), dtype=torch.float32, device=block.device)
        check(lib().kge_gather_rows(ptr(block), block.shape[0], block.shape[1], ptr(idx), idx.shape[0], ptr(out),
                                    stream_ptr()))
        ctx.save_for_backward(idx)
        ctx.shape = tuple(block.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = _f32(g)
        out = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        check(lib().kge_scatter_add_rows(ptr(out), out.shape[0], out.shape[1], ptr(idx), ptr(g), idx.shape[0],
                                         stream_ptr()))
        return out, None


def gather_local(block, idx):
    return _GatherLocal.apply(block, idx)


class _PnormPow(torch.autograd.Function):
    """x.norm(p) ** p (tensor_models.py:54 `norm`), value and gradient in libkge_hip."""
    @staticmethod
    def forward(ctx, x, p):
        x = _f32(x)
        x2 = x.reshape(-1, x.shape[-1]) if x.dim() > 1 else x.reshape(1, -1)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        ws = torch.empty(max(x2.shape[0], 1), dtype=torch.float32, device=x.device)
        check(lib().kge_pnorm_pow(ptr(x2), x2.shape[0], x2.shape[1], int(p), ptr(out), ptr(ws), ws.numel() * 4,
                                  stream_ptr()))
        ctx.save_for_backward(x2)
        ctx.meta = (int(p), tuple(x.shape))
        return out

    @staticmethod
    def backward(ctx, gout):
        (x2,) = ctx.saved_tensors
        p, shape = ctx.meta
        gx = torch.empty_like(x2)
        check(lib().kge_pnorm_pow_bwd(ptr(x2), x2.shape[0], x2.shape[1], p, ptr(_f32(gout).reshape(1)), ptr(gx),
                                      stream_ptr()))
        return gx.reshape(shape), None


def pnorm_pow(x, p):
    return _PnormPow.apply(x, p)


class _MaskDiag(torch.autograd.Function):
    """--neg_deg_sample: the chunk x chunk diagonal of the [C, chunk, Np] score block is the positive edge itself -
    its score becomes 0 and carries no gradient (general_models.py:401-402, 429-432)."""
    @staticmethod
    def forward(ctx, x, C, chunk, Np):
        y = _f32(x).clone()
        check(lib().kge_mask_diag(ptr(y), C, chunk, Np, stream_ptr()))
        ctx.meta = (C, chunk, Np)
        return y

    @staticmethod
    def backward(ctx, g):
        C, chunk, Np = ctx.meta
        g = _f32(g).clone()
        check(lib().kge_mask_diag(ptr(g), C, chunk, Np, stream_ptr()))
        return g, None, None, None


def mask_diag(x, C, chunk, Np):
    return _MaskDiag.apply(x, int(C), int(chunk), int(Np))


def rank_from_scores(neg, pos, bias=None):
    """rankings of KEModel.forward_test (general_models.py:463-478): 1 + #{j: neg[i,j] >= pos[i], bias[i,j] != -1}"""
    neg, pos = _f32(neg), _f32(pos).reshape(-1)
    E, N = neg.shape
    ranks = torch.empty(E, dtype=torch.int64, device=neg.device)
    b = _f32(bias.to(neg.device).float()) if bias is not None else None
    check(lib().kge_rank_from_scores(ptr(neg), ptr(pos), ptr(b) if b is not None else None, E, N, ptr(ranks), stream_ptr()))
    return ranks
