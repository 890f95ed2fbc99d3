"""torch-CPU port of one reference training step, used ONLY as bench.py's `cpu_baseline`
(kind "port") and cross-checked against the golden vectors in tests/.

*** TEST / MEASUREMENT INFRASTRUCTURE - never imported by the product (dgl-ke_amd/). ***

/root/reference is pure Python on torch + DGL and cannot travel to the GPU box, so the CPU
baseline cannot be the reference itself.  This file restates `KEModel.forward -> loss.backward()
-> KEModel.update` (models/general_models.py:529-588, train_pytorch.py:141-152) with the SAME
torch ops the reference uses on its CPU path - advanced-index gather + clone (tensor_models.py:
292-298), th.baddbmm / th.cdist / th.bmm negative scores (score_fun.py:26-38, 275-284, 359-375,
526-531), logsigmoid + softmax adversarial weighting (loss.py:87-98), autograd backward, and
index_add_ Adagrad (tensor_models.py:330-361) - on flat id arrays instead of DGL subgraphs, so its
wall time is representative of the reference's `--num_proc 1` CPU path without the DGL sampler.
"""
import numpy as np
import torch as th
import torch.nn.functional as Fn


class TorchPort(object):
    def __init__(self, model, n_ent, n_rel, hidden, gamma, lr, de=False, dr=False, adv=False,
                 adv_temp=1.0, reg_coef=0.0, reg_norm=3, seed=0):
        self.model = "TransE_l2" if model == "TransE" else model
        self.gamma, self.lr = gamma, lr
        self.adv, self.adv_temp = adv, adv_temp
        self.reg_coef, self.reg_norm = reg_coef, reg_norm
        self.emb_init = (gamma + 2.0) / hidden
        g = th.Generator().manual_seed(seed)
        d_e = 2 * hidden if de else hidden
        d_r = 2 * hidden if dr else hidden
        if model == "RESCAL":
            d_r = d_r * d_e
        self.ent = (th.rand(n_ent, d_e, generator=g) * 2 - 1) * self.emb_init
        self.rel = (th.rand(n_rel, d_r, generator=g) * 2 - 1) * self.emb_init
        self.ent_state = th.zeros(n_ent)
        self.rel_state = th.zeros(n_rel)

    # --- score functions (same op sequences as models/pytorch/score_fun.py) -------------------
    def _pos(self, h, r, t):
        m = self.model
        if m == "TransE_l2":
            return self.gamma - th.norm(h + r - t, p=2, dim=-1)
        if m == "TransE_l1":
            return self.gamma - th.norm(h + r - t, p=1, dim=-1)
        if m == "DistMult":
            return th.sum(h * r * t, dim=-1)
        if m == "ComplEx":
            rh, ih = th.chunk(h, 2, dim=-1)
            rt, it = th.chunk(t, 2, dim=-1)
            rr, ir = th.chunk(r, 2, dim=-1)
            return th.sum(rh * rt * rr + ih * it * rr + rh * it * ir - ih * rt * ir, -1)
        if m == "RESCAL":
            M = r.view(-1, h.shape[1], h.shape[1])
            return th.sum(h * th.matmul(M, t.unsqueeze(-1)).squeeze(-1), dim=-1)
        if m == "SimplE":
            hi, hj = th.chunk(h, 2, dim=-1)
            ti, tj = th.chunk(t, 2, dim=-1)
            rel, rinv = th.chunk(r, 2, dim=-1)
            return th.clamp(0.5 * (hi * rel * tj + ti * rinv * hj).sum(-1), -20, 20)
        rh, ih = th.chunk(h, 2, dim=-1)
        rt, it = th.chunk(t, 2, dim=-1)
        ph = r / (self.emb_init / np.pi)
        c, s = th.cos(ph), th.sin(ph)
        re = rh * c - ih * s - rt
        im = rh * s + ih * c - it
        return self.gamma - th.stack([re, im], dim=0).norm(dim=0).sum(-1)

    def _neg(self, x, r, neg, neg_head, C, chunk, N):
        m = self.model
        D = x.shape[1]
        if m in ("TransE_l2", "TransE_l1"):
            a = (x - r) if neg_head else (x + r)
            a = a.reshape(C, chunk, D)
            b = neg.reshape(C, N, D)
            if m == "TransE_l1":
                return self.gamma - th.cdist(a, b, p=1)
            a2 = a.norm(dim=-1).pow(2)
            b2 = b.norm(dim=-1).pow(2)
            sq = th.baddbmm(b2.unsqueeze(-2), a, b.transpose(-2, -1), alpha=-2).add_(a2.unsqueeze(-1))
            return self.gamma - sq.clamp_min_(1e-30).sqrt_()
        if m == "DistMult":
            return th.bmm((x * r).reshape(C, chunk, D), neg.reshape(C, N, D).transpose(1, 2))
        if m == "RESCAL":
            tmp = th.matmul(r.view(-1, D, D), x.unsqueeze(-1)).squeeze(-1).reshape(C, chunk, D)
            return th.bmm(tmp, neg.reshape(C, N, D).transpose(1, 2))
        if m == "SimplE":
            xi, xj = x[..., :D // 2], x[..., D // 2:]
            rel, rinv = r[..., :D // 2], r[..., D // 2:]
            nt = neg.reshape(C, N, D).transpose(1, 2)
            if neg_head:       # x = tails, neg = heads
                fwd, bwd = (rel * xj).reshape(C, chunk, D // 2), (rinv * xi).reshape(C, chunk, D // 2)
                tmp = 0.5 * (th.bmm(fwd, nt[..., :D // 2, :]) + th.bmm(bwd, nt[..., D // 2:, :]))
            else:              # x = heads, neg = tails
                fwd, bwd = (xi * rel).reshape(C, chunk, D // 2), (rinv * xj).reshape(C, chunk, D // 2)
                tmp = 0.5 * (th.bmm(fwd, nt[..., D // 2:, :]) + th.bmm(bwd, nt[..., :D // 2, :]))
            return th.clamp(tmp, -20, 20)
        rx, ix = x[..., :D // 2], x[..., D // 2:]
        if m == "ComplEx":
            rr, ir = r[..., :D // 2], r[..., D // 2:]
        else:
            ph = r / (self.emb_init / np.pi)
            rr, ir = th.cos(ph), th.sin(ph)
        if neg_head:
            real, imag = rx * rr + ix * ir, -rx * ir + ix * rr
        else:
            real, imag = rx * rr - ix * ir, rx * ir + ix * rr
        a = th.cat((real, imag), dim=-1)
        if m == "ComplEx":
            return th.bmm(a.reshape(C, chunk, D), neg.reshape(C, N, D).transpose(1, 2))
        sc = a.reshape(C, chunk, 1, D) - neg.reshape(C, 1, N, D)
        sc = th.stack([sc[..., :D // 2], sc[..., D // 2:]], dim=-1).norm(dim=-1)
        return self.gamma - sc.sum(-1)

    def step(self, p):
        """p: plan dict (nid, h_local, t_local, rel_ids, neg_ids, C, chunk, N, neg_head)."""
        nid = th.from_numpy(p["nid"])
        rid = th.from_numpy(p["rel_ids"])
        gid = th.from_numpy(p["neg_ids"])
        hl, tl = th.from_numpy(p["h_local"]), th.from_numpy(p["t_local"])
        C, chunk, N, neg_head = p["C"], p["chunk"], p["N"], bool(p["neg_head"])
        pos_emb = self.ent[nid].clone().detach().requires_grad_(True)
        rel = self.rel[rid].clone().detach().requires_grad_(True)
        neg = self.ent[gid].clone().detach().requires_grad_(True)
        h, t = pos_emb[hl], pos_emb[tl]
        pos = self._pos(h, rel, t)
        ns = self._neg(t if neg_head else h, rel, neg, neg_head, C, chunk, N).reshape(-1, N)
        pos_loss = -Fn.logsigmoid(pos)
        neg_loss = -Fn.logsigmoid(-ns)
        if self.adv:
            neg_loss = th.sum(th.softmax(ns * self.adv_temp, dim=-1).detach() * neg_loss, dim=-1)
        else:
            neg_loss = th.mean(neg_loss, dim=-1)
        loss = (th.mean(neg_loss) + th.mean(pos_loss)) / 2
        reg = 0.0
        if self.reg_coef > 0 and self.reg_norm > 0:
            q = self.reg_norm
            regt = self.reg_coef * (th.cat([pos_emb, neg], 0).norm(p=q) ** q + rel.norm(p=q) ** q)
            reg = regt.item()
            loss = loss + regt
        log = (th.mean(pos_loss).item(), th.mean(neg_loss).item(), loss.item() - reg, reg)
        loss.backward()
        with th.no_grad():
            for table, state, idx, g in ((self.ent, self.ent_state, nid, pos_emb.grad),
                                        (self.ent, self.ent_state, gid, neg.grad),
                                        (self.rel, self.rel_state, rid, rel.grad)):
                gs = (g * g).mean(1)
                state.index_add_(0, idx, gs)
                std = state[idx].sqrt_().add_(1e-10).unsqueeze(1)
                table.index_add_(0, idx, -self.lr * g / std)
        return dict(pos_score=pos.detach().numpy(), neg_score=ns.detach().numpy().reshape(C, chunk, N),
                    log=log, g_pos_ent=pos_emb.grad.numpy(), g_neg=neg.grad.numpy(),
                    g_rel=rel.grad.numpy())


# ---------------------------------------------------------------------------------------------
# the reference's multi-process CPU mode (`--num_proc P`, train.py:298-317): P trainer processes, one
# thread each, lock-free on ONE table in shared memory (ExternalEmbedding.share_memory, tensor_models.py:
# 233-236).  Used only by bench.py's cpu_baseline leg.
# ---------------------------------------------------------------------------------------------
def _hogwild_worker(rank, tables, w, seconds, ready, go, out):
    import time
    from oracle import kge_oracle as O
    th.set_num_threads(1)
    port = TorchPort(w["model"], 1, 1, w["hidden"], w["gamma"], w["lr"], w["de"], w["dr"], w["adv"], w["adv_temp"],
                     w["reg_coef"], w["reg_norm"])
    port.ent, port.ent_state, port.rel, port.rel_state = tables
    rng = np.random.RandomState(1000 + rank)
    plans = []
    for s in range(1, 13):
        bt = O.synth_batch(rng, w["n_ent"], w["n_rel"], w["B"], w["N"], w["N"], s)
        plans.append(dict(nid=bt["nid"], h_local=bt["h_local"], t_local=bt["t_local"], rel_ids=bt["r"], neg_ids=bt["neg"],
                          C=w["B"] // w["N"], chunk=w["N"], N=w["N"], neg_head=bt["neg_head"]))
    port.step(plans[0])
    ready.put(rank)
    go.wait()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        port.step(plans[n % len(plans)])
        n += 1
    out.put((rank, n, time.perf_counter() - t0))


def hogwild_cpu(w, procs, seconds=5.0, timeout=120.0):
    """aggregate edges/s of `procs` single-thread trainer processes sharing the tables; returns (edges/s, steps)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    base = TorchPort(w["model"], w["n_ent"], w["n_rel"], w["hidden"], w["gamma"], w["lr"], w["de"], w["dr"], w["adv"],
                     w["adv_temp"], w["reg_coef"], w["reg_norm"])
    tables = tuple(t.share_memory_() for t in (base.ent, base.ent_state, base.rel, base.rel_state))
    ready, out, go = ctx.Queue(), ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=_hogwild_worker, args=(r, tables, w, seconds, ready, go, out), daemon=True) for r in range(procs)]
    for p in ps:
        p.start()
    try:
        for _ in range(procs):
            ready.get(timeout=timeout)
        go.set()
        res = [out.get(timeout=seconds + timeout) for _ in range(procs)]
    finally:
        for p in ps:
            p.join(timeout=5)
            if p.is_alive():
                p.kill()          # our own children, by handle
    steps = sum(n for _, n, _ in res)
    wall = max(dt for _, _, dt in res)
    return steps * w["B"] / wall, steps
