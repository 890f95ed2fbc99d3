"""Range-sharded multi-GPU training step: one process per GPU, entity table +
Adagrad state sharded by contiguous id range, relation table replicated, RCCL all-to-all over xGMI
for the row pull and the gradient push.

This replaces the reference's parameter-server path (`KEModel.pull_model` / `push_gradient`,
models/general_models.py:650-680; server-side Adagrad `kvserver.py:41-51`) - same semantics (pull the
unique rows of the batch, compute locally, push per-row gradients, the OWNER applies Adagrad) -
with collectives instead of DGL's TCP KVStore.  The reference has no collective to translate; the
all-to-all is chosen because xGMI is point-to-point (a ring would be bound by one link).

Per step and rank:
  1. ids:   the sorted unique entity ids of the batch are split by owner (contiguous slices);
            counts + ids travel by `all_to_all_single` (`prepare_route`, can run steps ahead);
  2. pull:  owners gather the requested rows (kge_gather_rows) -> all-to-all -> row cache [UE, D];
  3. step:  `kge_step_grads` = the SAME kernels as the single-GPU step, run against the cache
            (batch ids remapped to cache rows); emits per-row trace-0 / trace-1 gradients and
            Adagrad increments instead of updating the entity table;
  4. push:  reverse all-to-all of the gradients; owners apply them in source-rank order with
            kge_adagrad_apply_rows (trace 0 then trace 1, like tensor_models.py:316);
  5. rels:  summed relation gradients are all-gathered and EVERY rank applies all of them in rank
            order, so the replicated relation tables stay bit-identical.

The routing (this file) is device-agnostic torch + torch.distributed code; the arithmetic is in
`HipOps` (libkge_hip).  CPU tests exercise the routing under gloo with a stand-in ops object
(tests/test_dist_gloo.py); the product path always uses HipOps and refuses host tensors.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, plan as _plan


class ShardSpec(object):
    def __init__(self, n_entities, world, rank):
        self.n_entities, self.world, self.rank = int(n_entities), int(world), int(rank)
        self.shard = (self.n_entities + self.world - 1) // self.world
        self.lo = min(self.rank * self.shard, self.n_entities)
        self.hi = min(self.lo + self.shard, self.n_entities)
        self.n_local = self.hi - self.lo

    def bounds(self):
        return np.minimum(np.arange(self.world + 1, dtype=np.int64) * self.shard, self.n_entities)


class Route(object):
    """who sends what for one batch: send = what this rank requests from each owner,
    recv = what this rank owns and the others request."""
    __slots__ = ("send_counts", "recv_counts", "recv_ids_local", "UE", "n_recv")


class HipOps(object):
    """the arithmetic of the sharded step, all in libkge_hip."""

    def gather(self, table, idx):
        from . import ops
        return ops.gather_rows(table, idx)

    def apply_rows(self, table, state, idx, g, gs, lr):
        from . import ops
        ops.adagrad_apply_rows(table, state, idx, g, gs, lr)

    def step_grads(self, engine, batch, cache, emit_bufs):
        """run kge_step_grads against the row cache."""
        dev = cache.device
        tb = _lib.KgeTables()
        zero_state = emit_bufs["zero_state"]
        tb.ent, tb.ent_state = _lib.ptr(cache), _lib.ptr(zero_state)
        tb.rel, tb.rel_state = _lib.ptr(engine.rel), _lib.ptr(engine.rel_state)
        tb.n_ent, tb.n_rel = cache.shape[0], engine.rel.shape[0]
        em = _lib.KgeEmit()
        for k in ("g0", "gs0", "g1", "gs1", "gr", "gsr"):
            setattr(em, k, _lib.ptr(emit_bufs[k]))
        out = _lib.KgeStepOut()
        out.loss_accum = _lib.ptr(engine.loss_accum)
        ws = engine.workspace_for(batch)
        _lib.check(_lib.lib().kge_step_grads(C.byref(engine.hp), C.byref(tb), C.byref(batch.c),
                                             C.byref(out), C.byref(em), _lib.ptr(ws),
                                             engine._ws_bytes, _lib.stream_ptr()))


def localize_plan(h, t, r, neg, chunk, N, neg_head, edge_w=None):
    """global-id batch -> (sorted unique entity ids, plan over cache-local row indices)."""
    h = np.asarray(h, np.int64)
    t = np.asarray(t, np.int64)
    neg = np.asarray(neg, np.int64)
    ue = np.unique(np.concatenate([h, t, neg]))
    p = _plan.build_plan(np.searchsorted(ue, h), np.searchsorted(ue, t), r,
                         np.searchsorted(ue, neg), chunk, N, neg_head, edge_w)
    assert p["UE"] == ue.shape[0]
    return ue, p


class DistEngine(object):
    """sharded counterpart of StepEngine.  `engine` is a StepEngine-like object that owns the
    hyper-parameters, the replicated relation table and the workspace; `ent`/`ent_state` are this
    rank's shard."""

    def __init__(self, engine, spec, ent_shard, ent_state_shard, ops=None, group=None):
        self.engine = engine
        self.spec = spec
        self.ent = ent_shard
        self.ent_state = ent_state_shard
        self.ops = ops or HipOps()
        self.group = group
        self.dev = ent_shard.device
        self.lr = float(engine.hp.lr) if hasattr(engine, "hp") else float(engine.lr)
        self.d_e = ent_shard.shape[1]
        self.d_r = engine.rel.shape[1]
        self._bufs = {}

    # ---- step 1: ids ---------------------------------------------------------------------
    def prepare_route(self, ue_ids_np):
        """exchange counts and ids for one batch (host-known sorted unique ids)."""
        sp = self.spec
        cuts = np.searchsorted(ue_ids_np, sp.bounds())
        send_counts = np.diff(cuts).astype(np.int64)
        sc = torch.from_numpy(send_counts).to(self.dev)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)
        recv_counts = rc.cpu().numpy()
        ids = torch.from_numpy(np.ascontiguousarray(ue_ids_np)).to(self.dev)
        recv = torch.empty(int(recv_counts.sum()), dtype=torch.int64, device=self.dev)
        dist.all_to_all_single(recv, ids, recv_counts.tolist(), send_counts.tolist(), group=self.group)
        rt = Route()
        rt.send_counts, rt.recv_counts = send_counts.tolist(), recv_counts.tolist()
        rt.recv_ids_local = recv - sp.lo
        rt.UE, rt.n_recv = int(ue_ids_np.shape[0]), int(recv.shape[0])
        return rt

    def _buf(self, name, shape, dtype=None):
        dtype = dtype or self.ent.dtype
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.zeros(shape, dtype=dtype, device=self.dev)
            self._bufs[name] = t
        return t

    # ---- steps 2-5 -----------------------------------------------------------------------
    def step(self, batch, route):
        sp, ops = self.spec, self.ops
        UE, nrecv = route.UE, route.n_recv
        B = batch.B
        # 2. pull
        rows_out = ops.gather(self.ent, route.recv_ids_local)
        cache = self._buf("cache", (UE, self.d_e))
        dist.all_to_all_single(cache, rows_out, route.send_counts, route.recv_counts, group=self.group)
        # 3. local compute against the cache
        em = dict(g0=self._buf("g0", (UE, self.d_e)), gs0=self._buf("gs0", (UE,)),
                  g1=self._buf("g1", (UE, self.d_e)), gs1=self._buf("gs1", (UE,)),
                  gr=self._buf("gr", (B, self.d_r)), gsr=self._buf("gsr", (B,)),
                  zero_state=self._buf("zero_state", (UE,)))
        ops.step_grads(self.engine, batch, cache, em)
        # 4. push entity gradients to their owners and apply in source-rank order
        r0 = self._buf("r_g0", (nrecv, self.d_e))
        r1 = self._buf("r_g1", (nrecv, self.d_e))
        gs = torch.stack([em["gs0"], em["gs1"]], dim=1).contiguous()
        rgs = self._buf("r_gs", (nrecv, 2))
        dist.all_to_all_single(r0, em["g0"], route.recv_counts, route.send_counts, group=self.group)
        dist.all_to_all_single(r1, em["g1"], route.recv_counts, route.send_counts, group=self.group)
        dist.all_to_all_single(rgs, gs, route.recv_counts, route.send_counts, group=self.group)
        off = 0
        for src in range(sp.world):
            n = route.recv_counts[src]
            if n:
                sl = slice(off, off + n)
                ids = route.recv_ids_local[sl]
                ops.apply_rows(self.ent, self.ent_state, ids, r0[sl], rgs[sl, 0].contiguous(), self.lr)
                ops.apply_rows(self.ent, self.ent_state, ids, r1[sl], rgs[sl, 1].contiguous(), self.lr)
            off += n
        # 5. relations: all-gather (padded to B rows) and apply everything in rank order
        ur = self._buf("ur_pad", (B,), torch.int64)
        ur.fill_(-1)
        ur[:batch.UR] = batch.view("ur_id")
        if batch.UR < B:
            em["gsr"][batch.UR:] = 0
        all_ur = self._buf("all_ur", (sp.world, B), torch.int64)
        all_gr = self._buf("all_gr", (sp.world, B, self.d_r))
        all_gsr = self._buf("all_gsr", (sp.world, B))
        dist.all_gather_into_tensor(all_ur.view(-1), ur, group=self.group)
        dist.all_gather_into_tensor(all_gr.view(-1), em["gr"].view(-1), group=self.group)
        dist.all_gather_into_tensor(all_gsr.view(-1), em["gsr"], group=self.group)
        for src in range(sp.world):
            ops.apply_rows(self.engine.rel, self.engine.rel_state, all_ur[src], all_gr[src],
                           all_gsr[src], self.lr)
