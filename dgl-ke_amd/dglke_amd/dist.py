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

    def apply_packed(self, table, state, idx, msg, ntraces, lr):
        """owner-side Adagrad of packed messages (kge_adagrad_apply_packed)."""
        _lib.check(_lib.lib().kge_adagrad_apply_packed(
            _lib.ptr(table), _lib.ptr(state), table.shape[0], table.shape[1],
            _lib.ptr(idx) if idx is not None else None, _lib.ptr(msg), msg.shape[1], msg.shape[0],
            ntraces, float(lr), 1e-10, _lib.stream_ptr()))

    def reset_rel_msg(self, rel_msg, d_r):
        rel_msg[:, d_r] = 0
        rel_msg.view(torch.int32)[:, d_r + 1:d_r + 3] = -1

    def step_grads(self, engine, batch, cache, ent_msg, rel_msg, zero_state):
        """run kge_step_grads against the row cache, emitting packed messages:
        ent_msg[u] = [g0 | g1 | gs0 gs1 . .],  rel_msg[u] = [gr | gsr | id_lo id_hi .]"""
        d_e, d_r = cache.shape[1], engine.rel.shape[1]
        tb = _lib.KgeTables()
        tb.ent, tb.ent_state = _lib.ptr(cache), _lib.ptr(zero_state)
        tb.rel, tb.rel_state = _lib.ptr(engine.rel), _lib.ptr(engine.rel_state)
        tb.n_ent, tb.n_rel = cache.shape[0], engine.rel.shape[0]
        em = _lib.KgeEmit()
        e0, r0 = ent_msg.data_ptr(), rel_msg.data_ptr()
        em.g0, em.g1 = e0, e0 + 4 * d_e
        em.gs0, em.gs1 = e0 + 8 * d_e, e0 + 8 * d_e + 4
        em.gr, em.gsr, em.rid = r0, r0 + 4 * d_r, r0 + 4 * d_r + 4
        em.ld_e, em.ld_r = ent_msg.shape[1], rel_msg.shape[1]
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
        self._frozen = False
        self.max_rows = 0

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
        """persistent scratch: allocated once for the largest row count seen so far, returned as a
        [rows, ...] view (so that a captured graph never allocates)."""
        dtype = dtype or self.ent.dtype
        rows = shape[0]
        t = self._bufs.get(name)
        if t is None or t.shape[0] < rows or tuple(t.shape[1:]) != tuple(shape[1:]) or t.dtype != dtype:
            if self._frozen:
                raise _lib.KgeError("DistEngine buffer %s would be re-allocated after graph capture" % name)
            cap = max(rows, self.max_rows)
            t = torch.zeros((cap,) + tuple(shape[1:]), dtype=dtype, device=self.dev)
            self._bufs[name] = t
        return t[:rows]

    # ---- steps 2-5 -----------------------------------------------------------------------
    def step(self, batch, route):
        sp, ops = self.spec, self.ops
        UE, nrecv, B = route.UE, route.n_recv, batch.B
        ld_e, ld_r = 2 * self.d_e + 4, self.d_r + 4
        # 2. pull: owners gather the requested rows, one all-to-all returns them into the row cache
        rows_out = ops.gather(self.ent, route.recv_ids_local)
        cache = self._buf("cache", (UE, self.d_e))
        dist.all_to_all_single(cache, rows_out, route.send_counts, route.recv_counts, group=self.group)
        # 3. local compute against the cache -> one packed message per unique entity / relation
        ent_msg = self._buf("ent_msg", (UE, ld_e))
        rel_msg = self._buf("rel_msg", (B, ld_r))
        ops.reset_rel_msg(rel_msg, self.d_r)       # padding rows: id = -1, increment = 0
        ops.step_grads(self.engine, batch, cache, ent_msg, rel_msg, self._buf("zero_state", (UE,)))
        # 4. push: ONE all-to-all carries both entity traces; owners apply per source rank, in order
        recv_msg = self._buf("recv_msg", (nrecv, ld_e))
        dist.all_to_all_single(recv_msg, ent_msg, route.recv_counts, route.send_counts, group=self.group)
        off = 0
        for src in range(sp.world):
            n = route.recv_counts[src]
            if n:
                ops.apply_packed(self.ent, self.ent_state, route.recv_ids_local[off:off + n],
                                 recv_msg[off:off + n], 2, self.lr)
            off += n
        # 5. relations: ONE all-gather of the packed messages (ids inside); every rank applies all
        #    ranks' updates in rank order -> replicas stay bit-identical
        all_rel = self._buf("all_rel", (sp.world * B, ld_r))
        dist.all_gather_into_tensor(all_rel.view(-1), rel_msg.reshape(-1), group=self.group)
        for src in range(sp.world):
            ops.apply_packed(self.engine.rel, self.engine.rel_state, None, all_rel[src * B:(src + 1) * B], 1, self.lr)

    def capture(self, batch, route, stream=None):
        """record one sharded step (kernels + RCCL collectives) into a HIP graph."""
        self._frozen = True
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            self.step(batch, route)
        return g
