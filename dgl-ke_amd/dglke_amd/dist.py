"""Range-sharded multi-GPU training step: one process per GPU, entity table +
Adagrad state sharded by contiguous id range, relation table replicated, RCCL all-to-all over xGMI
for the row pull and the gradient push.

This replaces the reference's parameter-server path (`KEModel.pull_model` / `push_gradient`,
models/general_models.py:650-680; server-side Adagrad `kvserver.py:41-51`) - same semantics (pull the
unique rows of the batch, compute locally, push per-row gradients, the OWNER applies Adagrad) -
with collectives instead of DGL's TCP KVStore.  The reference has no collective to translate; the
all-to-all is chosen because xGMI is point-to-point (a ring would be bound by one link).

Round 3: the routing lives ON THE DEVICE and every message has a FIXED size, so a step never touches the
host (no `.cpu()`, no numpy, no count exchange) and all collectives are equal-split:

  route   `kge_route_build` cuts the batch's sorted unique entity ids (the plan the sampler kernel wrote) into one
          bucket of `cap` ids per owner (-1 padded) and re-addresses the batch to CACHE ROWS (owner * cap + position);
  pull    all-to-all of the request ids -> owners gather (`kge_gather_rows_req`) -> all-to-all of the rows: the
          result IS the row cache, laid out [world][cap][D];
  step    `kge_step_grads` = the kernels of the single-GPU step against the cache; the update kernel emits ONE packed
          message per unique row at the row's cache position ([g0 | g1 | gs0 gs1 . .]);
  push    reverse all-to-all of the messages; `kge_adagrad_apply_merged` applies them on the owner: the messages of all
          source ranks in ONE launch, every row by the wavefront of its first source, in source-rank order;
  rels    packed relation gradients (ids inside) are all-gathered and EVERY rank applies all of them with the same
          merged kernel, so the replicated relation tables stay bit-identical.

Round 5: the request ids of a whole sampled GROUP are exchanged once (`prepare_group`): a step is two exchanges (rows, gradients);
and a group - routing, id exchange, steps WITH their collectives - replays from one hipGraph (`run_group`; RcclComm's calls can be
recorded - what looked like a replay hang in rounds 2-4 was ncclCommDestroy waiting for live graphs, `close()`).

`step_pipelined` overlaps the pull of step s+1 (side stream) with the compute of step s under the staleness the
reference's `--async_update` licenses (tensor_models.py:136-175): the rows of step s+1 are gathered after update
s-1 and before update s has landed - exact one-step staleness, deterministic.

The routing arithmetic is in `HipOps` (libkge_hip); the collectives in `RcclComm` (librccl called directly) / `TorchComm`
(torch.distributed: backend nccl = RCCL) / `HostStagedComm` (ranks sharing a GPU).
CPU tests run the same `DistEngine` under gloo with numpy stand-ins (tests/test_dist_gloo.py); GPU tests run HipOps at
world 2 with two processes on one device (tests/test_gpu_dist.py).  The product path uses HipOps and refuses host tensors.
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


def default_cap(ue_max, world, slack=1.5):
    """rows per (rank, owner) bucket: the mean share of a batch's unique entities plus slack, a multiple of 64.  Uniform ids
    put ~ue_max/world +- sqrt(ue_max/world) ids into a bucket; heavy-tailed graphs need more slack (the overflow counter of
    kge_route_build tells)."""
    if world == 1:
        return int(ue_max)
    return int(min(ue_max, (int(ue_max / world * slack) + 64 + 63) // 64 * 64))


class TorchComm(object):
    """equal-split collectives on device tensors (backend nccl = RCCL over xGMI)."""
    capturable = False          # (kept eager: the c10d wrappers are the fallback transport)

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def all_to_all(self, out, inp):
        dist.all_to_all_single(out, inp, group=self.group)

    def all_gather(self, out, inp):
        dist.all_gather_into_tensor(out, inp, group=self.group)

    def push_pair(self, a2a_out, a2a_in, ag_out, ag_in):
        self.all_to_all(a2a_out, a2a_in)
        self.all_gather(ag_out, ag_in)


class RcclComm(object):
    """the same collectives called on librccl directly (ctypes), on the CURRENT stream: no c10d work objects, no internal
    communication stream with its event hand-offs - a few microseconds of host time per call instead of ~40, and the two
    collectives of the push leave as ONE grouped launch.  The communicator is this class's own (ncclCommInitRank with an id
    broadcast over the torch.distributed group); byte counts, so any dtype."""
    _INT8 = 0                                       # ncclInt8
    # its calls may be recorded into a hipGraph (round 5: captured RCCL collectives replay fine on ROCm 7.0 / RCCL 2.26 - what
    # rounds 2-4 took for a replay hang was ncclCommDestroy waiting for the graphs that still referenced the communicator, see close())
    capturable = True

    class _Uid(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    def __init__(self, group=None):
        import os
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        L = self._L = C.CDLL(path)
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetErrorString.argtypes = [C.c_int]
        L.ncclGetUniqueId.argtypes = [C.POINTER(self._Uid)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, self._Uid, C.c_int]
        L.ncclAllToAll.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        uid = self._Uid()
        if self.rank == 0:
            self._ck(L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if self.world > 1:
            box = [bytes(bytearray(uid))]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            C.memmove(C.byref(uid), box[0], 128)
        self._comm = C.c_void_p()
        self._ck(L.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def _ck(self, rc, what):
        if rc != 0:
            raise _lib.KgeError("%s failed: %s" % (what, self._L.ncclGetErrorString(rc).decode()))

    @staticmethod
    def _bytes(t):
        if not t.is_contiguous():
            raise ValueError("RcclComm: contiguous tensors only")
        return t.numel() * t.element_size()

    def all_to_all(self, out, inp):
        n = self._bytes(inp)
        if n % self.world or self._bytes(out) != n:
            raise ValueError("all_to_all: %d bytes over %d ranks" % (n, self.world))
        self._ck(self._L.ncclAllToAll(inp.data_ptr(), out.data_ptr(), n // self.world, self._INT8, self._comm,
                                      _lib.stream_ptr()), "ncclAllToAll")

    def all_gather(self, out, inp):
        n = self._bytes(inp)
        if self._bytes(out) != n * self.world:
            raise ValueError("all_gather: %d bytes in, %d out" % (n, self._bytes(out)))
        self._ck(self._L.ncclAllGather(inp.data_ptr(), out.data_ptr(), n, self._INT8, self._comm, _lib.stream_ptr()),
                 "ncclAllGather")

    def push_pair(self, a2a_out, a2a_in, ag_out, ag_in):
        """the gradient all-to-all and the relation all-gather of one step as one grouped launch"""
        self._ck(self._L.ncclGroupStart(), "ncclGroupStart")
        try:
            self.all_to_all(a2a_out, a2a_in)
            self.all_gather(ag_out, ag_in)
        finally:
            self._ck(self._L.ncclGroupEnd(), "ncclGroupEnd")

    def close(self):
        """ncclCommDestroy BLOCKS for as long as a hipGraph that recorded one of this communicator's collectives is alive (the graph
        holds a reference on the communicator's persistent plans): destroy such graphs first (DistEngine.close_graphs())."""
        if self._comm:
            self._L.ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()


class HostStagedComm(object):
    """the same collectives carried by a host-side (gloo) process group through host copies.  Only for ranks that SHARE a
    device (`--gpu 0 0`: how the multi-process path is exercised on a one-GPU box - RCCL refuses two ranks on one GPU); the
    step synchronises the stream around every exchange."""

    capturable = False          # host copies

    def __init__(self, group=None):
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def all_to_all(self, out, inp):
        torch.cuda.current_stream().synchronize()
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu().contiguous(), group=self.group)
        out.copy_(o)

    def all_gather(self, out, inp):
        torch.cuda.current_stream().synchronize()
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu().contiguous(), group=self.group)
        out.copy_(o)


def make_comm(group=None, kind=None):
    """communicator of the sharded step: librccl called directly (default; KGE_DIST_COMM=torch or any failure to set it up on
    ANY rank: the c10d wrappers).  Every rank of the group must call it."""
    import os
    kind = kind or os.environ.get("KGE_DIST_COMM", "rccl")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    comm, ok = None, 1
    if kind == "rccl" and torch.cuda.is_available():
        try:
            comm = RcclComm(group)
        except Exception as e:           # noqa: BLE001 - anything: fall back together
            ok = 0
            import sys
            print("RcclComm unavailable (%r): using torch.distributed collectives" % (e,), file=sys.stderr)
    else:
        ok = 0
    if world > 1 and kind == "rccl" and torch.cuda.is_available():
        flag = torch.tensor([ok], device="cpu" if dist.get_backend(group) == "gloo" else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0 and comm is not None:
            comm.close()
            comm = None
    if comm is not None:
        return comm
    # the c10d wrappers need a backend that moves device tensors (nccl = RCCL); a host-side (gloo) group - the CLI's - carries the
    # messages through host copies instead (gloo's all_to_all takes CPU tensors only)
    if world > 1 and dist.get_backend(group) == "gloo":
        return HostStagedComm(group)
    return TorchComm(group)


class LocalBatch(object):
    """a batch re-addressed to cache rows (duck type of plan.Batch / DeviceBatch for StepEngine)."""

    def __init__(self, batch, c, keep):
        self.B, self.C, self.chunk, self.N = batch.B, batch.C, batch.chunk, batch.N
        self.U, self.UE, self.UR = batch.U, batch.UE, batch.UR
        self.neg_head = batch.neg_head
        self.c = c
        self._keep = (batch, keep)


class HipOps(object):
    """the device arithmetic of the sharded step, all in libkge_hip."""
    # round 6 (ABI 8): PACKED single-trace entity messages - a row's message carries ONE trace ([g | gs | link], d_e + 4 floats
    # instead of 2 d_e + 4); the rare row that is in both traces of a batch puts its negative trace into a small extra region of
    # its owner bucket (cap2 rows, sized per group like the buckets themselves).  The push all-to-all moves
    # (cap + cap2) (d_e + 4) floats per peer instead of cap (2 d_e + 4).  KGE_DIST_PACKED=0: the two-trace messages.
    import os as _os
    packed_messages = _os.environ.get("KGE_DIST_PACKED", "1") != "0"

    def route(self, batch, world, per, cap, bf, cap2=0):
        """fill bf.req_ids and the cache-row id arrays; return the re-addressed batch."""
        L = _lib.lib()
        _lib.check(L.kge_route_build(C.byref(batch.c), world, per, cap, _lib.ptr(bf.req_ids), _lib.ptr(bf.h_loc),
                                     _lib.ptr(bf.t_loc), _lib.ptr(bf.neg_loc), _lib.ptr(bf.ue_loc), _lib.ptr(bf.ue_rec_loc),
                                     _lib.ptr(bf.overflow), int(cap2), _lib.ptr(bf.ue_msg) if cap2 else None, _lib.stream_ptr()))
        kb = _lib.KgeBatch()
        _lib.check(L.kge_batch_localized(C.byref(batch.c), _lib.ptr(bf.h_loc), _lib.ptr(bf.t_loc), _lib.ptr(bf.neg_loc),
                                         _lib.ptr(bf.ue_loc), _lib.ptr(bf.ue_rec_loc), C.byref(kb)))
        lb = LocalBatch(batch, kb, bf)
        lb.req_ids = bf.req_ids
        lb.msg_rows = bf.ue_msg.data_ptr() if cap2 else 0
        return lb

    @staticmethod
    def route_layout(batch, world, cap):
        """byte offsets of one batch's routing outputs inside a pool row (32-byte aligned) and the row stride"""
        al = lambda x: (x + 31) & ~31
        o, off = 0, {}
        for name, nbytes in (("req_ids", 8 * world * cap), ("h_loc", 8 * batch.B), ("t_loc", 8 * batch.B),
                             ("neg_loc", 8 * batch.C * batch.N), ("ue_loc", 8 * batch.UE), ("ue_rec_loc", 32 * batch.UE),
                             ("ue_msg", 8 * batch.UE)):          # (packed messages: message rows per union entry)
            off[name] = o
            o = al(o + nbytes)
        return off, al(o)

    def route_group(self, batches, world, per, cap, pool, off, stride, overflow, cap2=0):
        """kge_route_build for a GROUP of consecutive sampler slots in ONE launch: batch k's outputs go to pool row (its sampler
        slot) - see routed_batch for the re-addressed batch of a slot."""
        L = _lib.lib()
        b0 = batches[0]
        base = pool.data_ptr() + b0.slot * stride
        _lib.check(L.kge_route_build_group(C.byref(b0.c), len(batches), b0.sampler.slot_bytes, world, per, cap,
                                           base + off["req_ids"], base + off["h_loc"], base + off["t_loc"], base + off["neg_loc"],
                                           base + off["ue_loc"], base + off["ue_rec_loc"], stride, _lib.ptr(overflow),
                                           int(cap2), (base + off["ue_msg"]) if cap2 else None, _lib.stream_ptr()))

    def routed_batch(self, b, world, cap, pool, off, stride):
        """the re-addressed batch of sampler slot b.slot inside the route pool (pure pointer arithmetic: built once per slot and
        corruption mode, then reused - the eager multi-GPU step is host-bound, 15 us of ctypes per batch showed)"""
        row = pool.data_ptr() + b.slot * stride
        kb = _lib.KgeBatch()
        _lib.check(_lib.lib().kge_batch_localized(C.byref(b.c), row + off["h_loc"], row + off["t_loc"], row + off["neg_loc"],
                                                  row + off["ue_loc"], row + off["ue_rec_loc"], C.byref(kb)))
        lb = LocalBatch(b, kb, pool)
        o0 = b.slot * stride + off["req_ids"]
        lb.req_ids = pool[o0:o0 + 8 * world * cap].view(torch.int64)
        lb.msg_rows = row + off["ue_msg"]            # (meaningful when the group was routed with cap2 > 0)
        return lb

    def route_fill(self, batches, world, per, out):
        """out[0] = max(out[0], the largest owner-bucket fill over `batches`), out[1] likewise for the entries of one bucket that are
        in BOTH traces (the packed messages' extra region): ONE launch when the batches are consecutive slots of a device sampler,
        one per batch otherwise (host-built plans)."""
        L = _lib.lib()
        b0 = batches[0]
        smp = getattr(b0, "sampler", None)
        slots = [getattr(b, "slot", None) for b in batches]
        if (smp is not None and len(batches) > 1 and all(getattr(b, "sampler", None) is smp for b in batches) and
                slots == list(range(slots[0], slots[0] + len(slots)))):
            _lib.check(L.kge_route_fill(C.byref(b0.c), len(batches), smp.slot_bytes, world, per, _lib.ptr(out), _lib.stream_ptr()))
            return
        for b in batches:
            _lib.check(L.kge_route_fill(C.byref(b.c), 1, 0, world, per, _lib.ptr(out), _lib.stream_ptr()))

    def gather_req(self, table, ids, lo, out):
        _lib.check(_lib.lib().kge_gather_rows_req(_lib.ptr(table), table.shape[0], table.shape[1], _lib.ptr(ids), int(lo),
                                                  ids.shape[0], _lib.ptr(out), _lib.stream_ptr()))

    def apply_merged(self, table, state, nsrc, cap, ids, lo, msg, ntraces, lr, cap_extra=0):
        """ids: int64 [nsrc * cap] (global ids, -1 pads) or None: the ids sit inside the messages behind the increments.
        cap_extra > 0: packed single-trace messages, cap + cap_extra message rows per source (ntraces = 1)"""
        dim, ld = table.shape[1], msg.shape[1]
        if ids is not None:
            idw, stride = ids.data_ptr(), 2
        else:
            idw, stride = msg.data_ptr() + 4 * (ntraces * dim + ntraces), ld
        _lib.check(_lib.lib().kge_adagrad_apply_merged(_lib.ptr(table), _lib.ptr(state), table.shape[0], dim, nsrc, cap, idw,
                                                       stride, int(lo), _lib.ptr(msg), ld, ntraces, int(cap_extra), float(lr), 1e-10,
                                                       _lib.stream_ptr()))

    @staticmethod
    def _job(table, state, nsrc, cap, ids, lo, msg, ntraces, cap_extra=0):
        dim, ld = table.shape[1], msg.shape[1]
        j = _lib.KgeMergeJob()
        j.table, j.state_sum, j.n_rows = _lib.ptr(table), _lib.ptr(state), table.shape[0]
        j.dim, j.nsrc, j.cap, j.ld, j.ntraces, j.cap_extra = dim, nsrc, cap, ld, ntraces, int(cap_extra)
        if ids is not None:
            j.id_words, j.id_stride_words = ids.data_ptr(), 2
        else:
            j.id_words, j.id_stride_words = msg.data_ptr() + 4 * (ntraces * dim + ntraces), ld
        j.id_offset, j.msg = int(lo), _lib.ptr(msg)
        return j

    def apply_merged_pair(self, job_a, job_b, lr):
        """two apply_merged jobs (tuples of its arguments without lr) in ONE launch: the entity-shard and the relation-replica apply"""
        # (the two argument structs of a given buffer set are built once: the eager multi-GPU step is host-bound.  The key names
        #  everything the structs hold EXCEPT the id arrays, which change with every step of a group - they are set per call)
        key = tuple((j[0].data_ptr(), j[1].data_ptr(), j[6].data_ptr(), j[2], j[3], int(j[5]), j[7], (j[8] if len(j) > 8 else 0))
                    for j in (job_a, job_b))
        if not hasattr(self, "_jobs"):
            self._jobs = {}
        st = self._jobs.get(key)
        if st is None:
            if len(self._jobs) > 64:
                self._jobs.clear()
            st = self._jobs[key] = (self._job(*job_a), self._job(*job_b))
        for j, args in zip(st, (job_a, job_b)):
            if args[4] is not None:
                j.id_words = args[4].data_ptr()
        _lib.check(_lib.lib().kge_adagrad_apply_merged_pair(C.byref(st[0]), C.byref(st[1]), float(lr), 1e-10, _lib.stream_ptr()))

    def reset_rel_pads(self, rel_msg, d_r, first):
        """host-built plans: message rows >= the batch's unique-relation count are pads (device-built plans: the kernel writes them)"""
        if first < rel_msg.shape[0]:
            rel_msg.view(torch.int32)[first:, d_r + 1:d_r + 3] = -1

    def step_local(self, engine, batch, ent, ent_state):
        """world 1 without collectives: every row of the batch is a row of THIS rank's shard (global id == shard row), so the step is
        the in-place single-table step on the shard - no routing, no row cache, no gradient messages, no apply launch (round 6:
        the cache copy and the messages were 66 MB of the a2a engine's 277 MB per step at world 1)."""
        key = ("local", ent.data_ptr(), ent_state.data_ptr(), engine.rel.data_ptr())
        if not hasattr(self, "_structs"):
            self._structs = {}
        st = self._structs.get(key)
        if st is None:
            tb = _lib.KgeTables()
            tb.ent, tb.ent_state = _lib.ptr(ent), _lib.ptr(ent_state)
            tb.rel, tb.rel_state = _lib.ptr(engine.rel), _lib.ptr(engine.rel_state)
            tb.n_ent, tb.n_rel = ent.shape[0], engine.rel.shape[0]
            out = _lib.KgeStepOut()
            out.loss_accum = _lib.ptr(engine.loss_accum)
            out.tickets = _lib.ptr(engine.tickets)
            st = self._structs[key] = (tb, out)
        tb, out = st
        ws = engine.workspace_for(batch)
        _lib.check(_lib.lib().kge_step_fused(C.byref(engine.hp), C.byref(tb), C.byref(batch.c), C.byref(out), _lib.ptr(ws),
                                             engine._ws_bytes, _lib.stream_ptr()))

    def step_grads(self, engine, lb, cache, ent_msg, rel_msg, zero_state):
        """run kge_step_grads against the row cache, emitting packed messages at the rows' cache positions:
        ent_msg[row] = [g0 | g1 | gs0 gs1 . .],  rel_msg[u] = [gr | gsr | id_lo id_hi .]"""
        # rel_msg None: no relation messages - the step applies the relation trace to engine.rel itself (relation partitioning:
        # every relation row of the batch belongs to this rank)
        key = (cache.data_ptr(), ent_msg.data_ptr(), rel_msg.data_ptr() if rel_msg is not None else 0, zero_state.data_ptr(),
               engine.rel.data_ptr())
        st = self._structs.get(key) if hasattr(self, "_structs") else None
        if st is None:                                   # the argument structs of a (cache, message buffers) pair: built once
            d_e, d_r = cache.shape[1], engine.rel.shape[1]
            tb = _lib.KgeTables()
            tb.ent, tb.ent_state = _lib.ptr(cache), _lib.ptr(zero_state)
            tb.rel, tb.rel_state = _lib.ptr(engine.rel), _lib.ptr(engine.rel_state)
            tb.n_ent, tb.n_rel = cache.shape[0], engine.rel.shape[0]
            if getattr(engine, "proj", None) is not None:      # TransR: the rank's own projection table (relation partitioning)
                tb.proj, tb.proj_state = _lib.ptr(engine.proj), _lib.ptr(engine.proj_state)
            em = _lib.KgeEmit()
            e0 = ent_msg.data_ptr()
            if ent_msg.shape[1] == d_e + 4:            # packed single-trace messages: [g | gs | link . .]; rows from lb.msg_rows
                em.g0 = e0
            else:
                em.g0, em.g1 = e0, e0 + 4 * d_e
                em.gs0, em.gs1 = e0 + 8 * d_e, e0 + 8 * d_e + 4
            em.ld_e = ent_msg.shape[1]
            if rel_msg is not None:
                r0 = rel_msg.data_ptr()
                em.gr, em.gsr, em.rid = r0, r0 + 4 * d_r, r0 + 4 * d_r + 4
                em.ld_r = rel_msg.shape[1]
            em.ent_by_id = 1
            out = _lib.KgeStepOut()
            out.loss_accum = _lib.ptr(engine.loss_accum)
            out.tickets = _lib.ptr(engine.tickets)
            if not hasattr(self, "_structs"):
                self._structs = {}
            st = self._structs[key] = (tb, em, out)
        tb, em, out = st
        if ent_msg.shape[1] == cache.shape[1] + 4:     # (per batch: the message rows of ITS routing; the bucket geometry of this buffer set)
            em.msg_rows, em.msg_cap, em.msg_cap_extra = lb.msg_rows, self._msg_cap[0], self._msg_cap[1]
        ws = engine.workspace_for(lb)
        _lib.check(_lib.lib().kge_step_grads(C.byref(engine.hp), C.byref(tb), C.byref(lb.c),
                                             C.byref(out), C.byref(em), _lib.ptr(ws),
                                             engine._ws_bytes, _lib.stream_ptr()))


class _Slot(object):
    """the buffers one in-flight pull needs (two slots: the pull of step s+1 runs while step s computes)."""
    pass


class DistEngine(object):
    """sharded counterpart of StepEngine.  `engine` is a StepEngine-like object that owns the
    hyper-parameters, the replicated relation table and the workspace; `ent`/`ent_state` are this
    rank's shard.  Batches carry GLOBAL entity ids and their plan (DeviceSampler slots or plan.upload)."""

    def __init__(self, engine, spec, ent_shard, ent_state_shard, ops=None, comm=None, cap=None, slack=1.5,
                 always_collective=False, rel_local=False, ue_bound=None):
        self.engine = engine
        self.spec = spec
        self.ent = ent_shard
        self.ent_state = ent_state_shard
        self.ops = ops or HipOps()
        self.comm = comm or TorchComm()
        if self.comm.world != spec.world:
            raise ValueError("communicator of %d ranks, shard spec of %d" % (self.comm.world, spec.world))
        self.dev = ent_shard.device
        self.lr = float(engine.hp.lr) if hasattr(engine, "hp") else float(engine.lr)
        self.d_e = ent_shard.shape[1]
        self.d_r = engine.rel.shape[1]
        self.cap, self.slack = cap, slack
        # ue_bound: the number of unique entities the exchange buffers are sized for, when the batches carry their EXACT count
        # (host-built plans; ADVICE r05: callers used to overwrite batch.UE with the bound).  None: the first batch's UE is the bound
        # (device-sampled batches, whose UE already is one)
        self.ue_bound = int(ue_bound) if ue_bound else None
        # packed single-trace entity messages (HipOps.packed_messages, round 6): cap2 = rows of a bucket's extra region (second
        # messages of the rows that are in both traces); 0 = two-trace messages (test doubles, KGE_DIST_PACKED=0)
        self.packed = bool(getattr(self.ops, "packed_messages", False))
        self.cap2 = 0
        # a single rank needs no collective (the buffers alias); always_collective keeps the calls (tests: the RCCL path at world 1)
        self.coll = spec.world > 1 or bool(always_collective)
        # rel_local: the training triples are PARTITIONED BY RELATION over the ranks (the reference's --rel_part,
        # dataloader/sampler.py:150-254, general_models.py:590-637 - the recipe of its multi-GPU Freebase runs): a relation's edges all
        # live on one rank, so its row is updated there and nowhere else - no relation exchange at all (no all-gather, no W-fold
        # apply); the owners' rows are collected when the tables are read (relation_rows_from_owners)
        self.rel_local = bool(rel_local)
        import os as _os
        # (KGE_DIST_REL_INPLACE=0: the relation trace as messages + an apply launch also under relation partitioning - A/B aid; ops
        #  without the capability - test doubles - say so with `rel_inplace = False`)
        self._rel_inplace = (self.rel_local and _os.environ.get("KGE_DIST_REL_INPLACE", "1") != "0" and
                             getattr(self.ops, "rel_inplace", True))
        self.slots = None
        import os
        # world 1 and no collective asked for: all rows are local - the in-place step on the shard (HipOps.step_local; test doubles
        # without it and KGE_DIST_LOCAL_SHORTCUT=0 keep the route -> gather -> step -> messages -> apply path)
        self.local_only = (spec.world == 1 and not self.coll and hasattr(self.ops, "step_local") and
                           os.environ.get("KGE_DIST_LOCAL_SHORTCUT", "1") != "0")
        self._pair_ok = os.environ.get("KGE_DIST_PAIR_APPLY", "1") != "0"      # (A/B aid: the two owner-side applies as two launches)
        # (compute graphs - precapture() - are OPT-IN: measured slower than six eager launches, 180 vs 172 us per forced-collective
        #  step at cfg-R: a graph launch costs more host time than the launches it replaces at this size)
        self._cg_on = os.environ.get("KGE_DIST_COMPUTE_GRAPH", "0") == "1" and isinstance(self.ops, HipOps)
        self._cgraphs = {}
        self._ggraphs = {}            # run_group: (sampler, slot range, corruption parity, capacity, schedule) -> hipGraph of a whole group
        self._routed, self._route_pool, self._route_key, self._route_lb = {}, None, None, {}   # prepare_group: routed ahead, per sampler slot
        self._pre = None              # (batch, LocalBatch, slot index, event) of a pull that ran ahead
        self._side = None
        self._parity = 0
        self._mark = None             # profile_phases(): called with a phase name behind every phase of the synchronous step

    # ---- buffers (allocated once, for the geometry of the first batch) -----------------------------------------------
    def _setup(self, b):
        W = self.spec.world
        if self.cap is None:
            self.cap = default_cap(self.ue_bound or b.UE, W, self.slack)
        if self.packed:
            # an entry is in both traces when a positive entity is also drawn as a negative: rare on large graphs (Freebase shard:
            # ~0.02 per batch), common on small ones - start small, ensure_capacity sizes it per group like the buckets.  Without an
            # exchange (world 1, no collective) nothing is read back per group: the region then holds a whole bucket
            self.cap2 = self.cap if not self.coll else max(1, min(self.cap, int(getattr(self, "cap2_start", 64))))
        self.geom = (b.B, b.C * b.N, self.ue_bound or b.UE)
        self.grown = []               # (old cap, new cap, fill that asked for it): ensure_capacity's record
        self.grown_extra = []         # the same for the extra region of the packed messages
        self.overflow = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.fill = torch.zeros(2, dtype=torch.int32, device=self.dev)           # {largest bucket fill, largest both-trace count of a bucket}
        self.fill_all = torch.zeros(2 * W, dtype=torch.int32, device=self.dev)
        self._alloc(b)

    def _alloc(self, b):
        """the exchange buffers for the current `cap` (again after ensure_capacity grew it; nothing of a step is in flight then)"""
        W = self.spec.world
        cap, dt, dev = self.cap, self.ent.dtype, self.dev
        ld_e, ld_r = (self.d_e + 4) if self.packed else (2 * self.d_e + 4), self.d_r + 4
        capT = cap + self.cap2            # message rows per owner bucket (cap2 = 0: two-trace messages)
        if self.packed:
            self.ops._msg_cap = (cap, self.cap2)

        def z(shape, dtype):
            return torch.zeros(shape, dtype=dtype, device=dev)
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()        # the old buffers may still be read by the previous group's steps
            if self._side is not None:
                self._side.synchronize()
        if hasattr(self.ops, "_structs"):
            self.ops._structs = {}    # HipOps caches its argument structs per buffer set (keyed by the buffers' addresses)
        if hasattr(self.ops, "_jobs"):
            self.ops._jobs = {}
        self._cgraphs = {}            # compute graphs hold the old buffers' addresses
        self.close_graphs()           # ... and so do the group graphs
        # whatever was routed ahead was routed for the OLD capacity (cache rows = owner * cap + position): drop it with the buffers
        self._routed, self._route_lb, self._route_pool, self._route_key = {}, {}, None, None
        self._gsend = self._graw = self._grecv = None
        self.slots = []
        for _ in range(2):
            s = _Slot()
            s.req_ids = z(W * cap, torch.int64)
            s.recv_ids = s.req_ids if not self.coll else z(W * cap, torch.int64)
            s.h_loc, s.t_loc, s.neg_loc = z(b.B, torch.int64), z(b.B, torch.int64), z(b.C * b.N, torch.int64)
            ue = self.ue_bound or b.UE
            s.ue_loc, s.ue_rec_loc = z(ue, torch.int64), z(ue * 8, torch.int32)
            s.ue_msg = z(ue * 2, torch.int32)
            s.cache = z((W * cap + 1, self.d_e), dt)              # + the dump row of overflowing entries
            s.rows_out = s.cache[:W * cap] if not self.coll else z((W * cap, self.d_e), dt)
            s.overflow = self.overflow
            self.slots.append(s)
        self.ent_msg = z((W * capT + 1, ld_e), dt)
        self.recv_msg = self.ent_msg[:W * capT] if not self.coll else z((W * capT, ld_e), dt)
        # (relation trace applied in place - relation partitioning: no relation messages at all; RESCAL's "row" is a d_e x d_e matrix)
        self.rel_msg = z((b.B, ld_r), dt) if not self._rel_inplace else z((1, 4), dt)
        self.all_rel = self.rel_msg if (not self.coll or self.rel_local) else z((W * b.B, ld_r), dt)
        self.zero_state = z(W * cap + 1, dt)
        self._ent_msg_alt = None      # the second entity-message buffer of the overlapped schedule (_steps_overlapped): made on first use

    def ensure_capacity(self, batches, log=None):
        """call with every freshly sampled GROUP of batches before its steps: measures the largest owner-bucket fill of the group on
        the device (one launch + one small read, all ranks' values exchanged so that every rank takes the same decision) and, when
        a bucket would overflow, grows `cap` (all exchange buffers are re-allocated) BEFORE any of the group's steps runs - no
        entity is ever trained against the dump row.  Heavy-tailed graphs cluster their hubs in a few shards: the default
        capacity (1.5 x the mean share) is a starting point, not a bound.  Returns the capacity in use."""
        if self.local_only:
            return self.cap
        if self.slots is None:
            self._setup(batches[0])
        W = self.spec.world
        if W == 1 and not (self.packed and self.coll):
            return self.cap                       # one owner: cap = the batch's bound on unique entities (extra region: a whole bucket)
        if self._pre is not None:
            raise _lib.KgeError("ensure_capacity: a pull is still in flight (call it between groups)")
        self.fill.zero_()
        self.ops.route_fill(batches, W, self.spec.shard, self.fill)
        if self.coll:
            self.comm.all_gather(self.fill_all, self.fill)
            both = self.fill_all.view(W, 2).max(0).values.tolist()
        else:
            both = self.fill.tolist()
        need, need2 = int(both[0]), int(both[1])
        grow = False
        if need > self.cap:
            ue_bound = self.geom[2]
            new = int(min(max(ue_bound, need), (int(need * 1.25) + 63) // 64 * 64))
            if log is not None:
                log("owner buckets grow from %d to %d rows (largest fill of the next group: %d)" % (self.cap, new, need))
            self.grown.append((self.cap, new, need))
            self.cap = new
            grow = True
        if self.packed and need2 > self.cap2:
            new2 = int(min(self.cap, (int(need2 * 1.25) + 63) // 64 * 64))
            if log is not None:
                log("extra region of the packed gradient messages grows from %d to %d rows per bucket (most both-trace rows of one "
                    "bucket in the next group: %d)" % (self.cap2, new2, need2))
            self.grown_extra.append((self.cap2, new2, need2))
            self.cap2 = new2
            grow = True
        if grow:
            self._alloc(batches[0])
        return self.cap

    def prepare_group(self, batches, log=None, check_capacity=True):
        """everything a freshly sampled GROUP of batches needs before its steps: the bucket capacity (ensure_capacity), - for
        consecutive slots of a device sampler - the routing of ALL its batches in ONE launch (kge_route_build_group) and, with
        peers, ONE all-to-all of the whole group's request ids (round 5): the ids every owner will be asked for during the group
        are known as soon as the group is sampled, so a step is left with two exchanges (rows, gradients) and its pull is one
        collective deep instead of two dependent ones.
        check_capacity=False (callers that record the group into a hipGraph, where nothing may be read back): the buckets keep
        their size and the overflow counter of the routing kernel is the check (check_overflow())."""
        if self.local_only:
            return                                   # every row is a row of this rank's table: nothing to size, route or exchange
        if check_capacity:
            self.ensure_capacity(batches, log)
        elif self.slots is None:
            self._setup(batches[0])
        self._routed = {}
        group = getattr(self.ops, "route_group", None)
        b0 = batches[0]
        smp = getattr(b0, "sampler", None)
        slots = [getattr(b, "slot", None) for b in batches]
        if (group is None or smp is None or any(getattr(b, "sampler", None) is not smp for b in batches) or
                slots != list(range(slots[0], slots[0] + len(slots)))):
            return                                   # host-built plans: routed one by one in the step
        W, cap, n = self.spec.world, self.cap, len(batches)
        off, stride = self.ops.route_layout(b0, W, cap)
        need = smp.n_slots * stride
        if self._route_pool is None or self._route_pool.numel() != need or self._route_key != (id(smp), cap, self.cap2):
            if self.dev.type == "cuda":
                torch.cuda.current_stream(self.dev).synchronize()
            self._route_pool = torch.zeros(need, dtype=torch.uint8, device=self.dev)
            self._route_key = (id(smp), cap, self.cap2)
            self._route_lb = {}
            self._cgraphs = {}
            if self.coll:                            # the group's id exchange: [owner][slot][cap] out, [source][slot][cap] in, and the
                z = lambda *sh: torch.full(sh, -1, dtype=torch.int64, device=self.dev)          # noqa: E731
                self._gsend, self._graw = z(W * smp.n_slots * cap), z(W * smp.n_slots * cap)    # per-step view [slot][source][cap]
                self._grecv = z(smp.n_slots, W * cap)
        if self.packed:
            group(batches, W, self.spec.shard, cap, self._route_pool, off, stride, self.overflow, cap2=self.cap2)
        else:
            group(batches, W, self.spec.shard, cap, self._route_pool, off, stride, self.overflow)
        if self._mark:
            self._mark("route")
        if self.coll:
            # every owner's share of the group's request ids in one piece: pool rows hold [owner][cap] per slot
            pool64 = self._route_pool.view(torch.int64)
            req = torch.as_strided(pool64, (n, W, cap), (stride // 8, cap, 1), (slots[0] * stride + off["req_ids"]) // 8)
            send, raw = self._gsend[:W * n * cap], self._graw[:W * n * cap]
            send.view(W, n, cap).copy_(req.permute(1, 0, 2))
            self.comm.all_to_all(raw, send)
            self._grecv[slots[0]:slots[0] + n].view(n, W, cap).copy_(raw.view(W, n, cap).permute(1, 0, 2))
            if self._mark:
                self._mark("ids_a2a")
        for b in batches:
            key = (b.slot, b.neg_head)
            self._routed[id(b)] = (self._slot_lb(b, off, stride), getattr(b, "gen", None))

    def _slot_lb(self, b, off, stride):
        """the re-addressed batch of a sampler slot (pointer arithmetic into the route pool: built once per slot and corruption mode)"""
        key = (b.slot, b.neg_head)
        lb = self._route_lb.get(key)
        if lb is None:
            lb = self._route_lb[key] = self.ops.routed_batch(b, self.spec.world, self.cap, self._route_pool, off, stride)
            lb.recv_ids = self._grecv[b.slot] if self.coll else lb.req_ids            # (filled by the group's id exchange)
        return lb

    def _routed_ahead(self, batch):
        """the re-addressed batch prepare_group made for `batch` - unless the sampler has overwritten the slot since (a DeviceBatch
        is one object per slot: its `gen` is the sampler launch that filled it), in which case the routing is stale and the
        step routes for itself"""
        hit = self._routed.get(id(batch))
        if hit is not None and hit[1] == getattr(batch, "gen", None):
            return hit[0]
        return None

    def close_graphs(self):
        """destroy the group graphs (they hold the exchange buffers' addresses and - through the recorded collectives - a reference
        on the communicator: RcclComm.close() blocks while one of them is alive)"""
        for g in getattr(self, "_ggraphs", {}).values():
            try:
                g.reset()
            except Exception:       # noqa: BLE001 - teardown
                pass
        self._ggraphs = {}

    def close(self):
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        self.close_graphs()
        if hasattr(self.comm, "close"):
            self.comm.close()

    def run_steps(self, batches, pipelined=False):
        """the steps of one group of batches, eagerly, under the given schedule (False: synchronous; True: the pull of step s+1 next
        to step s; "overlap": every exchange on the side stream) - what run_group records into its graphs; callers with host-built
        plans (no sampler slots) use this after ensure_capacity()."""
        return self._steps(batches, pipelined)

    def _steps(self, batches, pipelined):
        if self.local_only:                           # (no exchange to overlap: every schedule is the same in-place steps)
            for b in batches:
                self.ops.step_local(self.engine, b, self.ent, self.ent_state)
            return
        if pipelined == "overlap":
            return self._steps_overlapped(batches)
        for k, b in enumerate(batches):
            if pipelined:
                self.step_pipelined(b, batches[k + 1] if k + 1 < len(batches) else None)
            else:
                self.step(b)

    def _steps_overlapped(self, batches):
        """the steps of one group with EVERY exchange off the compute stream (round 5; the same --async_update licence as
        step_pipelined and the SAME results bit for bit: step s+1 computes on entity rows gathered after update s-1 and before
        update s, relation rows are never stale):

            main   pull(0) | compute(0) rel(0) | compute(1) rel(1) | compute(2) rel(2) | ...                         | join
            side           | pull(1)           | push(0) apply(0) pull(2) | push(1) apply(1) pull(3) | ... push(n-1) apply(n-1)

        compute(k) waits for pull(k) only; push(k) = the reverse all-to-all of step k's entity messages, apply(k) = the owner-side
        Adagrad on the shard, pull(k+2) = owner gather + row all-to-all into the cache slot compute(k) has just left.  The step's
        time is max(compute, its exchanges) instead of their sum (step_pipelined hides the pull only; the reference's
        --async_update hides the update the same way, tensor_models.py:136-175).  rel(k): the relation exchange (none under
        relation partitioning) and the relation apply stay on the compute stream, in step order - compute(k+1) reads the
        relation table itself (with all-gathered relations the exchange itself heads the side chain: one stream issues every
        collective).  Entity messages alternate between two buffers (push(k) reads one while compute(k+1) writes the
        other); the first step of a group pulls for itself behind the previous group's last apply, the group ends joined."""
        n = len(batches)
        if self.slots is None:
            self._setup(batches[0])
        host = self.dev.type != "cuda"                # CPU tensors (the gloo schedule tests): no streams - every call runs where it is
        if host:                                      # issued, which IS one valid serialisation of the two queues below
            class _Now(object):
                def record(self, stream): pass
                def wait(self, stream): pass
            main = side = None
            explicit = False
            ev_pull = [dict(main=_Now(), gather=_Now(), rows=_Now()) for _ in range(2)]
            ev_ov = [dict(comp=_Now(), rel=_Now()) for _ in range(2)] + [dict(done=_Now())]
        else:
            main = torch.cuda.current_stream(self.dev)
            if self._side is None:
                import os
                # the exchange stream goes ahead of the compute stream in the hardware queues: its chain is what the next-but-one
                # step waits for (world-1 proxy: 122.8 vs 125.8 us with all-gathered relations, neutral with partitioned ones;
                # KGE_DIST_SIDE_PRIORITY=0: equal priorities)
                self._side = torch.cuda.Stream(device=self.dev, priority=int(os.environ.get("KGE_DIST_SIDE_PRIORITY", "-1")))
                self._explicit = isinstance(self.comm, RcclComm) and isinstance(self.ops, HipOps)
                E = _lib.RawEvent if self._explicit else _lib.TorchEvent
                self._ev = [dict(main=E(), gather=E(), rows=E()) for _ in range(2)]
            if getattr(self, "_ev_ov", None) is None:
                E = _lib.RawEvent if self._explicit else _lib.TorchEvent
                self._ev_ov = [dict(comp=E(), rel=E()) for _ in range(2)] + [dict(done=E())]
            side, explicit, ev_pull, ev_ov = self._side, self._explicit, self._ev, self._ev_ov
        W, sp = self.spec.world, self.spec
        if self._ent_msg_alt is None:
            self._ent_msg_alt = torch.zeros_like(self.ent_msg)
        msgs = (self.ent_msg, self._ent_msg_alt)

        def on_side(fn):
            if host:
                return fn()
            if explicit:
                _lib.use_stream(side.cuda_stream)
                try:
                    return fn()
                finally:
                    _lib.use_stream(main.cuda_stream)
            with torch.cuda.stream(side):
                return fn()

        outer = _lib.use_stream(main.cuda_stream) if explicit else None
        try:
            if self._pre is not None:                 # a pull that ran ahead outside this schedule: behind it, then dropped
                self._pre[2].wait(main)
                self._pre = None
            p0 = self._parity
            lbs = [None] * n
            lbs[0] = self.pull(batches[0], p0)        # behind everything enqueued so far: the last apply of the previous group
            if n > 1:
                ev = ev_pull[p0 ^ 1]
                ev["main"].record(main)
                ev["main"].wait(side)
                lbs[1] = on_side(lambda: self._pull_ahead(batches[1], p0 ^ 1, ev))
            Wr = 1 if self.rel_local else W
            for k in range(n):
                par = k & 1
                lb = lbs[k]
                if k > 0:
                    ev_pull[lb.slot]["rows"].wait(main)
                msg = msgs[par]
                self._compute(lb, msg)
                evc = ev_ov[par]["comp"]
                evc.record(main)
                evc.wait(side)
                # the relation half: applied on the compute stream, in step order (compute(k+1) reads the table it updates); its
                # exchange - none under relation partitioning - heads the side chain, so that ONE stream issues every collective
                # of the communicator (two streams driving one communicator concurrently is what the supervisor's rccl-sync attempt
                # exists to rule out) and the compute stream waits for this one exchange only
                if self.coll and not self.rel_local:
                    on_side(lambda: self.comm.all_gather(self.all_rel.view(-1), self.rel_msg.view(-1)))
                    evr = ev_ov[par]["rel"]
                    evr.record(side)
                    evr.wait(main)
                if not self._rel_inplace:
                    self.ops.apply_merged(self.engine.rel, self.engine.rel_state, Wr, lb.B, None, 0, self.all_rel, 1, self.lr)

                def side_chain(k=k, lb=lb, msg=msg):
                    capT = self.cap + self.cap2
                    recv = self.recv_msg if self.coll else msg[:W * capT]
                    if self.coll:
                        self.comm.all_to_all(self.recv_msg, msg[:W * capT])
                    self._apply_ent(lb, recv)
                    if k + 2 < n:                     # into the cache slot compute(k) has just left
                        lbs[k + 2] = self._pull_ahead(batches[k + 2], lb.slot, ev_pull[lb.slot])
                on_side(side_chain)
            done = ev_ov[2]["done"]
            done.record(side)
            done.wait(main)
            self._parity = p0 ^ (n & 1)
        finally:
            if explicit:
                _lib.use_stream(outer)

    def run_group(self, batches, log=None, graph=True, pipelined=False):
        """one freshly sampled GROUP of batches, the trainer's order: size the owner buckets for it (ensure_capacity: the one device
        read of the group, every rank decides alike), then [route the whole group, exchange its request ids once, run its steps]
        - replayed from ONE hipGraph per (slot range, corruption parity, capacity, schedule) when the communicator can be recorded
        (RcclComm; no collective at world 1): kernels AND collectives of the group without a host call in between (round 5: the
        eager step was host-bound, 154 us against 138 us per cfg-R step at world 1 with the collectives forced).  The first
        group of a geometry runs eagerly (it allocates every buffer) and is recorded afterwards; a capacity change drops the graphs.
        pipelined: the pull of step s+1 on the side stream next to step s (--async_update licence), inside the graph as a fork;
        pipelined="overlap": push, owner-side apply and pull all on the side stream (_steps_overlapped) - the same results."""
        self.ensure_capacity(batches, log)
        b0 = batches[0]
        smp = getattr(b0, "sampler", None)
        slots = [getattr(b, "slot", None) for b in batches]
        ok = (graph and self.dev.type == "cuda" and smp is not None and isinstance(self.ops, HipOps) and
              (not self.coll or getattr(self.comm, "capturable", False)) and
              all(getattr(b, "sampler", None) is smp for b in batches) and slots == list(range(slots[0], slots[0] + len(slots))) and
              not torch.cuda.is_current_stream_capturing())
        if not ok:
            self.prepare_group(batches, check_capacity=False)
            self._steps(batches, pipelined)
            return False
        key = (id(smp), slots[0], len(slots), tuple(b.neg_head for b in batches[:2]), self.cap, self.cap2,
               pipelined if isinstance(pipelined, str) else bool(pipelined))
        g = self._ggraphs.get(key)
        if g is None:
            # (marks that are not multiples of each other produce many distinct group lengths; every graph pins buffers and
            #  - through its recorded collectives - communicator plans.  The cache is bounded: the oldest graph goes first)
            while len(self._ggraphs) >= int(getattr(self, "max_group_graphs", 16)):
                old = next(iter(self._ggraphs))
                try:
                    self._ggraphs.pop(old).reset()
                except Exception:       # noqa: BLE001
                    pass
            self.prepare_group(batches, check_capacity=False)      # this group: eagerly (allocates the pool and every buffer) ...
            self._steps(batches, pipelined)
            g = torch.cuda.CUDAGraph()                             # ... and recorded (nothing executes) for the groups to come
            with _lib.graph_capture(g):
                self.prepare_group(batches, check_capacity=False)
                self._steps(batches, pipelined)
            self._ggraphs[key] = g
            if hasattr(self.engine, "_graphs"):
                self.engine._graphs += 1                           # the workspace's address is baked in: it may not move any more
            return False
        g.replay()
        return True

    def precapture(self, sampler):
        """record the compute graphs of EVERY (sampler slot, corruption mode, cache slot) combination - pointer arithmetic only,
        nothing executes - so that no capture ever falls into a timed step.  Call once after the first prepare_group (buffers and
        route pool exist) and before training; a later change of the bucket capacity drops the graphs (the steps then launch
        their kernels one by one again until precapture is called again).  Returns the number of graphs, 0 if switched off
        (KGE_DIST_COMPUTE_GRAPH=0) or not applicable."""
        from .dataloader import DeviceBatch
        if not self._cg_on or self.dev.type != "cuda" or self._route_pool is None or self._route_key[0] != id(sampler):
            return 0
        W = self.spec.world
        b0 = sampler._batches.get((0, False)) or sampler._batches.get((0, True)) or DeviceBatch(sampler, 0, False)
        off, stride = self.ops.route_layout(b0, W, self.cap)
        self.engine.workspace_for(b0)
        try:
            for slot in range(sampler.n_slots):
                for nh in (False, True):
                    key = (slot, nh)
                    b = sampler._batches.get(key)
                    if b is None:
                        b = sampler._batches[key] = DeviceBatch(sampler, slot, nh)
                    lb = self._slot_lb(b, off, stride)
                    for par in (0, 1):
                        g = torch.cuda.CUDAGraph()
                        with _lib.graph_capture(g):
                            self.ops.step_grads(self.engine, lb, self.slots[par].cache, self.ent_msg,
                                                None if self._rel_inplace else self.rel_msg, self.zero_state)
                        self._cgraphs[(id(lb), par)] = g
            if hasattr(self.engine, "_graphs"):
                self.engine._graphs += 1             # the workspace's address is baked in: it may not move any more
        except Exception as e:                       # noqa: BLE001 - the eager launches are always available
            import sys
            print("DistEngine: compute graphs switched off (%r)" % (e,), file=sys.stderr)
            self._cgraphs, self._cg_on = {}, False
        return len(self._cgraphs)

    def check_overflow(self):
        """entries that did not fit their owner bucket since the last call (one 4-byte D2H read: call at the log interval)."""
        if self.slots is None or self.local_only:
            return 0
        n = int(self.overflow.item())
        if n:
            self.overflow.zero_()
        return n

    # ---- pull: route -> ids all-to-all -> owner gather -> rows all-to-all -----------------------------------------------
    def pull(self, batch, slot):
        if self.slots is None:
            self._setup(batch)
        if (batch.B, batch.C * batch.N) != self.geom[:2] or (batch.UE > self.geom[2] if self.ue_bound else batch.UE != self.geom[2]):
            raise _lib.KgeError("DistEngine: batch geometry changed (B, C*N, UE bound) %r -> %r" % (self.geom, (batch.B, batch.C * batch.N, batch.UE)))
        sp, s, W = self.spec, self.slots[slot], self.spec.world
        lb = self._routed_ahead(batch)               # routed by prepare_group - the owners already hold this step's request ids
        if lb is None:
            lb = self.ops.route(batch, W, sp.shard, self.cap, s, **({"cap2": self.cap2} if self.packed else {}))
            if self.coll:
                self.comm.all_to_all(s.recv_ids, lb.req_ids)
                lb.recv_ids = s.recv_ids
            else:
                lb.recv_ids = lb.req_ids
        if self._mark:
            self._mark("route")                     # (only when this step routed for itself: a no-op span otherwise)
        self.ops.gather_req(self.ent, lb.recv_ids, sp.lo, s.rows_out)
        if self._mark:
            self._mark("gather")
        if self.coll:
            self.comm.all_to_all(s.cache[:W * self.cap], s.rows_out)
            if self._mark:
                self._mark("rows_a2a")
        lb.slot = slot
        return lb

    # ---- compute + push + owner-side apply -----------------------------------------------------------------------------
    def _compute(self, lb, ent_msg=None):
        s = self.slots[lb.slot]
        ent_msg = self.ent_msg if ent_msg is None else ent_msg
        # relation partitioning: every relation row of the batch belongs to this rank - the step applies the relation trace to the
        # table itself (no relation messages, no relation apply launch; round 5, last session)
        rel_msg = None if self._rel_inplace else self.rel_msg
        if rel_msg is not None and not lb.c.counts_dev:
            self.ops.reset_rel_pads(self.rel_msg, self.d_r, lb.UR)      # host-built plan: UR is exact, the rows behind it are pads
        # The step's kernels between the pull and the push replay from a small hipGraph per (routed batch, cache slot) when
        # precapture() recorded one (opt-in, KGE_DIST_COMPUTE_GRAPH=1: bit-identical, and SLOWER than the six eager launches it
        # replaces - 180 vs 172 us per step).  These small per-step graphs predate run_group (round 5), which records the whole group WITH its collectives.
        if self._cgraphs and ent_msg is self.ent_msg and not torch.cuda.is_current_stream_capturing():
            g = self._cgraphs.get((id(lb), lb.slot))
            if g is not None:
                g.replay()
                return
        self.ops.step_grads(self.engine, lb, s.cache, ent_msg, rel_msg, self.zero_state)
        if self._mark:
            self._mark("compute")

    def _push_apply(self, lb, before_apply=None):
        sp, s, W = self.spec, self.slots[lb.slot], self.spec.world
        Wr = 1 if self.rel_local else W    # sources of relation messages: this rank alone under relation partitioning
        capT = self.cap + self.cap2        # message rows per bucket
        if self.coll and self.rel_local:
            self.comm.all_to_all(self.recv_msg, self.ent_msg[:W * capT])
        elif self.coll:                   # both exchanges depend on step_grads only: one grouped launch where the communicator can
            pair = getattr(self.comm, "push_pair", None)
            if pair is not None:
                pair(self.recv_msg, self.ent_msg[:W * capT], self.all_rel.view(-1), self.rel_msg.view(-1))
            else:
                self.comm.all_to_all(self.recv_msg, self.ent_msg[:W * capT])
                self.comm.all_gather(self.all_rel.view(-1), self.rel_msg.view(-1))
        if before_apply is not None:
            before_apply()
        if self._mark:
            self._mark("push")
        if self._rel_inplace:             # the relation trace was applied by the step itself
            self._apply_ent(lb, self.recv_msg)
            if self._mark:
                self._mark("apply")
            return
        pair = getattr(self.ops, "apply_merged_pair", None) if self._pair_ok else None
        if pair is not None and self.d_e % 4 == 0 and self.d_r % 4 == 0:      # both applies in one launch (same results)
            ent_job = ((self.ent, self.ent_state, W, self.cap, lb.recv_ids, sp.lo, self.recv_msg, 1, self.cap2) if self.packed else
                       (self.ent, self.ent_state, W, self.cap, lb.recv_ids, sp.lo, self.recv_msg, 2))
            pair(ent_job, (self.engine.rel, self.engine.rel_state, Wr, lb.B, None, 0, self.all_rel, 1), self.lr)
        else:
            self._apply_ent(lb, self.recv_msg)
            self.ops.apply_merged(self.engine.rel, self.engine.rel_state, Wr, lb.B, None, 0, self.all_rel, 1, self.lr)
        if self._mark:
            self._mark("apply")

    def _apply_ent(self, lb, recv):
        """owner-side Adagrad of the received entity messages (two-trace messages, or packed single-trace ones with their extra region)"""
        sp, W = self.spec, self.spec.world
        if self.packed:
            self.ops.apply_merged(self.ent, self.ent_state, W, self.cap, lb.recv_ids, sp.lo, recv, 1, self.lr, cap_extra=self.cap2)
        else:
            self.ops.apply_merged(self.ent, self.ent_state, W, self.cap, lb.recv_ids, sp.lo, recv, 2, self.lr)

    def profile_phases(self, batches):
        """diagnostic (round 6, VERDICT r05 next-7: the first run on a multi-GPU node must explain itself): ONE group of batches as
        eager, SYNCHRONOUS steps with a HIP event behind every phase - group level: route (kge_route_build_group), ids_a2a (the
        group's request ids); per step: gather (owner-side kge_gather_rows_req), rows_a2a, compute (kge_step_grads: the step's
        kernels against the row cache), push (gradient all-to-all + relation all-gather), apply (owner-side Adagrad).  Trains the
        batches like any other step.  Returns {phase: microseconds per STEP} of this rank (+ 'steps'); at world 1 without
        collectives (local_only) the step is the in-place single-table step: one 'compute' phase."""
        if self.dev.type != "cuda":
            raise _lib.KgeError("profile_phases needs the device path")
        n = len(batches)
        marks = []

        def mark(name):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((name, ev))
        torch.cuda.synchronize(self.dev)
        mark("start")
        if self.local_only:
            for b in batches:
                self.ops.step_local(self.engine, b, self.ent, self.ent_state)
            mark("compute")
        else:
            self.ensure_capacity(batches)
            mark("capacity_check")
            self._mark = mark
            try:
                self.prepare_group(batches, check_capacity=False)
                for b in batches:
                    self.step(b)
            finally:
                self._mark = None
        torch.cuda.synchronize(self.dev)
        out = {}
        for (_, e0), (name, e1) in zip(marks[:-1], marks[1:]):
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1) * 1e3
        res = {k: round(v / n, 2) for k, v in out.items()}
        res["steps"] = n
        res["sum_us_per_step"] = round(sum(out.values()) / n, 2)
        return res

    def step(self, batch):
        """one synchronous sharded step (pull, compute, push, apply), all enqueued on the current stream."""
        if self.local_only:
            return self.ops.step_local(self.engine, batch, self.ent, self.ent_state)
        lb = self.pull(batch, 0)
        self._compute(lb)
        self._push_apply(lb)

    def _pull_ahead(self, next_batch, nslot, ev):
        """the pull of the NEXT step (route, id exchange, owner gather, row exchange) on the side stream"""
        sp, s, W = self.spec, self.slots[nslot], self.spec.world      # (the buffers exist: the first step's own pull made them)
        nlb = self._routed_ahead(next_batch)
        if nlb is None:
            nlb = self.ops.route(next_batch, W, sp.shard, self.cap, s, **({"cap2": self.cap2} if self.packed else {}))
            if self.coll:
                self.comm.all_to_all(s.recv_ids, nlb.req_ids)
                nlb.recv_ids = s.recv_ids
            else:
                nlb.recv_ids = nlb.req_ids
        nlb.slot = nslot
        self.ops.gather_req(self.ent, nlb.recv_ids, sp.lo, s.rows_out)
        ev["gather"].record(self._side)
        if self.coll:
            self.comm.all_to_all(s.cache[:W * self.cap], s.rows_out)
        ev["rows"].record(self._side)
        return nlb

    def step_pipelined(self, batch, next_batch=None):
        """one step with the pull of `next_batch` (the batch of the following call) overlapped: its rows are gathered on a
        side stream after update s-1 and before update s lands (exact one-step staleness: the reference's --async_update
        licence, tensor_models.py:136-175), while this step computes."""
        if self.local_only:
            return self.ops.step_local(self.engine, batch, self.ent, self.ent_state)
        main = torch.cuda.current_stream(self.dev)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
            # communicators that take the stream from _lib.stream_ptr() (RcclComm, like every library call) need no
            # `with torch.cuda.stream(...)` around the side-stream section; their events are raw hipEvents on raw stream pointers
            # (~1 us of host time per call instead of 4-5: the eager step is host-bound)
            self._explicit = isinstance(self.comm, RcclComm) and isinstance(self.ops, HipOps)
            E = _lib.RawEvent if self._explicit else _lib.TorchEvent
            # events are reused (one set per slot): creating them per step costs more host time than the calls they order
            self._ev = [dict(main=E(), gather=E(), rows=E()) for _ in range(2)]
        # every library call of this step takes its stream from the thread-local override: torch.cuda.current_stream() costs
        # ~5 us per call, and a step makes a dozen calls
        outer = _lib.use_stream(main.cuda_stream) if self._explicit else None
        try:
            if self._pre is not None and self._pre[0] is batch:
                _, lb, ev_rows = self._pre
                ev_rows.wait(main)
            else:
                if self._pre is not None:
                    # a pull that ran ahead for ANOTHER batch is dropped: the side stream may still be filling its slot (and using
                    # the communicator) - the fresh pull below must start behind it
                    self._pre[2].wait(main)
                lb = self.pull(batch, self._parity)
            self._pre = None
            ev_gather = None
            if next_batch is not None:
                nslot = lb.slot ^ 1
                ev = self._ev[nslot]
                ev["main"].record(main)
                ev["main"].wait(self._side)               # behind everything enqueued so far: the apply of step s-1
                if self._explicit:
                    _lib.use_stream(self._side.cuda_stream)
                    try:
                        nlb = self._pull_ahead(next_batch, nslot, ev)
                    finally:
                        _lib.use_stream(main.cuda_stream)
                else:
                    with torch.cuda.stream(self._side):
                        nlb = self._pull_ahead(next_batch, nslot, ev)
                ev_gather = ev["gather"]
                self._pre = (next_batch, nlb, ev["rows"])
            self._compute(lb)
            # the apply of step s must not start before the gather of step s+1 has read the shard (else the staleness is a race)
            self._push_apply(lb, (lambda: ev_gather.wait(main)) if ev_gather is not None else None)
            self._parity = lb.slot ^ 1
        finally:
            if self._explicit:
                _lib.use_stream(outer)


def relation_partition(rels, world):
    """whole relations to ranks, most frequent first, each to the rank with the fewest edges so far (the greedy core of the
    reference's partitioners, dataloader/sampler.py:32-254, WITHOUT their splitting of a large relation over several partitions:
    a split relation needs the cross-relation machinery - dual writes to a global table, general_models.py:590-637 - that
    relation-local updates are there to avoid; the price is a less even edge split when one relation holds more than 1 / world of
    the edges - soft_relation_partition / choose_relation_partition below take over then).
    Returns (owner[n_rel_seen_max + 1] with -1 for relations without edges, part[i] = rank of edge i)."""
    rels = np.asarray(rels, np.int64)
    uniq, cnts = np.unique(rels, return_counts=True)
    order = np.argsort(-cnts, kind="stable")
    load = np.zeros(world, np.int64)
    owner = np.full(int(uniq.max()) + 1 if len(uniq) else 0, -1, np.int64)
    for k in order:
        w = int(np.argmin(load))
        owner[uniq[k]] = w
        load[w] += cnts[k]
    return owner, owner[rels]


def soft_relation_partition(rels, world, threshold=0.05):
    """the partition the reference's `--rel_part` really computes (`TrainDataset` calls SoftRelationPartition,
    dataloader/sampler.py:32-148, 363-365): relations most frequent first (the reference's order: `np.flip(np.argsort(cnts))`, ties
    included); a LARGE relation - more than min(threshold * |E|, |E| / world) edges - is dealt evenly over ALL ranks (`cnt // world + 1`
    edges to rank 0, 1, ... until it is used up; these are the reference's `cross_rels`), every other relation goes whole to the rank
    with the fewest edges so far.  A relation's edges fill its ranks in the order of the edge list.
    Returns (part[i] = rank of edge i, rel_parts[k] = the relations rank k holds in the order they were dealt, cross_rels).
    Pinned against the reference's own function: tests/golden/relpart/*.npz (tests/golden/gen_golden_relpart.py)."""
    rels = np.asarray(rels, np.int64)
    n_edges = len(rels)
    uniq, cnts = np.unique(rels, return_counts=True)
    order = np.flip(np.argsort(cnts))
    large = min(int(n_edges * threshold), int(n_edges / world))
    load = np.zeros(world, np.int64)
    rel_parts = [[] for _ in range(world)]
    cross = []
    # one (first rank-in-relation, rank) segment per (relation, rank) pair: edge number k of relation r (k counted along the edge list)
    # belongs to the last segment of r that starts at or below k
    seg_rel, seg_start, seg_rank = [], [], []
    for k in order:
        r, cnt = int(uniq[k]), int(cnts[k])
        if cnt > large:
            share, done = cnt // world + 1, 0
            for w in range(world):
                take = min(share, cnt - done)
                rel_parts[w].append(r)            # (the reference lists r on every rank, also on one whose share came out empty)
                if take > 0:
                    seg_rel.append(r); seg_start.append(done); seg_rank.append(w)
                load[w] += take
                done += take
            cross.append(r)
        else:
            w = int(np.argmin(load))
            rel_parts[w].append(r)
            seg_rel.append(r); seg_start.append(0); seg_rank.append(w)
            load[w] += cnt
    # k = position of every edge among the edges of its relation, in edge-list order
    by_rel = np.argsort(rels, kind="stable")
    first = np.searchsorted(rels[by_rel], uniq)
    k_in_rel = np.empty(n_edges, np.int64)
    k_in_rel[by_rel] = np.arange(n_edges) - np.repeat(first, cnts)
    if n_edges == 0:
        return np.zeros(0, np.int64), [np.asarray(x, np.int64) for x in rel_parts], np.asarray(cross, np.int64)
    span = int(cnts.max()) + 1
    seg_key = np.asarray(seg_rel, np.int64) * span + np.asarray(seg_start, np.int64)
    so = np.argsort(seg_key, kind="stable")
    seg = np.searchsorted(seg_key[so], rels * span + k_in_rel, side="right") - 1
    part = np.asarray(seg_rank, np.int64)[so][seg]
    return part, [np.asarray(x, np.int64) for x in rel_parts], np.asarray(cross, np.int64)


def choose_relation_partition(rels, world, policy="auto", tolerance=1.1):
    """what `--rel_part` does with the training triples (`--rel_part_policy`):
      whole  whole relations to ranks (relation_partition): every relation row has ONE owner, no relation exchange at all;
      soft   the reference's partition (soft_relation_partition): large relations dealt over all ranks - an even edge split, and as
             soon as one relation lives on several ranks the relation gradients are all-gathered and applied by every rank
             (DistEngine(rel_local=False): exact for ANY edge split);
      auto   whole while the fullest rank stays within `tolerance` x the mean edge share, soft otherwise.
    Returns (mode, part[i] = rank of edge i, owner[r] = the rank of relation r / -1 no edges / -2 split over ranks, cross_rels)."""
    if policy not in ("auto", "soft", "whole"):
        raise ValueError("relation partition policy %r" % (policy,))
    rels = np.asarray(rels, np.int64)
    if policy in ("auto", "whole"):
        owner, part = relation_partition(rels, world)
        cnt = np.bincount(part, minlength=world)
        if policy == "whole" or cnt.max() <= tolerance * cnt.mean():
            return "whole", part, owner, np.zeros(0, np.int64)
    part, rel_parts, cross = soft_relation_partition(rels, world)
    owner = np.full(int(rels.max()) + 1 if len(rels) else 0, -1, np.int64)
    for k, rs in enumerate(rel_parts):
        owner[rs] = k
    owner[cross] = -2
    return "soft", part, owner, cross


def relation_rows_from_owners(rel, rel_state, owner, group=None):
    """after training with rel_local: every rank's replica holds the current rows of ITS relations only - collect the owners' rows
    into rank 0's replica (gather through the process group; relations nobody owns keep their initial rows)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    own = torch.as_tensor(np.nonzero(np.asarray(owner) == rank)[0], dtype=torch.int64)
    mine = (own, rel[own.to(rel.device)].cpu(), rel_state[own.to(rel.device)].cpu())
    parts = [None] * world if rank == 0 else None
    dist.gather_object(mine, parts, dst=0, group=group)
    if rank == 0:
        for ids, rows, st in parts:
            if len(ids):
                rel[ids.to(rel.device)] = rows.to(rel.device)
                rel_state[ids.to(rel.device)] = st.to(rel.device)


def localize_plan(h, t, r, neg, chunk, N, neg_head, edge_w=None):
    """(kept for callers that build cache-local plans on the host) global-id batch -> (sorted unique entity ids, plan over
    row indices into that sorted list)."""
    h = np.asarray(h, np.int64)
    t = np.asarray(t, np.int64)
    neg = np.asarray(neg, np.int64)
    ue = np.unique(np.concatenate([h, t, neg]))
    p = _plan.build_plan(np.searchsorted(ue, h), np.searchsorted(ue, t), r,
                         np.searchsorted(ue, neg), chunk, N, neg_head, edge_w)
    assert p["UE"] == ue.shape[0]
    return ue, p
