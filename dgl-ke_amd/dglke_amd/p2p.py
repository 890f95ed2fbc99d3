"""Peer-to-peer sharded tables: the multi-GPU Hogwild mode of the step (`kge_step_sharded`).

The reference's multi-GPU trainer (`train.py:298-317`, `--num_proc` = one trainer process per GPU)
keeps the entity table in *shared host memory* (`ExternalEmbedding.share_memory`,
tensor_models.py:233-236): every process gathers its rows from it and applies its sparse Adagrad
update to it without locks (`:304-362`).  On an 8 x MI355X node the natural home of that shared
table is the union of the GPUs' HBM: rank k owns rows [k*per, (k+1)*per) of the entity table, the
relation table and both Adagrad states, all ranks map every peer shard into their address space
(hipIpc handles exchanged once through `torch.distributed.all_gather_object`) and the step kernels
resolve a row id to `shard_base[id // per] + (id % per) * dim` - remote rows are read and
read-modify-written directly over xGMI.  No collective, no host work per step; within one rank the
update stays owner-computes and deterministic, across ranks it is Hogwild like the reference.

`ShardedTables(..., emulate=k)` builds k shards inside ONE process (no IPC) so that the shard
addressing of the kernels can be parity-tested on a single GPU.

`rel_local=True` (round 6, ABI 8 `kge_shards.rel_local`): only the ENTITY table is spread over the GPUs; the
relation-side tables - relation rows (RESCAL: d_e x d_e matrices) and, with `proj_dim`, TransR's projection table
(score_fun.py:114-118) - are whole tables LOCAL to every rank.  That is the reference's `--rel_part` layout
(general_models.py:590-637; its multi-GPU TransR recipe passes it, examples/freebase/multi_gpu.sh:80-89): the
trainer that holds a relation's edges holds and updates its rows; `collect_relations()` brings the owners' rows
together when the tables are read.  TransR and RESCAL train on sharded tables this way (their kernels read the
batch's entity rows from dense copies gathered through the shard map, kge_api.hip `sh_dense`).
"""
import ctypes as C

import torch

from . import _lib


def _align(x, a=256):
    return (x + a - 1) // a * a


class ShardedTables(object):
    def __init__(self, n_entities, n_relations, d_e, d_r, device, world=1, rank=0, group=None,
                 emulate=0, rel_local=False, proj_dim=0):
        self.dev = torch.device(device)
        if self.dev.type != 'cuda':
            raise _lib.KgeError("ShardedTables needs a CUDA (HIP) device; there is no CPU path")
        self.n_entities, self.n_relations = int(n_entities), int(n_relations)
        self.d_e, self.d_r = int(d_e), int(d_r)
        self.n_shards = int(emulate) if emulate else int(world)
        self.rank = 0 if emulate else int(rank)
        self.emulate = bool(emulate)
        self.group = group
        self.rel_local = bool(rel_local)
        self.proj_dim = int(proj_dim)
        if self.proj_dim and not self.rel_local:
            raise _lib.KgeError("a projection table needs rel_local=True (TransR on sharded tables)")
        self.ent_per = (self.n_entities + self.n_shards - 1) // self.n_shards
        # (rel_local: no relation rows in the arenas - one padding row keeps the layout arithmetic)
        self.rel_per = 1 if self.rel_local else (self.n_relations + self.n_shards - 1) // self.n_shards
        # arena of one shard: [entity rows | entity state | relation rows | relation state]
        self._off = [0]
        for nbytes in (self.ent_per * self.d_e * 4, self.ent_per * 4, (4 if self.rel_local else self.rel_per * self.d_r) * 4,
                       self.rel_per * 4):
            self._off.append(self._off[-1] + _align(nbytes))
        self.arena_bytes = self._off[-1]
        self._opened = []
        n_local = self.n_shards if emulate else 1
        self.arenas = [torch.zeros(self.arena_bytes, dtype=torch.uint8, device=self.dev) for _ in range(n_local)]
        if emulate:
            bases = [a.data_ptr() for a in self.arenas]
        else:
            bases = self._exchange(self.arenas[0])
        self.bases = bases
        tbl = [[b + self._off[k] for b in bases] for k in range(4)]
        self.ptr_table = torch.tensor(tbl, dtype=torch.int64, device=self.dev)      # [4, n_shards]
        sh = _lib.KgeShards()
        sh.n_shards = self.n_shards
        sh.ent_rows_per_shard, sh.rel_rows_per_shard = self.ent_per, self.rel_per
        p0 = self.ptr_table.data_ptr()
        sh.ent_rows, sh.ent_state = p0, p0 + 8 * self.n_shards
        sh.rel_rows, sh.rel_state = p0 + 16 * self.n_shards, p0 + 24 * self.n_shards
        sh.n_ent, sh.n_rel = self.n_entities, self.n_relations
        self.rel_tab = self.rel_state_tab = self.proj_tab = self.proj_state_tab = None
        if self.rel_local:
            self.rel_tab = torch.zeros(self.n_relations, self.d_r, dtype=torch.float32, device=self.dev)
            self.rel_state_tab = torch.zeros(self.n_relations, dtype=torch.float32, device=self.dev)
            sh.rel_local, sh.rel_state_local = _lib.ptr(self.rel_tab), _lib.ptr(self.rel_state_tab)
            if self.proj_dim:
                self.proj_tab = torch.zeros(self.n_relations, self.proj_dim, dtype=torch.float32, device=self.dev)
                self.proj_state_tab = torch.zeros(self.n_relations, dtype=torch.float32, device=self.dev)
                sh.proj_local, sh.proj_state_local = _lib.ptr(self.proj_tab), _lib.ptr(self.proj_state_tab)
        self.c = sh

    # ---- IPC ---------------------------------------------------------------------------------
    def _exchange(self, arena):
        import torch.distributed as dist
        lib = _lib.lib()
        handle = (C.c_ubyte * _lib.IPC_HANDLE_BYTES)()
        off = C.c_int64(0)
        _lib.check(lib.kge_ipc_export(arena.data_ptr(), handle, C.byref(off)))
        my_dev = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        mine = (bytes(handle), int(off.value), int(my_dev))
        world = dist.get_world_size(self.group)
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=self.group)
        # refuse BEFORE any kernel touches a peer mapping: a load from memory this GPU cannot reach is a GPU fault
        # (the process dies), not an exception a caller could turn into the all-to-all fall-back
        bad = [d for r, (_, _, d) in enumerate(everyone)
               if r != self.rank and d != my_dev and not torch.cuda.can_device_access_peer(my_dev, d)]
        verdicts = [None] * world                      # decided TOGETHER: every rank raises or none does
        dist.all_gather_object(verdicts, bad, group=self.group)
        if any(verdicts):
            raise _lib.KgeError("no peer-to-peer access between some GPUs: %s" % (
                "; ".join("rank %d -> GPUs %s" % (r, v) for r, v in enumerate(verdicts) if v)))
        bases = []
        for r, (h, o, _) in enumerate(everyone):
            if r == self.rank:
                bases.append(arena.data_ptr())
                continue
            buf = (C.c_ubyte * _lib.IPC_HANDLE_BYTES).from_buffer_copy(h)
            base = C.c_void_p()
            _lib.check(lib.kge_ipc_open(buf, C.byref(base)))
            self._opened.append(base.value)
            bases.append(base.value + o)
        return bases

    def close(self):
        for b in self._opened:
            _lib.lib().kge_ipc_close(b)
        self._opened = []

    # ---- views of the LOCAL shard(s) ----------------------------------------------------------
    def _view(self, shard, k, rows, cols):
        a = self.arenas[shard if self.emulate else 0]
        t = a[self._off[k]:self._off[k] + rows * cols * 4].view(torch.float32)
        return t.view(rows, cols) if cols > 1 else t

    def ent(self, shard=0):
        return self._view(shard, 0, self.ent_per, self.d_e)

    def ent_state(self, shard=0):
        return self._view(shard, 1, self.ent_per, 1)

    def rel(self, shard=0):
        if self.rel_local:
            return self.rel_tab
        return self._view(shard, 2, self.rel_per, self.d_r)

    def rel_state(self, shard=0):
        if self.rel_local:
            return self.rel_state_tab
        return self._view(shard, 3, self.rel_per, 1)

    def init_uniform(self, emb_init, seed):
        """U(-emb_init, emb_init) rows, zero state (general_models.py:217-218, tensor_models.py:246-249)
        for the shard(s) this process owns; rank-dependent seed."""
        g = torch.Generator(device=self.dev)
        for s in range(len(self.arenas)):
            g.manual_seed(int(seed) * 1000003 + self.rank * 101 + s)
            self.ent(s).uniform_(-emb_init, emb_init, generator=g)
            if not self.rel_local:
                self.rel(s).uniform_(-emb_init, emb_init, generator=g)
                self.rel_state(s).zero_()
            self.ent_state(s).zero_()
        if self.rel_local:
            # identical replicas on every rank (rank-independent seed): a rank only ever changes the rows of ITS relations
            g.manual_seed(int(seed) * 1000003 + 77)
            self.rel_tab.uniform_(-emb_init, emb_init, generator=g)
            self.rel_state_tab.zero_()
            if self.proj_tab is not None:         # TransRScore.reset_parameters: projection_emb.init(1.0)
                self.proj_tab.uniform_(-1.0, 1.0, generator=g)
                self.proj_state_tab.zero_()

    def load_full(self, ent, rel, ent_state=None, rel_state=None, proj=None, proj_state=None):
        """scatter full tables over the LOCAL shards (emulation: all of them; multi-process: the own
        range) - test / checkpoint-load helper.  rel_local: the relation-side tables are whole local tables."""
        if self.rel_local:
            for full, tab in ((rel, self.rel_tab), (rel_state, self.rel_state_tab), (proj, self.proj_tab),
                              (proj_state, self.proj_state_tab)):
                if full is not None and tab is not None:
                    tab.copy_(torch.as_tensor(full).to(self.dev).view(tab.shape))
            rel = rel_state = None
        for s in range(len(self.arenas)):
            k = s if self.emulate else self.rank
            for full, view, per in ((ent, self.ent(s), self.ent_per), (rel, None if self.rel_local else self.rel(s), self.rel_per),
                                    (ent_state, self.ent_state(s), self.ent_per),
                                    (rel_state, None if self.rel_local else self.rel_state(s), self.rel_per)):
                if full is None:
                    continue
                full = torch.as_tensor(full)
                lo, hi = min(k * per, full.shape[0]), min((k + 1) * per, full.shape[0])
                view[:hi - lo].copy_(full[lo:hi])

    def full(self, which):
        """emulation only: the logical table re-assembled from the shards."""
        if not self.emulate:
            raise _lib.KgeError("full() needs all shards in this process (emulate mode)")
        fn, n = {"ent": (self.ent, self.n_entities), "ent_state": (self.ent_state, self.n_entities),
                 "rel": (self.rel, self.n_relations), "rel_state": (self.rel_state, self.n_relations)}[which]
        if self.rel_local and which in ("rel", "rel_state"):
            return fn(0).clone()
        return torch.cat([fn(s) for s in range(self.n_shards)], 0)[:n].clone()

    def gather(self, which, idx):
        """rows (or state values) of the sharded table by GLOBAL id, from whichever GPU holds them."""
        k, dim, per = {"ent": (0, self.d_e, self.ent_per), "ent_state": (1, 1, self.ent_per),
                       "rel": (2, self.d_r, self.rel_per), "rel_state": (3, 1, self.rel_per)}[which]
        idx = idx.contiguous()
        if self.rel_local and which in ("rel", "rel_state"):
            return (self.rel_tab if which == "rel" else self.rel_state_tab)[idx]
        out = torch.empty(idx.shape[0], dim, dtype=torch.float32, device=self.dev)
        _lib.check(_lib.lib().kge_gather_rows_sharded(
            self.ptr_table.data_ptr() + 8 * self.n_shards * k, self.n_shards, per, dim, _lib.ptr(idx),
            idx.shape[0], _lib.ptr(out), _lib.stream_ptr()))
        return out if dim > 1 else out.view(-1)

    def collect_relations(self, owner, group=None):
        """rel_local: every rank's relation-side tables hold the current rows of ITS relations only (owner[r] = rank of relation r,
        dist.relation_partition) - bring the owners' rows (relation rows + state, projection rows + state) into rank 0's tables.
        Collective over the process group."""
        import numpy as np
        import torch.distributed as dist
        if not self.rel_local:
            return
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        own = torch.as_tensor(np.nonzero(np.asarray(owner) == rank)[0], dtype=torch.int64)
        tabs = [t for t in (self.rel_tab, self.rel_state_tab, self.proj_tab, self.proj_state_tab) if t is not None]
        mine = (own, [t[own.to(self.dev)].cpu() for t in tabs])
        parts = [None] * world if rank == 0 else None
        dist.gather_object(mine, parts, dst=0, group=group)
        if rank == 0:
            for ids, rows in parts[1:]:
                if len(ids):
                    for t, r in zip(tabs, rows):
                        t[ids.to(self.dev)] = r.to(self.dev)

    def probe(self):
        """write a marker into the own shard, read every shard's marker through the map: proves
        that the peer mappings really reach the other GPUs' memory.  Collective (barriers)."""
        import torch.distributed as dist
        st = self.ent_state(0)
        old = st[0].clone()
        st[0] = float(self.rank + 1)
        torch.cuda.synchronize()
        dist.barrier(group=self.group)
        ids = torch.arange(self.n_shards, dtype=torch.int64, device=self.dev) * self.ent_per
        got = self.gather("ent_state", ids).cpu().tolist()
        dist.barrier(group=self.group)
        st[0] = old
        torch.cuda.synchronize()
        dist.barrier(group=self.group)
        return got == [float(r + 1) for r in range(self.n_shards)]
