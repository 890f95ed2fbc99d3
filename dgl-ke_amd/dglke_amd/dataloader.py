"""Batch objects with the members the reference reads from DGL's sampler output
(dataloader/sampler.py:421-457, 862-876; models/general_models.py:376-427, 548-549), backed by
flat id tensors instead of DGL subgraphs, plus a uniform chunked negative sampler that restates
the EdgeSampler call of dataloader/sampler.py:408-419 (negatives uniform over all entities, with
replacement, positives not excluded, C*N = B corrupt ids per batch) and the alternating
head/tail iterator of dataloader/sampler.py:823-876."""
import numpy as np
import torch as th

from . import _lib

from . import plan as _plan


class _Edges(object):
    def __init__(self, src, dst, data):
        self.src, self.dst, self.data = src, dst, data


class PosGraph(object):
    """pos_g: ndata['id'] unique entity ids, edata['id'] relation ids, all_edges -> local ids."""

    def __init__(self, batch):
        self.batch = batch
        self.ndata = {'id': batch.view('nid')}
        self.edata = {'id': batch.view('rel_ids')}
        if batch.p['edge_w'] is not None:
            self.edata['impts'] = batch.view('edge_w')
        self._h = batch.view('h_local')
        self._t = batch.view('t_local')

    def all_edges(self, order='eid'):
        return self._h, self._t

    def number_of_edges(self):
        return self.batch.B

    def apply_edges(self, fn):
        from . import ops
        emb = self.ndata['emb']
        self.edata.update(fn(_Edges({'emb': ops.gather_local(emb, self._h)}, {'emb': ops.gather_local(emb, self._t)},
                                    self.edata)))


class NegGraph(object):
    """neg_g: ndata['id'] corrupt entity ids, head_nid/tail_nid local ids, chunk geometry."""

    def __init__(self, batch):
        self.batch = batch
        self.ndata = {'id': batch.view('neg_ids')}
        self.edata = {}
        n = batch.C * batch.N
        self.head_nid = th.arange(n, device=batch.buf.device)
        self.tail_nid = self.head_nid
        self.num_chunks = batch.C
        self.chunk_size = batch.chunk
        self.neg_sample_size = batch.N
        self.neg_head = batch.neg_head


class UniformChunkedSampler(object):
    """Infinite iterator over (pos_g, neg_g): shuffled epochs over the training triples, batches
    whose size is not a multiple of the chunk size are dropped (sampler.py:503-504), odd steps
    corrupt tails and even steps heads (sampler.py:853-859)."""

    def __init__(self, heads, rels, tails, n_entities, batch_size, neg_sample_size, device,
                 neg_chunk_size=None, seed=0, edge_importance=None, prefetch=64):
        self.h = np.asarray(heads, np.int64)
        self.r = np.asarray(rels, np.int64)
        self.t = np.asarray(tails, np.int64)
        self.w = None if edge_importance is None else np.asarray(edge_importance, np.float32)
        self.n_entities = n_entities
        self.B = batch_size
        self.N = neg_sample_size
        self.chunk = neg_chunk_size or neg_sample_size
        if self.B % self.chunk:
            raise ValueError("batch_size should be divisible by the chunk size")
        self.device = device
        self.rng = np.random.RandomState(seed)
        self.step = 0
        self.prefetch = prefetch
        self._queue = []
        self._perm = None
        self._pos = 0

    def _next_ids(self):
        n = self.h.shape[0]
        while True:
            if self._perm is None or self._pos + self.B > n:
                self._perm = self.rng.permutation(n)
                self._pos = 0
                if n < self.B:
                    raise ValueError("fewer training triples than the batch size")
            sel = self._perm[self._pos:self._pos + self.B]
            self._pos += self.B
            return sel

    def next_plans(self, count):
        plans = []
        for _ in range(count):
            self.step += 1
            sel = self._next_ids()
            neg = self.rng.randint(0, self.n_entities, size=(self.B // self.chunk) * self.N)
            plans.append(_plan.build_plan(self.h[sel], self.t[sel], self.r[sel], neg, self.chunk,
                                          self.N, self.step % 2 == 0,
                                          None if self.w is None else self.w[sel]))
        return plans

    def next_batches(self, count):
        return _plan.upload(self.next_plans(count), self.device)

    def __iter__(self):
        return self

    def __next__(self):
        if not self._queue:
            self._queue = self.next_batches(self.prefetch)
        b = self._queue.pop(0)
        return PosGraph(b), NegGraph(b)


class DeviceBatch(object):
    """a batch that lives in a sampler slot in HBM: same duck type as plan.Batch for StepEngine
    (`.c` = kge_batch, sizes as attributes); UE / UR are upper bounds, the kernels read the actual
    counts from the slot."""

    def __init__(self, sampler, slot, neg_head):
        from . import _lib
        import ctypes as C
        self.sampler = sampler
        self.slot = slot
        self.B, self.C, self.chunk, self.N = sampler.B, sampler.C, sampler.chunk, sampler.N
        self.neg_head = bool(neg_head)
        kb = _lib.KgeBatch()
        _lib.check(_lib.lib().kge_batch_from_slot(_lib.ptr(sampler.slots), sampler.slot_bytes, slot, self.B,
                                                  self.C, self.chunk, self.N, int(self.neg_head), C.byref(kb)))
        self.c = kb
        self.U, self.UE, self.UR = 0, kb.UE, kb.UR


class DeviceSampler(object):
    """On-device sampler + plan builder (libkge_hip `kge_sample_batches`): the training triples, the
    epoch permutation, the RNG state and the batch slots all live in HBM; `sample()` enqueues ONE
    kernel that builds `n_slots` consecutive batches (no host work, graph-capturable).  Restates the
    EdgeSampler wrappers of dataloader/sampler.py:376-419, 823-876 (uniform negatives with
    replacement over all entities, alternating tail/head corruption, whole batches only)."""

    def __init__(self, heads, rels, tails, n_entities, batch_size, neg_sample_size, device, n_slots=64,
                 neg_chunk_size=None, seed=0, shuffle=True):
        from . import _lib
        self.dev = th.device(device)
        if self.dev.type != 'cuda':
            raise _lib.KgeError("DeviceSampler needs a CUDA (HIP) device")
        self.B, self.N = int(batch_size), int(neg_sample_size)
        self.chunk = int(neg_chunk_size or neg_sample_size)
        if self.B % self.chunk:
            raise ValueError("batch_size should be divisible by the chunk size")
        self.C = self.B // self.chunk
        if 2 * self.B + self.C * self.N > self.MAX_ELEMENTS:
            raise _lib.KgeError("DeviceSampler handles 2*batch + chunks*neg <= %d elements per step; "
                                "use UniformChunkedSampler (host plan) for larger batches" % self.MAX_ELEMENTS)
        self.n_entities = int(n_entities)
        def put(x):       # triples may already live in HBM (generated or loaded there)
            if isinstance(x, th.Tensor):
                return x.to(self.dev, th.int64).contiguous()
            return th.as_tensor(np.asarray(x, np.int64)).to(self.dev)
        self.H, self.R, self.T = put(heads), put(rels), put(tails)
        self.n_train = int(self.H.shape[0])
        self.seed = int(seed)
        g = th.Generator(device=self.dev)
        g.manual_seed(self.seed)
        self.perm = th.randperm(self.n_train, device=self.dev, generator=g) if shuffle else None
        self.state = th.tensor([0, 1, 0, 0, -1, 0, 0, 0], dtype=th.int64, device=self.dev)   # {position, step (1-based), ticket, -; cached epoch, mul, add, 1/n (tail jobs)}
        self.n_slots = int(n_slots)
        self.slot_bytes = int(_lib.lib().kge_sampler_slot_bytes(self.B, self.C, self.N))
        self.slots = th.zeros(self.n_slots * self.slot_bytes, dtype=th.uint8, device=self.dev)
        self.host_step = 1                                                     # next step to be sampled
        self._batches = {}                                                     # (slot, corrupt-head) -> DeviceBatch
        self.launches = 0                                                      # sample() calls so far (a batch's `gen`)

    def sample(self, n=None, slot0=0):
        """enqueue the construction of the next `n` (default: all slots) batches into slots slot0..slot0+n-1;
        returns the DeviceBatch objects (their neg_head flag follows the step parity).  `slot0` lets a caller
        double-buffer: sample the next group of batches (on another stream) while the current group trains."""
        from . import _lib
        n = self.n_slots if n is None else int(n)
        slot0 = int(slot0)
        if slot0 < 0 or slot0 + n > self.n_slots:
            raise ValueError("more batches than slots")
        _lib.check(_lib.lib().kge_sample_batches(
            _lib.ptr(self.H), _lib.ptr(self.R), _lib.ptr(self.T),
            _lib.ptr(self.perm) if self.perm is not None else None, self.n_train, self.n_entities, self.B,
            self.C, self.chunk, self.N, self.seed, _lib.ptr(self.state),
            self.slots.data_ptr() + slot0 * self.slot_bytes, self.slot_bytes, n, _lib.stream_ptr()))
        # (a DeviceBatch is pointer arithmetic on a slot: built once per slot and corruption mode - a group of 120 fresh ones was
        #  ~1 ms of ctypes calls in front of every group of the eager multi-GPU step)
        out = []
        for k in range(n):
            key = (slot0 + k, (self.host_step + k) % 2 == 0)
            b = self._batches.get(key)
            if b is None:
                b = self._batches[key] = DeviceBatch(self, key[0], key[1])
            # the SAME object comes back whenever its slot is refilled: `gen` tells consumers that cache per-batch work keyed by the
            # object (DistEngine's routing) which filling they are looking at
            b.gen = self.launches + 1
            out.append(b)
        self.launches += 1
        self.host_step += n
        return out

    # elements (2 * batch + chunks * neg) of a batch the sampler LAUNCH builds (csrc/kge_sampler_common.hpp SP_MAXE_BIG: the reference's
    # batch-2048 recipes are 6144) and the tail jobs that ride on a step's launches (SP_MAXE)
    MAX_ELEMENTS = 8192
    MAX_ELEMENTS_TAIL = 4096

    def prepare_tail(self):
        """what tail_jobs needs once: the scratch buffer and - the jobs read the triples in base-permutation order, their first phase
        is a chain of dependent memory rounds under a 9-us launch and perm[e] was one of them - permuted copies of the triples.
        NEVER inside a graph capture: the three gather kernels would become part of the graph and run with every replay (they did:
        24 us per replayed group, profiles/r05_sampler_tail.txt) - callers that capture call this first.
        HBM cost: the permuted copies of H, R and T are 24 bytes per training triple (FB15k: 11.6 MB; 338 M Freebase edges: 8.1 GB),
        held for the sampler's lifetime."""
        from . import _lib
        if getattr(self, "_tail_scratch", None) is not None:
            return
        if th.cuda.is_current_stream_capturing():
            raise _lib.KgeError("DeviceSampler.prepare_tail() must run before the graph capture that uses tail_jobs()")
        if 2 * self.B + self.C * self.N > self.MAX_ELEMENTS_TAIL:
            raise _lib.KgeError("sampler tail jobs handle 2*batch + chunks*neg <= %d elements per step (larger batches: the sampler "
                                "launch)" % self.MAX_ELEMENTS_TAIL)
        nb = int(_lib.lib().kge_sampler_tail_scratch_bytes(self.B, self.C, self.N, self.n_entities))
        if self.perm is not None:
            self._Hp, self._Rp, self._Tp = self.H[self.perm].contiguous(), self.R[self.perm].contiguous(), self.T[self.perm].contiguous()
        self._tail_scratch = th.zeros(nb, dtype=th.uint8, device=self.dev)

    def tail_jobs(self, n, slot0=0):
        """the next `n` batches as JOBS for the training steps' own launches instead of a launch of their own (round 5,
        kge_step_fused_sampling): returns (jobs, batches) - jobs[k] is handed to the step that should build batches[k] (any n steps
        of the current group, CONSECUTIVE steps in order: the step of job k + 1 also finishes batch k; the last job finishes its own
        batch and advances the device state), batches may be trained on once the step of the last job has run.  Same ids and plan, bit for bit, as sample(n, slot0) would give from the same state."""
        from . import _lib
        import ctypes as C
        n, slot0 = int(n), int(slot0)
        if n <= 0 or slot0 < 0 or slot0 + n > self.n_slots:
            raise ValueError("more batches than slots")
        self.prepare_tail()
        Hs, Rs, Ts = (self._Hp, self._Rp, self._Tp) if self.perm is not None else (self.H, self.R, self.T)
        jobs, out = [], []
        for k in range(n):
            j = _lib.KgeSamplerJob()
            j.heads, j.rels, j.tails = _lib.ptr(Hs), _lib.ptr(Rs), _lib.ptr(Ts)
            j.pre_permuted = 1 if self.perm is not None else 0
            j.perm = _lib.ptr(self.perm) if self.perm is not None else None
            j.n_train, j.n_ent = self.n_train, self.n_entities
            j.B, j.C, j.chunk, j.N, j.seed = self.B, self.C, self.chunk, self.N, self.seed
            j.state = _lib.ptr(self.state)
            j.slot = self.slots.data_ptr() + (slot0 + k) * self.slot_bytes
            j.scratch, j.scratch_bytes = _lib.ptr(self._tail_scratch), self._tail_scratch.numel()
            j.k, j.advance = k, (n if k == n - 1 else 0)
            j.prev_slot = (self.slots.data_ptr() + (slot0 + k - 1) * self.slot_bytes) if k > 0 else None
            jobs.append(j)
            key = (slot0 + k, (self.host_step + k) % 2 == 0)
            b = self._batches.get(key)
            if b is None:
                b = self._batches[key] = DeviceBatch(self, key[0], key[1])
            b.gen = self.launches + 1
            out.append(b)
        self.launches += 1
        self.host_step += n
        return jobs, out

    def slot_arrays(self, slot):
        """copy one slot back to the host as numpy arrays (tests / debugging)."""
        B, CN = self.B, self.C * self.N
        NE = 2 * B + CN
        raw = self.slots[slot * self.slot_bytes:(slot + 1) * self.slot_bytes].cpu().numpy()

        def al(x):
            return (x + 31) & ~31
        o, out = 0, {}
        for name, n, dt in (("h_gid", B, np.int64), ("t_gid", B, np.int64), ("rel_ids", B, np.int64),
                            ("neg_ids", CN, np.int64), ("ue_id", NE, np.int64), ("ur_id", B, np.int64),
                            ("ue_pos_ptr", NE + 1, np.int32), ("ue_pos_adj", 2 * B, np.int32),
                            ("ue_neg_ptr", NE + 1, np.int32), ("ue_neg_slot", CN, np.int32),
                            ("ur_ptr", B + 1, np.int32), ("ur_edge", B, np.int32),
                            ("ue_rec", 8 * NE, np.int32), ("ur_rec", 8 * B, np.int32), ("counts", 4, np.int32)):
            nb = n * np.dtype(dt).itemsize
            out[name] = raw[o:o + nb].view(dt).copy()
            o = al(o + nb)
        return out


class _GraphSeq(object):
    """hipGraphs replayed one after the other (a group of steps cut in two, PrefetchedGroups.head)"""

    def __init__(self, graphs):
        self.graphs = graphs

    def replay(self):
        for g in self.graphs:
            g.replay()


class PrefetchedGroups(object):
    """Groups of training steps over a double-buffered slot array: while group g trains, the batches of group g+1 are built
    into the other half (the prefetching of the reference's sampler workers, dataloader/sampler.py:823-876:
    `NewBidirectionalOneShotIterator` over `num_workers` sampler threads).  Where the sampler launch of group g+1 runs:
      'serial'  - behind group g's steps, on the same stream (one launch per group; the steps replay from a hipGraph);
      'streams' - on a second stream next to the steps, joined by an event at the end of the group;
      'fork'    - on a second branch inside the group's hipGraph;
      'fused'   - (round 5) NO launch: step k of group g builds batch k of group g + 1 with a few tail workgroups on its own
                  launches (kge_step_fused_sampling; step_fn must accept sample_job=); a next group larger than the current one
                  is sampled by a launch like 'serial';
      'fork_tail' - the same as 'fork', but the branch forks in front of the group's LAST step only (round 5: the second queue is then active for
                  one step instead of the whole group).
    Measured on MI355X / ROCm 7.0 (profiles/r03_merged_fwd.txt): the two concurrent modes hide the ~45 us launch but make every
    STEP ~3.5 us slower (a second active queue next to the graph's), so 'serial' is what bench.py uses.

    sampler: a DeviceSampler with n_slots >= 2 * the largest group; step_fn(batch): enqueues one training step."""

    def __init__(self, sampler, step_fn, group_max=None, mode="serial", fused_max=64, head=0):
        self.smp, self.step_fn = sampler, step_fn
        self.half = sampler.n_slots // 2 if group_max is None else int(group_max)
        if 2 * self.half > sampler.n_slots:
            raise ValueError("the sampler needs 2 x group_max slots")
        if mode not in ("streams", "fork", "fork_tail", "serial", "fused"):
            raise ValueError("mode: streams | fork | fork_tail | serial | fused")
        self.mode = mode
        # 'fused': groups of more steps than this keep the launch.  Back-to-back groups, us/step launch vs tail (tools/ab_long.py,
        # profiles/r05_sampler_tail.txt): 20 steps 32.26 / 31.38, 40: 31.52 / 31.22, 60: 31.25 / 31.16, 120: 31.02 / 31.06
        self.fused_max = int(fused_max)
        self.head = int(head)         # > 0: a group's graph is cut into [head steps | rest] (see run)
        if mode == "fused":
            sampler.prepare_tail()    # (outside any capture)
        self.side = th.cuda.Stream(device=sampler.dev)      # (its priority makes no difference: profiles/r03_merged_fwd.txt)
        self.buf = 0                  # half holding the batches of the NEXT group to train
        self.ready = None             # DeviceBatch objects in that half
        self.graphs = {}
        # how the NEXT group's batches were built, per run() call since the last reset_stats(): by tail workgroups of the steps'
        # own launches ('fused'), by a sampler launch ('launch': serial / streams / fork), or not at all (n_next == 0)
        self.stats = {"fused": 0, "launch": 0, "none": 0}

    def reset_stats(self):
        for k in self.stats:
            self.stats[k] = 0

    def prefill(self, n):
        """build the first group's batches on the current stream (outside any graph)."""
        self.ready = self.smp.sample(n, slot0=self.buf * self.half)

    def _sample_next(self, n_next):
        return self.smp.sample(n_next, slot0=(self.buf ^ 1) * self.half)

    def _enqueue_fork(self, n_next):
        cur = th.cuda.current_stream(self.smp.dev)
        batches, nxt = self.ready, None
        late = self.mode == "fork_tail" and len(batches) > 1

        def fork():
            self.side.wait_stream(cur)                                  # fork
            with th.cuda.stream(self.side):
                return self._sample_next(n_next)
        if n_next and not late:
            nxt = fork()
        for k, b in enumerate(batches):
            if n_next and late and k == len(batches) - 1:
                nxt = fork()
            self.step_fn(b)
        if n_next:
            cur.wait_stream(self.side)                                  # join
        return nxt

    def run(self, n_next, graph=True):
        """train the ready group and (concurrently) build the next one of n_next batches.  graph=True: the steps replay
        from a hipGraph captured on first use per (group size, buffer half)."""
        n_cur = len(self.ready)
        if n_next > self.half:
            raise ValueError("group larger than half of the slots")
        # (graph keys carry the corruption parity of the group's first batch and of the next group's: the cached DeviceBatch objects
        #  of a replayed graph must be the ones sampler.host_step would hand out now - an odd-sized group flips the parity)
        par = (bool(self.ready[0].neg_head) if n_cur else None, self.smp.host_step % 2)
        if self.mode in ("fork", "fork_tail"):
            self.stats["launch" if n_next else "none"] += 1
            key = (n_cur, n_next, self.buf, par)
            if not graph:
                nxt = self._enqueue_fork(n_next)
            elif key in self.graphs:
                g, nxt = self.graphs[key]
                self.smp.host_step += n_next
                g.replay()
            else:
                g = th.cuda.CUDAGraph()
                with _lib.graph_capture(g):
                    nxt = self._enqueue_fork(n_next)
                self.graphs[key] = (g, nxt)
                g.replay()
            self.ready = nxt
            self.buf ^= 1
            return
        if self.mode == "fused" and 0 < n_next <= n_cur <= self.fused_max:
            self.stats["fused"] += 1
            key = (n_cur, n_next, self.buf, par)
            if graph and key in self.graphs:
                g, nxt, _ = self.graphs[key]
                self.smp.host_step += n_next
                self.smp.launches += 1
                for b in nxt:
                    b.gen = self.smp.launches
                g.replay()
            else:
                def enqueue():
                    jobs, nb = self.smp.tail_jobs(n_next, slot0=(self.buf ^ 1) * self.half)
                    for k, b in enumerate(self.ready):
                        self.step_fn(b, sample_job=jobs[k] if k < n_next else None)
                    return nb, jobs
                if graph:
                    # `head` > 0: the group as TWO graphs, [first `head` steps | rest] - after a pause of the queue the GPU starts
                    # when the first graph's packets are written instead of the whole group's
                    head = self.head if 0 < self.head and 2 * self.head <= n_cur else 0
                    jobs, nxt = self.smp.tail_jobs(n_next, slot0=(self.buf ^ 1) * self.half)
                    parts = [(0, head), (head, n_cur)] if head else [(0, n_cur)]
                    gs = []
                    for lo, hi in parts:
                        g1 = th.cuda.CUDAGraph()
                        with _lib.graph_capture(g1):
                            for k in range(lo, hi):
                                self.step_fn(self.ready[k], sample_job=jobs[k] if k < n_next else None)
                        gs.append(g1)
                    g = _GraphSeq(gs)
                    self.graphs[key] = (g, nxt, jobs)      # (the job structs live as long as the graph that was recorded with them)
                    g.replay()
                else:
                    nxt, _ = enqueue()
            self.ready = nxt
            self.buf ^= 1
            return
        self.stats["launch" if n_next else "none"] += 1
        cur = th.cuda.current_stream(self.smp.dev)
        nxt = None
        if n_next and self.mode == "streams":
            self.side.wait_stream(cur)          # the half it overwrites was read by the group before this one
            with th.cuda.stream(self.side):
                nxt = self._sample_next(n_next)
        if not graph:
            for b in self.ready:
                self.step_fn(b)
        else:
            key = (n_cur, self.buf, par[0])
            if key not in self.graphs:
                head = self.head if 0 < self.head and 2 * self.head <= n_cur else 0
                gs = []
                for lo, hi in ([(0, head), (head, n_cur)] if head else [(0, n_cur)]):
                    g1 = th.cuda.CUDAGraph()
                    with _lib.graph_capture(g1):
                        for b in self.ready[lo:hi]:
                            self.step_fn(b)
                    gs.append(g1)
                self.graphs[key] = _GraphSeq(gs)
            self.graphs[key].replay()
        if n_next and self.mode in ("serial", "fused"):
            # 'serial': the sampler launch BEHIND the group's steps on the same stream (it fills the other half of the slots): the
            # GPU starts on the steps at once and the host builds the next group's batch descriptors while they run
            nxt = self._sample_next(n_next)
        if n_next and self.mode == "streams":
            cur.wait_stream(self.side)          # join: the next group's batches are complete
        self.ready = nxt
        self.buf ^= 1
