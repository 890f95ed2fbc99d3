"""Batch objects with the members the reference reads from DGL's sampler output
(dataloader/sampler.py:421-457, 862-876; models/general_models.py:376-427, 548-549), backed by
flat id tensors instead of DGL subgraphs, plus a uniform chunked negative sampler that restates
the EdgeSampler call of dataloader/sampler.py:408-419 (negatives uniform over all entities, with
replacement, positives not excluded, C*N = B corrupt ids per batch) and the alternating
head/tail iterator of dataloader/sampler.py:823-876."""
import numpy as np
import torch as th

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
        emb = self.ndata['emb']
        self.edata.update(fn(_Edges({'emb': emb[self._h]}, {'emb': emb[self._t]}, self.edata)))


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
