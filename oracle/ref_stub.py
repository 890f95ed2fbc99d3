"""What makes the UNMODIFIED reference (awslabs/dgl-ke, python/dglke) importable and drivable without DGL.

*** TEST / MEASUREMENT INFRASTRUCTURE ONLY - never imported by the product (dgl-ke_amd/). ***

`dgl` and `ogb` are not installed, so stub modules are registered in sys.modules before anything of `dglke` is imported; the stub
provides the thin tensor shim `dgl.backend` and placeholder classes, no arithmetic.  Batches are duck-typed
objects exposing the members the reference reads of DGL's positive / negative subgraphs (general_models.py:376-427, 548-569):
PosG / NegG.  Used by tests/golden/gen_golden*.py (the golden vectors) and by oracle/ref_baseline.py (bench.py's CPU baseline of
kind "reference": the reference's own step timed on the host cores).  Everything here is this repository's own code.
"""
import sys
import types

import torch as th


def install_stubs():
    """Register fake dgl / ogb modules (members listed in SURVEY.md Appendix A)."""
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    dgl = mod("dgl")
    F = mod("dgl.backend")
    F.float32 = th.float32
    F.int64 = th.int64
    F.cpu = lambda: th.device("cpu")
    F.ones = lambda shape, dtype, ctx: th.ones(shape, dtype=dtype, device=ctx)
    F.context = lambda t: t.device
    F.cat = lambda seq, dim: th.cat(seq, dim=dim)
    F.copy_to = lambda t, ctx: t.to(ctx)
    F.tensor = lambda x, dtype=None: th.tensor(x, dtype=dtype)
    F.asnumpy = lambda t: t.detach().cpu().numpy()
    F.sum = lambda t, dim: th.sum(t, dim=dim)
    F.shape = lambda t: t.shape
    F.reshape = lambda t, s: t.reshape(s)
    F.arange = lambda a, b: th.arange(a, b)
    F.argsort = lambda t, dim, descending: th.argsort(t, dim=dim, descending=descending)
    dgl.backend = F
    dep = mod("dgl._deprecate")
    depg = mod("dgl._deprecate.graph")

    class DGLGraph(object):
        pass
    depg.DGLGraph = DGLGraph
    dep.graph = depg
    dgl._deprecate = dep
    base = mod("dgl.base")
    base.NID = "_ID"
    base.EID = "_ID"
    dgl.base = base
    contrib = mod("dgl.contrib")
    contrib.KVClient = object
    contrib.KVServer = object
    contrib.sampling = mod("dgl.contrib.sampling")
    dgl.contrib = contrib
    ogb = mod("ogb")
    lsc = mod("ogb.lsc")
    lsc.WikiKG90MDataset = object
    lsc.WikiKG90MEvaluator = object
    ogb.lsc = lsc


class Args(dict):
    """attribute dict, like the reference's own tests use (tests/test_score.py:45-49)."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


class _Edges(object):
    def __init__(self, src, dst, data):
        self.src, self.dst, self.data = src, dst, data


class PosG(object):
    """Duck type of the positive DGL subgraph (members: SURVEY.md section 8b)."""

    def __init__(self, nid, h_local, t_local, rel_id, impts=None):
        self.ndata = {"id": nid}
        self.edata = {"id": rel_id}
        if impts is not None:
            self.edata["impts"] = impts
        self._h, self._t = h_local, t_local

    def all_edges(self, order="eid"):
        return self._h, self._t

    def number_of_edges(self):
        return int(self._h.shape[0])

    def apply_edges(self, fn):
        e = _Edges({"emb": self.ndata["emb"][self._h]}, {"emb": self.ndata["emb"][self._t]},
                   self.edata)
        self.edata.update(fn(e))


class NegG(object):
    def __init__(self, ids, num_chunks, chunk_size, neg_sample_size, neg_head):
        self.ndata = {"id": ids}
        self.edata = {}
        n = ids.shape[0]
        self.head_nid = th.arange(n)
        self.tail_nid = th.arange(n)
        self.num_chunks = num_chunks
        self.chunk_size = chunk_size
        self.neg_sample_size = neg_sample_size
        self.neg_head = neg_head
