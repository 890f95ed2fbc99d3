"""Ranking evaluation on the device (libkge_hip `kge_rank_eval`): the filtered MRR / MR / HITS@k
protocol of the reference's `test()` loop (train_pytorch.py:199-253) - every test triple against
ALL entities as corrupted heads and as corrupted tails (EvalSampler, dataloader/sampler.py:514-597,
`--neg_sample_size_eval -1`), triples that exist in the graph masked out when `eval_filter`
(general_models.py:463-475) - without the per-triple Python loop of `forward_test`.

The host part (this file) only builds the filter lists once per dataset; scoring and counting run
in HIP.  There is no CPU path."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def build_filter(known_h, known_r, known_t, test_h, test_r, test_t, neg_head, n_relations=None):
    """filter lists for `kge_rank_eval`: for test triple i the entities e such that the triple with
    its head (neg_head) / tail replaced by e is a KNOWN triple (train + valid + test, like the
    graph the reference's EvalSampler draws `false_neg` from).  Triples sharing (r,t) / (h,r) share
    one list.  Returns (rng [E,2] int64, ids [M] int64) numpy arrays."""
    known_h, known_r, known_t = (np.asarray(x, np.int64) for x in (known_h, known_r, known_t))
    test_h, test_r, test_t = (np.asarray(x, np.int64) for x in (test_h, test_r, test_t))
    R = int(n_relations if n_relations is not None else max(known_r.max(initial=0), test_r.max(initial=0)) + 1)
    if neg_head:
        key, val = known_t * R + known_r, known_h
        tkey = test_t * R + test_r
    else:
        key, val = known_h * R + known_r, known_t
        tkey = test_h * R + test_r
    order = np.lexsort((val, key))
    key, val = key[order], val[order]
    if key.shape[0]:
        keep = np.ones(key.shape[0], bool)
        keep[1:] = (key[1:] != key[:-1]) | (val[1:] != val[:-1])      # unique (key, entity) pairs
        key, val = key[keep], val[keep]
    rng = np.stack([np.searchsorted(key, tkey, "left"), np.searchsorted(key, tkey, "right")], 1).astype(np.int64)
    return rng, val.astype(np.int64)


_FORCE_TWO_KEY_SORT = False      # tests: take the large-graph path of build_filter_device on a small graph


def build_filter_device(known, test, neg_head, n_relations, n_entities, dev):
    """`build_filter` with the sort on the device: the same lists in the same order (unique (key, entity) pairs sorted by key, then
    entity; per test triple the [left, right) range of its key), as int64 DEVICE tensors ready for `Ranker.ranks`.  One composite
    key `(key * n_entities + entity)` through `torch.unique` instead of a host `np.lexsort` over every known triple - 0.09 s per
    corruption side at FB15k's 592 k known triples, which was 87 % of a validation (tools/eval_timing.py).  When the composite key
    does not fit int64 (Freebase: 86 M entities x 14 824 relations x 86 M) the same order comes from two stable device sorts
    (entity, then key).  Returns None only when even `key` would overflow."""
    R, NE = int(n_relations), int(n_entities)
    if NE * R >= (1 << 62):
        return None
    two_key = NE * R * NE >= (1 << 62) or _FORCE_TWO_KEY_SORT

    def put(x):
        if isinstance(x, torch.Tensor):
            return x.to(dev, torch.int64)
        return torch.as_tensor(np.ascontiguousarray(np.asarray(x, np.int64))).to(dev)
    kh, kr, kt = (put(x) for x in known)
    th_, tr_, tt_ = (put(x) for x in test)
    if neg_head:
        key, val, tkey = kt * R + kr, kh, tt_ * R + tr_
    else:
        key, val, tkey = kh * R + kr, kt, th_ * R + tr_
    if two_key:
        o = torch.argsort(val, stable=True)
        o = o[torch.argsort(key[o], stable=True)]            # lexicographic (key, entity) order
        key, val = key[o], val[o]
        if key.shape[0]:
            keep = torch.ones(key.shape[0], dtype=torch.bool, device=key.device)
            keep[1:] = (key[1:] != key[:-1]) | (val[1:] != val[:-1])
            key, val = key[keep], val[keep]
    else:
        comp = torch.unique(key * NE + val)                  # sorted unique (key, entity) pairs
        key = torch.div(comp, NE, rounding_mode='floor')
        val = comp - key * NE
    rng = torch.stack([torch.searchsorted(key, tkey, right=False), torch.searchsorted(key, tkey, right=True)], 1)
    return rng.contiguous(), val.contiguous()


class Ranker(object):
    """device-resident evaluation of one (ent, rel) table pair."""

    def __init__(self, model_name, ent, rel, gamma, emb_init, batch=1024, flags=0, proj=None):
        if not ent.is_cuda:
            raise _lib.KgeError("Ranker needs CUDA (HIP) tensors; there is no CPU path")
        self.model = _lib.model_id(model_name)
        self.ent, self.rel = ent, rel
        self.gamma, self.emb_init = float(gamma), float(emb_init)
        self.batch = int(batch)
        self.flags = int(flags)
        self.proj = proj                   # TransR: projection table [n_rel, d_e * d_r]
        if model_name == 'TransR' and proj is None:
            raise _lib.KgeError("TransR ranking needs the projection table (proj=...)")
        self._ws = None
        self._all = None

    def ranks(self, h, r, t, neg_head, filt=None, cand=None, want_pos_score=False):
        """int32 [E] ranks of the true triples among the corruptions of the chosen side."""
        dev = self.ent.device

        def put(x, dt=torch.int64):
            if x is None:
                return None
            if isinstance(x, torch.Tensor):
                return x.to(dev, dt).contiguous()
            return torch.as_tensor(np.ascontiguousarray(x)).to(dev, dt)
        h, r, t, cand = put(h), put(r), put(t), put(cand)
        E = int(h.shape[0])
        if cand is None and self.proj is not None:         # TransR kernels walk an explicit candidate list
            if self._all is None:
                self._all = torch.arange(self.ent.shape[0], dtype=torch.int64, device=dev)
            cand = self._all
        n_cand = int(cand.shape[0]) if cand is not None else int(self.ent.shape[0])
        frng = fids = None
        if filt is not None:
            frng, fids = put(filt[0].reshape(-1)), put(filt[1])
            if fids.shape[0] == 0:
                fids = torch.zeros(1, dtype=torch.int64, device=dev)
        Eb = max(1, min(self.batch, E))
        need = _lib.lib().kge_rank_workspace_bytes(Eb, n_cand, self.ent.shape[1])
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need + 4096, dtype=torch.uint8, device=dev)
        ranks = torch.zeros(E, dtype=torch.int32, device=dev)
        pos = torch.empty(E, dtype=torch.float32, device=dev) if want_pos_score else None
        _lib.check(_lib.lib().kge_rank_eval_ex(
            self.model, int(bool(neg_head)), _lib.ptr(self.ent), self.ent.shape[0], _lib.ptr(self.rel),
            self.rel.shape[0], _lib.ptr(self.proj), _lib.ptr(h), _lib.ptr(r), _lib.ptr(t), E, self.ent.shape[1],
            self.rel.shape[1],
            self.gamma, self.emb_init, _lib.ptr(cand), n_cand, _lib.ptr(frng), _lib.ptr(fids), Eb,
            _lib.ptr(ranks), _lib.ptr(pos), _lib.ptr(self._ws), self._ws.numel(), self.flags, _lib.stream_ptr()))
        return (ranks, pos) if want_pos_score else ranks


def metrics_from_ranks(ranks):
    """the averages `test()` prints (train_pytorch.py:236-247): MRR, MR, HITS@1/3/10."""
    rk = ranks.to(torch.float64)
    return {"MRR": float((1.0 / rk).mean()), "MR": float(rk.mean()),
            "HITS@1": float((rk <= 1).double().mean()), "HITS@3": float((rk <= 3).double().mean()),
            "HITS@10": float((rk <= 10).double().mean())}


def sampled_ranks(rk, h, r, t, neg_head, filt, n_entities, n_cand, chunk, rng, cand_of_chunk=None):
    """`--neg_sample_size_eval n_cand` (EvalSampler with a negative sample size below the entity count,
    dataloader/sampler.py:514-597): every chunk of `chunk` test triples is ranked against ITS OWN n_cand candidates
    drawn uniformly with replacement from all entities; with a filter, candidates that make a known triple do not
    count (the sampler's false-negative bias, general_models.py:463-475).  rank = 1 + #{unfiltered candidates scoring
    >= the true triple}.  cand_of_chunk(k): override of the draw (tests)."""
    h, r, t = (np.asarray(x, np.int64) for x in (h, r, t))
    E = h.shape[0]
    out = []
    for k, e0 in enumerate(range(0, E, chunk)):
        e1 = min(E, e0 + chunk)
        cand = cand_of_chunk(k) if cand_of_chunk is not None else rng.randint(0, n_entities, size=n_cand)
        cand = np.asarray(cand, np.int64)
        f = None
        if filt is not None:
            # columns of this chunk's candidate list that hold a filtered entity, per triple (duplicates in the draw
            # are separate columns): sorted candidates + two binary searches per filtered id
            order = np.argsort(cand, kind="stable")
            sc = cand[order]
            frng, fids = filt
            ptr, cols = [0], []
            for i in range(e0, e1):
                ids = fids[frng[i, 0]:frng[i, 1]]
                lo, hi = np.searchsorted(sc, ids, "left"), np.searchsorted(sc, ids, "right")
                hit = [order[a:b] for a, b in zip(lo, hi) if b > a]
                c = np.concatenate(hit) if hit else np.zeros(0, np.int64)
                cols.append(c)
                ptr.append(ptr[-1] + c.shape[0])
            ptr = np.asarray(ptr, np.int64)
            f = (np.stack([ptr[:-1], ptr[1:]], 1), np.concatenate(cols) if cols else np.zeros(0, np.int64))
        out.append(rk.ranks(h[e0:e1], r[e0:e1], t[e0:e1], neg_head, f, cand=cand))
    return torch.cat(out)


def evaluate(model_name, ent, rel, gamma, emb_init, test, known=None, batch=1024, modes=("head", "tail"), proj=None,
             n_cand=None, chunk=None, seed=0, cache=None):
    """filtered (known given) or raw ranking metrics over both corruption modes, averaged over all
    2E rankings like the reference (logs of the head and the tail sampler are concatenated,
    train_pytorch.py:221-231).  test / known: (h, r, t) triples of int64 arrays.  n_cand (< number of entities):
    rank against n_cand sampled candidates per chunk of `chunk` triples instead of all entities.
    cache: a dict the caller keeps per (split, known set) - the filter lists and the test triples stay on the device between
    calls (a training run validates the SAME split against the SAME known triples every --eval_interval steps)."""
    rk = Ranker(model_name, ent, rel, gamma, emb_init, batch, proj=proj)
    th_, tr_, tt_ = test
    n_ent = int(ent.shape[0])
    sampled = n_cand is not None and 0 < n_cand < n_ent
    rng = np.random.RandomState(seed)
    dev = ent.device
    if cache is not None and not sampled:
        if "test" not in cache:
            cache["test"] = tuple(torch.as_tensor(np.ascontiguousarray(np.asarray(x, np.int64))).to(dev) for x in test)
        th_, tr_, tt_ = cache["test"]
    allr = []
    for mode in modes:
        neg_head = mode == "head"
        filt = None
        if known is not None:
            if cache is not None and ("filt", mode, sampled) in cache:
                filt = cache[("filt", mode, sampled)]
            else:
                # lists built on the device - and kept there, unless candidates are sampled: then the host maps them to columns
                filt = build_filter_device(known, (th_, tr_, tt_), neg_head, rel.shape[0], n_ent, dev)
                if sampled and filt is not None:
                    filt = (filt[0].cpu().numpy(), filt[1].cpu().numpy())
                if filt is None:
                    filt = build_filter(known[0], known[1], known[2], *(x.cpu().numpy() if isinstance(x, torch.Tensor) else x
                                                                        for x in (th_, tr_, tt_)), neg_head, rel.shape[0])
                if cache is not None:
                    cache[("filt", mode, sampled)] = filt
        if sampled:
            allr.append(sampled_ranks(rk, th_, tr_, tt_, neg_head, filt, n_ent, int(n_cand), int(chunk or batch), rng))
        else:
            allr.append(rk.ranks(th_, tr_, tt_, neg_head, filt))
    return metrics_from_ranks(torch.cat(allr))
