"""Planted TransE knowledge graph (no network -> no real FB15k): entities and relations get hidden
ground-truth vectors and every triple is (h, r, argmin_t |h + r - t|), so a TransE model CAN fit it and
filtered MRR is a meaningful end-to-end signal (SURVEY.md 8d suggests exactly this for time-to-MRR)."""
import numpy as np


def make_planted(n_ent=2000, n_rel=12, n_edges=30000, dim=12, seed=0):
    rng = np.random.RandomState(seed)
    E = rng.randn(n_ent, dim).astype(np.float32)
    R = (rng.randn(n_rel, dim) * 1.5).astype(np.float32)
    h = rng.randint(0, n_ent, n_edges)
    r = rng.randint(0, n_rel, n_edges)
    t = np.empty(n_edges, np.int64)
    for s in range(0, n_edges, 4096):
        q = E[h[s:s + 4096]] + R[r[s:s + 4096]]
        d = ((q[:, None, :] - E[None, :, :]) ** 2).sum(-1)
        d[np.arange(q.shape[0]), h[s:s + 4096]] = np.inf      # no self loops
        t[s:s + 4096] = d.argmin(1)
    trip = np.unique(np.stack([h, r, t], 1), axis=0)
    rng.shuffle(trip)
    n_test = max(200, trip.shape[0] // 20)
    return trip[n_test:], trip[:n_test]


def filter_bias(all_trip, test, n_ent, corrupt_head):
    """bias[i, e] = -1 where replacing the tail (head) of test triple i by e gives a KNOWN triple other than
    the test triple itself (the `bias` edge data of the reference's EvalSampler, sampler.py:514-597)."""
    known = set(map(tuple, all_trip.tolist()))
    bias = np.zeros((test.shape[0], n_ent), np.float32)
    for i, (h, r, t) in enumerate(test.tolist()):
        for e in range(n_ent):
            cand = (e, r, t) if corrupt_head else (h, r, e)
            if cand in known and cand != (h, r, t):
                bias[i, e] = -1
    return bias
