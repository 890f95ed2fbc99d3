"""Ranking evaluation on the device (kge_rank_eval through dglke_amd.eval) against (a) the rankings
recorded from the reference's forward_test and (b) the oracle on larger seeded cases.

fp32 scores of the HIP kernels differ from the reference's in the last bits, so a candidate whose
score is within `TOL` of the true triple's may fall on either side of `>=`: the bar is
lo <= rank <= hi with lo / hi the ranks under scores shifted by -/+ TOL (oracle rank_eval), plus
exact equality whenever lo == hi (the vast majority)."""
import os

import numpy as np
import pytest
import torch

from golden_util import eval_golden_names, load_golden
from oracle import kge_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4           # BASELINE.json north_star: 1e-4 on fp32 scores


_RANK_STATS = {}


def _filt_from_mask(mask):
    rng, ids, o = [], [], 0
    for i in range(mask.shape[0]):
        cols = np.nonzero(mask[i])[0]
        rng.append((o, o + len(cols)))
        ids.extend(cols.tolist())
        o += len(cols)
    return np.array(rng, np.int64).reshape(-1, 2), np.array(ids, np.int64)


@pytest.mark.parametrize("name", eval_golden_names())
@pytest.mark.parametrize("flags", [0, 1])
def test_rank_eval_matches_reference_rankings(name, flags):
    from dglke_amd import eval as E
    z, case = load_golden(name)
    ent = torch.from_numpy(z["entity"]).to(DEV)
    rel = torch.from_numpy(z["relation"]).to(DEV)
    test, known = z["test"], z["known"]
    h, r, t = test[:, 0], test[:, 1], test[:, 2]
    proj = torch.from_numpy(z["projection"]).to(DEV) if case["model"] == "TransR" else None
    proj64 = z["projection"].astype(np.float64) if case["model"] == "TransR" else None
    rk = E.Ranker(case["model"], ent, rel, case["gamma"], float(z["emb_init"]), batch=5, flags=flags, proj=proj)
    for mode in ("head", "tail"):
        neg_head = mode == "head"
        filt = E.build_filter(known[:, 0], known[:, 1], known[:, 2], h, r, t, neg_head, rel.shape[0])
        # the host filter lists equal the reference's false-negative mask
        mask = z[mode + "_false_neg"] > 0
        for i in range(len(h)):
            assert np.array_equal(filt[1][filt[0][i, 0]:filt[0][i, 1]], np.nonzero(mask[i])[0])
        ranks, pos = rk.ranks(h, r, t, neg_head, filt, want_pos_score=True)
        np.testing.assert_allclose(pos.cpu().numpy(), z[mode + "_pos_score"], rtol=1e-4, atol=1e-4)
        ent64, rel64 = z["entity"].astype(np.float64), z["relation"].astype(np.float64)
        for f, key in ((filt, "filtered"), (None, "raw")):
            got = rk.ranks(h, r, t, neg_head, f).cpu().numpy()
            (lo, hi), _, _ = O.rank_eval(case["model"], ent64, rel64, h, r, t, neg_head, case["gamma"],
                                         float(z["emb_init"]), mask if f is not None else None, tol=TOL, proj=proj64)
            want = z["%s_ranks_%s" % (mode, key)]
            assert np.all((lo <= got) & (got <= hi)), (mode, key, lo, got, hi)
            exact = lo == hi
            assert np.array_equal(got[exact], want[exact]), (mode, key, got, want)
            # the record VERDICT r03 asked for: how many rankings are pinned exactly (degenerate interval under +-1e-4 score
            # shifts) and how many GPU ranks equal the reference's overall
            _RANK_STATS["%s flags=%d %s %s" % (name, flags, mode, key)] = (int(exact.sum()), int(len(exact)), int((got == want).sum()))
            try:
                out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
                os.makedirs(out, exist_ok=True)
                with open(os.path.join(out, "rank_eval_exactness.txt"), "w") as fh:
                    te = sum(v[0] for v in _RANK_STATS.values()); tn = sum(v[1] for v in _RANK_STATS.values())
                    tq = sum(v[2] for v in _RANK_STATS.values())
                    fh.write("ranking evaluation vs the reference's forward_test rankings (9 goldens x {MFMA, pairwise} x {head, tail} x "
                             "{filtered, raw}):\n%d of %d rankings have a degenerate rank interval under +-1e-4 score shifts and are "
                             "asserted EXACTLY; %d of %d GPU ranks equal the reference's rank\n\n" % (te, tn, tq, tn))
                    for k in sorted(_RANK_STATS):
                        fh.write("%-52s exact-interval %3d / %3d   rank == reference %3d / %3d\n" % ((k,) + _RANK_STATS[k][:2] + (_RANK_STATS[k][2], _RANK_STATS[k][1])))
            except OSError:
                pass
        assert np.array_equal(ranks.cpu().numpy(), rk.ranks(h, r, t, neg_head, filt).cpu().numpy())


@pytest.mark.parametrize("model,de_,dr_,hidden,n_ent", [("TransE_l2", False, False, 400, 14951),
                                                        ("DistMult", False, False, 100, 3000),
                                                        ("ComplEx", True, True, 50, 2000),
                                                        ("RotatE", True, False, 32, 1500),
                                                        ("TransE_l1", False, False, 40, 1500)])
def test_rank_eval_matches_oracle_at_scale(model, de_, dr_, hidden, n_ent):
    """FB15k-sized candidate set for the headline model; random tables, 96 test triples, candidate
    subset variant, metrics helper."""
    from dglke_amd import eval as E
    rng = np.random.RandomState(5)
    n_rel, Et = 11, 96
    d_e = 2 * hidden if de_ else hidden
    d_r = 2 * hidden if dr_ else hidden
    gamma = 12.0
    emb_init = (gamma + 2.0) / hidden
    ent = (rng.rand(n_ent, d_e).astype(np.float32) - 0.5) * 2 * emb_init * 3
    rel = (rng.rand(n_rel, d_r).astype(np.float32) - 0.5) * 2 * emb_init * 3
    known = np.stack([rng.randint(0, n_ent, 4000), rng.randint(0, n_rel, 4000), rng.randint(0, n_ent, 4000)], 1)
    test = known[:Et]
    h, r, t = test[:, 0], test[:, 1], test[:, 2]
    rk = E.Ranker(model, torch.from_numpy(ent).to(DEV), torch.from_numpy(rel).to(DEV), gamma, emb_init, batch=40)
    for neg_head in (False, True):
        filt = E.build_filter(known[:, 0], known[:, 1], known[:, 2], h, r, t, neg_head, n_rel)
        mask = np.zeros((Et, n_ent), bool)
        for i in range(Et):
            mask[i, filt[1][filt[0][i, 0]:filt[0][i, 1]]] = True
        got = rk.ranks(h, r, t, neg_head, filt).cpu().numpy()
        (lo, hi), _, _ = O.rank_eval(model, ent.astype(np.float64), rel.astype(np.float64), h, r, t, neg_head, gamma,
                                     emb_init, mask, tol=TOL)
        assert np.all((lo <= got) & (got <= hi)), (neg_head, np.nonzero((got < lo) | (got > hi)))
        assert (hi - lo).mean() < 0.005 * n_ent      # the tolerance band is narrow: the check is meaningful
    # candidate subset: ranks among 500 sampled candidates (neg_sample_size_eval > 0 protocol)
    cand = rng.choice(n_ent, 500, replace=False).astype(np.int64)
    got = rk.ranks(h, r, t, False, None, cand=cand).cpu().numpy()
    _, p, S = O.rank_eval(model, ent.astype(np.float64), rel.astype(np.float64), h, r, t, False, gamma, emb_init)
    lo = (S[:, cand] >= p[:, None] + TOL).sum(1) + 1
    hi = (S[:, cand] >= p[:, None] - TOL).sum(1) + 1
    assert np.all((lo <= got) & (got <= hi))
    m = E.metrics_from_ranks(torch.tensor([1, 2, 4, 20]))
    assert abs(m["MRR"] - (1 + 0.5 + 0.25 + 0.05) / 4) < 1e-12 and m["HITS@3"] == 0.5 and m["MR"] == 6.75


def test_rank_eval_argument_errors():
    from dglke_amd import _lib, eval as E
    ent = torch.zeros(10, 8, device=DEV)
    rel = torch.zeros(3, 8, device=DEV)
    with pytest.raises(_lib.KgeError):
        E.Ranker("TransE_l2", ent.cpu(), rel.cpu(), 1.0, 1.0)
    rk = E.Ranker("RotatE", ent, rel, 1.0, 1.0)          # RotatE needs d_r == d_e / 2
    with pytest.raises(_lib.KgeError):
        rk.ranks(np.array([0]), np.array([0]), np.array([1]), False)
    rk = E.Ranker("TransE_l2", ent, rel, 1.0, 1.0)
    assert rk.ranks(np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64), False).shape == (0,)


def test_sampled_candidate_ranking_is_consistent_with_full_ranking():
    """--neg_sample_size_eval: (a) with every chunk's candidates = a permutation of ALL entities the sampled path
    (candidate list + per-chunk filter columns) reproduces the full filtered ranking exactly; (b) with fewer
    candidates the rank is 1 + the number of unfiltered candidates that score >= the true triple (brute force)."""
    from dglke_amd import eval as kev
    rng = np.random.RandomState(5)
    n_ent, n_rel, D, E = 300, 7, 32, 90
    ent = torch.tensor(rng.uniform(-1, 1, (n_ent, D)).astype(np.float32), device=DEV)
    rel = torch.tensor(rng.uniform(-1, 1, (n_rel, D)).astype(np.float32), device=DEV)
    kh, kr, kt = rng.randint(0, n_ent, 4000), rng.randint(0, n_rel, 4000), rng.randint(0, n_ent, 4000)
    h, r, t = kh[:E].copy(), kr[:E].copy(), kt[:E].copy()
    rk = kev.Ranker("TransE_l2", ent, rel, 8.0, 0.3, batch=32)
    for neg_head in (False, True):
        filt = kev.build_filter(kh, kr, kt, h, r, t, neg_head, n_rel)
        full = rk.ranks(h, r, t, neg_head, filt).cpu().numpy()
        perms = [rng.permutation(n_ent) for _ in range(8)]
        got = kev.sampled_ranks(rk, h, r, t, neg_head, filt, n_ent, n_ent, 16, rng, cand_of_chunk=lambda k: perms[k]).cpu().numpy()
        assert np.array_equal(got, full)
        # (b) 40 candidates with replacement per chunk of 16
        draws = [rng.randint(0, n_ent, 40) for _ in range(8)]
        got = kev.sampled_ranks(rk, h, r, t, neg_head, filt, n_ent, 40, 16, rng, cand_of_chunk=lambda k: draws[k]).cpu().numpy()
        _, pos = rk.ranks(h, r, t, neg_head, None, want_pos_score=True)
        pos = pos.cpu().numpy()
        e64, r64 = ent.cpu().numpy().astype(np.float64), rel.cpu().numpy().astype(np.float64)
        for i in range(E):
            cand = draws[i // 16]
            x = e64[cand]
            sc = 8.0 - (np.linalg.norm(x + r64[r[i]] - e64[t[i]], axis=1) if neg_head else np.linalg.norm(e64[h[i]] + r64[r[i]] - x, axis=1))
            known = set(filt[1][filt[0][i, 0]:filt[0][i, 1]].tolist())
            keep = np.array([c not in known for c in cand])
            lo = 1 + int(((sc >= pos[i] + 1e-4) & keep).sum())
            hi = 1 + int(((sc >= pos[i] - 1e-4) & keep).sum())
            assert lo <= got[i] <= hi, (i, got[i], lo, hi)


def test_evaluate_with_a_cache_returns_the_same_metrics():
    """eval.evaluate(cache=...) - what dglke_train's validations use: filter lists built on the device once and kept there with the
    test ids - gives exactly the metrics of the uncached call with host-built lists, on the first call and on the calls that hit
    the cache (tables changed in between: only the lists are reused)."""
    from dglke_amd import eval as kev
    rng = np.random.RandomState(5)
    n_ent, n_rel, D = 3000, 40, 64
    known = tuple(rng.randint(0, n, 30000) for n in (n_ent, n_rel, n_ent))
    test = tuple(k[:2000] for k in known)
    torch.manual_seed(1)
    ent = torch.empty(n_ent, D, device=DEV).uniform_(-0.3, 0.3)
    rel = torch.empty(n_rel, D, device=DEV).uniform_(-0.3, 0.3)
    cache = {}
    for it in range(3):
        want = {}
        for mode in ("head", "tail"):      # the uncached reference: host lists, one mode at a time
            filt = kev.build_filter(known[0], known[1], known[2], test[0], test[1], test[2], mode == "head", n_rel)
            want[mode] = kev.Ranker("TransE_l2", ent, rel, 12.0, 0.3, 512).ranks(test[0], test[1], test[2], mode == "head", filt)
        m_want = kev.metrics_from_ranks(torch.cat([want["head"], want["tail"]]))
        m_got = kev.evaluate("TransE_l2", ent, rel, 12.0, 0.3, test, known, batch=512, cache=cache)
        assert m_got == m_want, (it, m_got, m_want)
        assert ("filt", "head", False) in cache and cache["test"][0].is_cuda
        ent.add_(torch.empty_like(ent).uniform_(-0.05, 0.05))      # "training" between the validations


@pytest.mark.parametrize("model,de_,dr_,hidden", [("TransE_l2", False, False, 36), ("DistMult", False, False, 100), ("SimplE", True, True, 20),
                                                  ("ComplEx", True, True, 66), ("RESCAL", False, False, 24)])
def test_tiled_rank_gemm_agrees_with_the_score_block_path(model, de_, dr_, hidden):
    """kge_rank_gemm.hip (128 x 128 tiles, one comparison bit per pair, ranks from the mask) against the score block +
    rank_count_kernel (flags = KGE_FLAG_FORCE_PAIRWISE: the pairwise kernels' fp32 scores): several row blocks with a partial
    last one, a candidate count that is no multiple of 128 (and one below a tile), widths that are no multiple of the 32-column
    stage, filtered and raw, explicit candidate lists with repeats.  fp32 sums in another order may flip a comparison within an ulp
    of the true triple's score: ranks equal on >= 99.5 % of the triples and never more than 2 apart; both inside the oracle's band."""
    from dglke_amd import eval as E
    rng = np.random.RandomState(11)
    n_ent, n_rel, Et = 1000, 7, 300
    d_e = 2 * hidden if de_ else hidden
    d_r = d_e * d_e if model == "RESCAL" else (2 * hidden if dr_ else hidden)
    gamma, emb_init = 12.0, 14.0 / hidden
    ent = (rng.rand(n_ent, d_e).astype(np.float32) - 0.5) * 2 * emb_init
    rel = (rng.rand(n_rel, d_r).astype(np.float32) - 0.5) * 2 * emb_init
    known = np.stack([rng.randint(0, n_ent, 6000), rng.randint(0, n_rel, 6000), rng.randint(0, n_ent, 6000)], 1)
    h, r, t = known[:Et, 0], known[:Et, 1], known[:Et, 2]
    te, tr = torch.from_numpy(ent).to(DEV), torch.from_numpy(rel).to(DEV)
    new, old = E.Ranker(model, te, tr, gamma, emb_init, batch=160), E.Ranker(model, te, tr, gamma, emb_init, batch=160, flags=1)
    for neg_head in (False, True):
        filt = E.build_filter(known[:, 0], known[:, 1], known[:, 2], h, r, t, neg_head, n_rel)
        # filtered: the true triple's own column is in the list, so its comparison cancels in both paths
        a, b = new.ranks(h, r, t, neg_head, filt).cpu().numpy(), old.ranks(h, r, t, neg_head, filt).cpu().numpy()
        assert (a != b).mean() <= 0.005 and np.abs(a - b).max() <= 2, (neg_head, np.nonzero(a != b)[0][:10])
        assert a.min() >= 1 and a.max() <= n_ent
        mask = np.zeros((Et, n_ent), bool)
        for i in range(Et):
            mask[i, filt[1][filt[0][i, 0]:filt[0][i, 1]]] = True
        # raw: the true entity is a candidate whose score equals the positive score up to rounding (a coin flip in any fp32
        # implementation, the reference's included): the oracle's tolerance band is the bar, as everywhere in this file
        for f, mk in ((filt, mask), (None, None)):
            (lo, hi), _, _ = O.rank_eval(model, ent.astype(np.float64), rel.astype(np.float64), h, r, t, neg_head, gamma, emb_init, mk, tol=TOL)
            got = new.ranks(h, r, t, neg_head, f).cpu().numpy()
            assert np.all((lo <= got) & (got <= hi)), (neg_head, f is None)
    for n_c in (70, 257):                  # below one tile; three column tiles with a 1-candidate tail; repeats in the list
        cand = rng.randint(0, n_ent, n_c).astype(np.int64)
        a, b = new.ranks(h, r, t, False, None, cand=cand).cpu().numpy(), old.ranks(h, r, t, False, None, cand=cand).cpu().numpy()
        own = np.array([(cand == t[i]).sum() for i in range(Et)])      # (copies of the true tail in the list: coin flips, see above)
        assert np.all(np.abs(a - b) <= own + 1) and (np.abs(a - b) > own).mean() <= 0.01 and a.max() <= n_c + 1
