"""GPU parity tests (run with `-m gpu` on the MI355X box): the HIP path, called through the C ABI
(ctypes -> libkge_hip.so), against (a) the golden vectors recorded from the unmodified reference
and (b) the CPU oracle on seeded inputs, including the BASELINE config shapes.

Tolerances (fp32, BASELINE.json north_star: 1e-4 on fp32 scores): scores 1e-4 abs (+1e-4 rel),
loss 1e-5 rel/abs, gradients 3e-4 of the largest gradient component (fp32 cancellation noise of
the reference's own |a|^2+|b|^2-2ab formulation, see tests/test_oracle_golden.py), post-update
rows 5e-3*lr (Adagrad's first steps normalise the gradient).
"""
import os

import numpy as np
import pytest
import torch

from golden_util import golden_names, load_golden, oracle_config
from oracle import kge_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _close(a, b, rtol, atol, what):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    assert np.all(np.isfinite(a)), what + ": non-finite values"
    err = np.abs(a - b)
    lim = atol + rtol * np.abs(b)
    bad = err > lim
    assert not bad.any(), "%s: %d/%d out of tolerance, max err %.3e at %s (got %.6g want %.6g)" % (
        what, bad.sum(), bad.size, err.max(), np.unravel_index(np.argmax(err), err.shape),
        a.flat[np.argmax(err)], b.flat[np.argmax(err)])


class Args(dict):
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def make_args(case):
    a = Args()
    a.gpu = [0]
    a.mix_cpu_gpu = False
    a.has_edge_importance = bool(case.get("impts", False))
    a.strict_rel_part = False
    a.soft_rel_part = False
    a.lr = case["lr"]
    a.neg_deg_sample = bool(case.get("neg_deg", False))
    a.neg_deg_sample_eval = False
    a.eval_filter = False
    a.regularization_coef = case["reg_coef"]
    a.regularization_norm = case["reg_norm"]
    a.loss_genre = case.get("loss_genre", "Logsigmoid")
    a.neg_adversarial_sampling = case["adv"]
    a.adversarial_temperature = case["adv_temp"]
    a.pairwise = case.get("pairwise", False)
    a.margin = case.get("margin", 1.0)
    return a


def build_model(case, z):
    from dglke_amd.general_models import KEModel
    m = KEModel(make_args(case), case["model"], case["n_ent"], case["n_rel"], case["hidden"],
                case["gamma"], double_entity_emb=case["de"], double_relation_emb=case["dr"])
    m.entity_emb.emb.copy_(torch.from_numpy(z["init_entity"]))
    m.relation_emb.emb.copy_(torch.from_numpy(z["init_relation"]))
    m.entity_emb.state_sum.zero_()
    m.relation_emb.state_sum.zero_()
    if "init_projection" in z:           # TransR: third table (score_fun.py:114-118)
        m.score_func.projection_emb.emb.copy_(torch.from_numpy(z["init_projection"]))
        m.score_func.projection_emb.state_sum.zero_()
    return m


def golden_batch(z, case, s):
    from dglke_amd import plan
    p = "s%d_" % s
    w = z[p + "w"] if (p + "w") in z else None
    return plan.make_batch(z[p + "h"], z[p + "t"], z[p + "r"], z[p + "neg"], case["chunk"],
                           case["N"], bool(z[p + "neg_head"]), DEV, w)


_NOISE = None
ROW_ERRORS = {}          # name -> largest row error / lr seen in this session (written to gpurun_out/golden_row_errors.txt)


def rows_close(got, ref, name, tab, lr, what):
    """post-update table rows against a golden: |difference| <= max(1e-4 * lr, 3 x the distance of the float32 REFERENCE itself
    from the same reference code run in float64 on this golden (tests/golden/noise.json, `gen_golden.py --noise`)) - round 3's
    blanket 5e-3 * lr was argued, this bound is measured per case."""
    global _NOISE
    if _NOISE is None:
        import json
        _NOISE = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "noise.json")))
    noise = _NOISE[name][tab]
    tol = max(1e-4, 3.0 * noise) * lr
    got = np.asarray(got, dtype=np.float64)
    err = float(np.abs(got - np.asarray(ref, dtype=np.float64)).max())
    key = "%s %s" % (name, tab)
    ROW_ERRORS[key] = max(ROW_ERRORS.get(key, 0.0), err / lr)
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "golden_row_errors.txt"), "w") as f:
            f.write("golden table: largest |GPU - golden| / lr over every path tested this session   (reference fp32-vs-fp64 noise / lr)\n")
            for k in sorted(ROW_ERRORS):
                n_, t_ = k.split()
                f.write("%-44s %.3e   (%.3e)\n" % (k, ROW_ERRORS[k], _NOISE[n_][t_]))
    except OSError:
        pass
    assert err <= tol + 1e-6 * float(np.abs(ref).max()), \
        "%s %s: max|err| %.3e = %.3e lr > %.3e lr (reference's own fp32 noise: %.3e lr)" % (name, what, err, err / lr, tol / lr, noise)


def grad_tol(ref):
    return 3e-4 * max(float(np.abs(ref).max()), 1e-12)


@pytest.mark.parametrize("name", golden_names(transr=None, nd=None))
def test_dropin_model_matches_reference(name):
    """reference loop: model.forward -> loss.backward() -> model.update (train_pytorch.py:141-152)
    on the HIP-backed KEModel, every op one C-ABI call (TransR: its projections are library GEMMs, the two
    projection traces are checked too)."""
    from dglke_amd.dataloader import NegGraph, PosGraph
    z, case = load_golden(name)
    m = build_model(case, z)
    for s in range(1, case["steps"] + 1):
        p = "s%d_" % s
        b = golden_batch(z, case, s)
        pos_g, neg_g = PosGraph(b), NegGraph(b)
        loss, log = m.forward(pos_g, neg_g, 0)
        _close(pos_g.edata["score"].detach().cpu(), z[p + "pos_score"], 1e-4, 1e-4, name + " pos_score")
        ref_log = z[p + "log"]
        if not case.get("pairwise", False):
            _close(log["pos_loss"], ref_log[0], 1e-4, 1e-5, name + " pos_loss")
            _close(log["neg_loss"], ref_log[1], 1e-4, 1e-5, name + " neg_loss")
        _close(log["loss"], ref_log[2], 1e-4, 1e-5, name + " loss")
        _close(log.get("regularization", 0.0), ref_log[3], 1e-4, 1e-7, name + " reg")
        _close(loss.item(), z[p + "loss_total"], 1e-4, 1e-5, name + " total loss")
        loss.backward()
        et, rt = m.entity_emb.trace, m.relation_emb.trace
        assert len(et) == 2 and len(rt) == 1
        _close(et[0][1].grad.cpu(), z[p + "g_pos_ent"], 3e-4, grad_tol(z[p + "g_pos_ent"]), name + " g_pos_ent")
        _close(et[1][1].grad.cpu(), z[p + "g_neg"], 3e-4, grad_tol(z[p + "g_neg"]), name + " g_neg")
        _close(rt[0][1].grad.cpu(), z[p + "g_rel"], 3e-4, grad_tol(z[p + "g_rel"]), name + " g_rel")
        if case["model"] == "TransR":
            assert len(m.score_func.projection_emb.trace) == 2          # prepare + neg-prepare (score_fun.py:131-166)
        m.update(0)
        _close(m.entity_emb.state_sum.cpu(), z[p + "entity_state"], 2e-3, 1e-9, name + " ent state")
        _close(m.relation_emb.state_sum.cpu(), z[p + "relation_state"], 2e-3, 1e-9, name + " rel state")
        if (p + "projection_state") in z:
            pe = m.score_func.projection_emb
            _close(pe.state_sum.cpu(), z[p + "projection_state"], 2e-3, 1e-9, name + " projection state")
            _close(pe.emb.cpu(), z[p + "projection"], 1e-4, 5e-3 * case["lr"], name + " projection rows")
        if (p + "entity") in z:
            rows_close(m.entity_emb.emb.cpu(), z[p + "entity"], name, "entity", case["lr"], "entity rows (drop-in)")
            rows_close(m.relation_emb.emb.cpu(), z[p + "relation"], name, "relation", case["lr"], "relation rows (drop-in)")
    rows_close(m.entity_emb.emb.cpu(), z["final_entity"], name, "entity", case["lr"], "final entity (drop-in)")
    rows_close(m.relation_emb.emb.cpu(), z["final_relation"], name, "relation", case["lr"], "final relation (drop-in)")


@pytest.mark.parametrize("flags", [0, 1, 2, 8, 128, 512], ids=["auto", "force_pairwise", "no_transe_fast", "fused_loss", "split_fwd", "direct_tiles"])
@pytest.mark.parametrize("name", golden_names(transr=False))
def test_fused_step_matches_reference(name, flags):
    """kge_step_fused (one call per step) vs the reference's recorded scores / gradients / tables;
    both the matrix-core and the pairwise negative-score kernels."""
    from dglke_amd import _lib
    z, case = load_golden(name)
    m = build_model(case, z)
    eng = m.engine
    eng.hp.flags = flags       # _lib.FLAG_FORCE_PAIRWISE = 1, _lib.FLAG_NO_TRANSE_FAST = 2, _lib.FLAG_FUSED_LOSS = 8
    for s in range(1, case["steps"] + 1):
        p = "s%d_" % s
        b = golden_batch(z, case, s)
        want = eng.alloc_outputs(b)
        eng.step(b, want)
        torch.cuda.synchronize()
        _close(want["pos_score"].cpu(), z[p + "pos_score"], 1e-4, 1e-4, name + " pos_score")
        _close(want["neg_score"].cpu(), z[p + "neg_score"], 1e-4, 1e-4, name + " neg_score")
        l4 = eng.read_loss()
        ref_log = z[p + "log"]
        if not case.get("pairwise", False):
            _close(l4[0], ref_log[0], 1e-4, 1e-5, name + " pos_loss")
            _close(l4[1], ref_log[1], 1e-4, 1e-5, name + " neg_loss")
        _close(l4[2], ref_log[2], 1e-4, 1e-5, name + " loss")
        _close(l4[3], ref_log[3], 1e-4, 1e-7, name + " reg")
        ue = b.p["ue_id"]
        sel = np.searchsorted(ue, z[p + "nid"])
        _close(want["g_pos_ent"].cpu().numpy()[sel], z[p + "g_pos_ent"], 3e-4, grad_tol(z[p + "g_pos_ent"]), name + " g_pos_ent")
        _close(want["g_neg"].cpu(), z[p + "g_neg"], 3e-4, grad_tol(z[p + "g_neg"]), name + " g_neg")
        _close(want["g_rel"].cpu(), z[p + "g_rel"], 3e-4, grad_tol(z[p + "g_rel"]), name + " g_rel")
        _close(eng.ent_state.cpu(), z[p + "entity_state"], 2e-3, 1e-9, name + " ent state")
        _close(eng.rel_state.cpu(), z[p + "relation_state"], 2e-3, 1e-9, name + " rel state")
        if (p + "entity") in z:
            rows_close(eng.ent.cpu(), z[p + "entity"], name, "entity", case["lr"], "entity rows")
            rows_close(eng.rel.cpu(), z[p + "relation"], name, "relation", case["lr"], "relation rows")
    rows_close(eng.ent.cpu(), z["final_entity"], name, "entity", case["lr"], "final entity")
    rows_close(eng.rel.cpu(), z["final_relation"], name, "relation", case["lr"], "final relation")


# ---------------------------------------------------------------------------------------------
# oracle comparisons on seeded synthetic batches, up to the BASELINE config shapes
# ---------------------------------------------------------------------------------------------
SHAPES = [
    # model, n_ent, n_rel, hidden, de, dr, B, N, chunk, gamma, lr, adv, reg
    ("TransE_l2", 14951, 1345, 400, False, False, 1000, 200, 200, 19.9, 0.25, True, 1e-9),   # cfg-T
    ("DistMult", 14951, 1345, 400, False, False, 1000, 200, 200, 143.0, 0.08, True, 2e-6),   # cfg-D
    ("ComplEx", 50000, 535, 200, True, True, 1024, 256, 256, 143.0, 0.1, True, 2e-6),        # cfg-C shape
    ("RotatE", 20000, 300, 200, True, False, 512, 128, 128, 12.0, 0.01, True, 1e-7),
    ("TransE_l1", 14951, 1345, 400, False, False, 400, 200, 200, 16.0, 0.01, True, 1e-7),
    # TransE_l1 at the full cfg shape: sign(a - b) of a difference that fp32 and fp64 round to opposite sides of zero moves one
    # gradient element in ~750 000 by 2 w_ij (9.9e-7 against a 1e-9-scale tolerance).  Those elements are found from the fp64
    # operands (_l1_ambiguous) and the gradient / table rows they feed are excluded - a handful of 3 000 rows, asserted
    ("TransE_l1", 14951, 1345, 400, False, False, 1000, 200, 200, 16.0, 0.01, True, 1e-7),
    # the FB15k recipe's full shape of RotatE: the shared-pair backward runs its balanced split here (1024 workgroups)
    ("RotatE", 14951, 1345, 200, True, False, 1024, 256, 256, 12.0, 0.009, True, 1e-7),
    ("TransE_l2", 300, 10, 36, False, False, 120, 24, 40, 10.0, 0.1, False, 0.0),             # chunk != N, dups
    ("DistMult", 5000, 50, 64, False, False, 128, 288, 64, 143.0, 0.08, True, 1e-6),          # N > 256: stand-alone loss kernel
    ("TransE_l2", 5000, 50, 64, False, False, 96, 250, 48, 12.0, 0.1, True, 1e-6),            # ragged last column tile, fused loss
]


def _l1_ambiguous(bt, ent64, rel64, chunk, N, tau=4e-9):
    """TransE_l1's gradient is sign(a_id - b_jd): where |a_id - b_jd| < tau in fp64 the fp32 pos-side vector fl(x +- r) (half an
    ulp of its ~0.09 magnitude = 3.7e-9; the subtraction of two nearly equal floats is exact) may sit on the other side of b and
    the element moves by 2 w_ij.  Returns the gradient rows such elements feed - negative slots, positive
    entities (indices into bt['nid']), edges - and the table rows (entity ids, relation ids) whose Adagrad step they enter (the
    row mean of g^2 scales the WHOLE row's step, so the row is excluded, not the column)."""
    h, t, r, neg = ent64[bt["h"]], ent64[bt["t"]], rel64[bt["r"]], bt["neg"]
    a = (t - r) if bt["neg_head"] else (h + r)                      # pos-side vector of edge i
    B = a.shape[0]
    slots, edges = set(), set()
    for i in range(B):                                              # positive score |h + r - t|
        if (np.abs(h[i] + r[i] - t[i]) < tau).any():
            edges.add(i)
    both = set(edges)                                               # (a positive-score flip reaches head AND tail)
    for c in range(B // chunk):
        bn = ent64[neg[c * N:(c + 1) * N]]                          # [N, D]
        d = np.abs(a[c * chunk:(c + 1) * chunk, None, :] - bn[None, :, :]) < tau
        ii, jj = np.nonzero(d.any(axis=2))
        edges.update((c * chunk + ii).tolist())
        slots.update((c * N + jj).tolist())
    ent_ids, rel_ids, pos_local = set(), set(), set()
    for i in edges:
        side = bt["t"][i] if bt["neg_head"] else bt["h"][i]
        ents = [side] + ([bt["h"][i], bt["t"][i]] if i in both else [])
        ent_ids.update(int(x) for x in ents)
        rel_ids.add(int(bt["r"][i]))
    for j in slots:
        ent_ids.add(int(neg[j]))
    pos_local = set(np.searchsorted(bt["nid"], [e for e in ent_ids if e in set(bt["nid"].tolist())]).tolist())
    return dict(slots=sorted(slots), edges=sorted(edges), pos_local=sorted(pos_local), ent=sorted(ent_ids), rel=sorted(rel_ids))


def _masked(got, want, rows):
    """`got` with the listed rows replaced by the oracle's (rows excluded from a comparison)"""
    got = np.array(got, dtype=np.float64, copy=True)
    if len(rows):
        got[rows] = np.asarray(want, dtype=np.float64)[rows]
    return got


@pytest.mark.parametrize("flags", [0, 8, 128, 512 + 256], ids=["loss_kernel", "fused_loss", "split_fwd", "direct_tiles_dense_bwd"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%s-B%d-N%d-D%d" % (s[0], s[6], s[7], s[3]))
def test_fused_step_matches_oracle_at_config_shapes(shape, flags):
    from dglke_amd import plan
    from dglke_amd.engine import StepEngine
    model, n_ent, n_rel, hidden, de, dr, B, N, chunk, gamma, lr, adv, reg = shape
    cfg = O.Config(model, gamma, hidden, lr, adv=adv, adv_temp=1.0, reg_coef=reg, reg_norm=3,
                   double_ent=de, double_rel=dr)
    rng = np.random.RandomState(1234)
    ent = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(n_ent, cfg.ent_dim)).astype(np.float32)
    rel = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(n_rel, cfg.rel_dim)).astype(np.float32)
    eng = StepEngine(model, n_ent, n_rel, hidden, gamma, lr, DEV, de, dr, adv, 1.0, reg, 3, flags=flags)
    eng.load_tables(ent, rel)
    for step in (1, 2):
        # the oracle runs in fp64 from the SAME fp32 tables the GPU step starts from
        ent64 = eng.ent.cpu().numpy().astype(np.float64)
        rel64 = eng.rel.cpu().numpy().astype(np.float64)
        es64 = eng.ent_state.cpu().numpy().astype(np.float64)
        rs64 = eng.rel_state.cpu().numpy().astype(np.float64)
        bt = O.synth_batch(rng, n_ent, n_rel, B, N, chunk, step)
        b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], chunk, N, bt["neg_head"], DEV)
        want = eng.alloc_outputs(b)
        eng.step(b, want)
        torch.cuda.synchronize()
        amb = dict(slots=[], edges=[], pos_local=[], ent=[], rel=[])
        if model == "TransE_l1":           # rows fed by a sign(a - b) whose operands differ by less than fp32 resolves: excluded
            amb = _l1_ambiguous(bt, ent64, rel64, chunk, N)
            assert len(amb["ent"]) <= 30 and len(amb["rel"]) <= 15, "too many sign-ambiguous rows to call this a comparison: %r" % amb
        out = O.train_step(cfg, ent64, es64, rel64, rs64, bt["nid"], bt["h_local"], bt["t_local"],
                           bt["r"], bt["neg"], bt["neg_head"], chunk, N)
        tag = "%s step %d" % (model, step)
        _close(want["pos_score"].cpu(), out["pos_score"], 1e-4, 1e-4, tag + " pos_score")
        _close(want["neg_score"].cpu(), out["neg_score"], 1e-4, 1e-4, tag + " neg_score")
        l4 = eng.read_loss()
        _close(l4[:3], out["log"][:3], 1e-4, 1e-5, tag + " loss")
        _close(l4[3], out["log"][3], 1e-3, 1e-7, tag + " reg")
        sel = np.searchsorted(b.p["ue_id"], bt["nid"])
        _close(_masked(want["g_pos_ent"].cpu().numpy()[sel], out["g_pos_ent"], amb["pos_local"]), out["g_pos_ent"], 3e-4,
               grad_tol(out["g_pos_ent"]), tag + " g_pos_ent")
        _close(_masked(want["g_neg"].cpu(), out["g_neg"], amb["slots"]), out["g_neg"], 3e-4, grad_tol(out["g_neg"]), tag + " g_neg")
        _close(_masked(want["g_rel"].cpu(), out["g_rel"], amb["edges"]), out["g_rel"], 3e-4, grad_tol(out["g_rel"]), tag + " g_rel")
        _close(_masked(eng.ent_state.cpu(), es64, amb["ent"]), es64, 2e-3, 1e-9, tag + " ent state")
        _close(_masked(eng.rel_state.cpu(), rs64, amb["rel"]), rs64, 2e-3, 1e-9, tag + " rel state")
        _close(_masked(eng.ent.cpu(), ent64, amb["ent"]), ent64, 1e-4, 1e-3 * lr, tag + " entity rows")   # (fp64 oracle from the SAME
        _close(_masked(eng.rel.cpu(), rel64, amb["rel"]), rel64, 1e-4, 1e-3 * lr, tag + " relation rows")  #  fp32 tables: the per-step
        #                                                                                error, profiles/r03_row_error_trajectory.txt)


# RotatE shapes that put the shared-pair backward's balanced split (kge_neg_bcast.hip, NegArgs::lc_P) through its cases: columns
# (chunk, slab, row group) = C * ceil(K / 32) * chunk / 32, quad groups = N / 16; P = 1024 / columns parts for most columns,
# P + 1 for the rest, with and without remainders in either class, P = 1 included
BALANCED = [
    # model, n_ent, n_rel, hidden (= complex columns K), de, dr, B, N, chunk, gamma, lr, adv, reg
    ("RotatE", 30000, 300, 400, True, False, 512, 256, 256, 12.0, 0.01, True, 1e-7),     # 208 columns: 4 parts of 4 | 5 parts of 4,3,3,3,3
    ("RotatE", 30000, 300, 320, True, False, 1024, 128, 256, 12.0, 0.01, True, 1e-7),    # 320 columns, 8 groups: 3 parts of 3,3,2 | 4 of 2
    ("RotatE", 30000, 300, 96, True, False, 2048, 256, 256, 12.0, 0.01, True, 1e-7),     # 192 columns: 5 parts of 4,3,3,3,3 | 6 of 3,3,3,3,2,2
    ("RotatE", 30000, 300, 400, True, False, 2048, 256, 256, 12.0, 0.01, True, 1e-7),    # 832 columns: 1 part of 16 | 2 parts of 8
]


@pytest.mark.parametrize("shape", BALANCED, ids=lambda s: "%s-B%d-N%d-K%d" % (s[0], s[6], s[7], s[3]))
def test_balanced_split_of_the_shared_pair_backward_matches_oracle(shape):
    test_fused_step_matches_oracle_at_config_shapes(shape, 0)


def test_fused_step_is_deterministic_and_graph_replay_matches_eager():
    """size-independent property at the full cfg-T shape: the owner-computes update has no atomics,
    so two runs are bit-identical, and a HIP-graph replay equals eager launches."""
    from dglke_amd import plan
    from dglke_amd.engine import StepEngine
    rng = np.random.RandomState(7)
    n_ent, n_rel, B, N, D = 14951, 1345, 1000, 200, 400
    plans = []
    for step in range(1, 5):
        bt = O.synth_batch(rng, n_ent, n_rel, B, N, N, step)
        plans.append(plan.build_plan(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"]))
    results = []
    for mode in ("eager", "eager", "graph"):
        torch.manual_seed(0)
        eng = StepEngine("TransE_l2", n_ent, n_rel, D, 19.9, 0.25, DEV, False, False, True, 1.0, 1e-9, 3)
        batches = plan.upload(plans, DEV)
        if mode == "eager":
            for b in batches:
                eng.step(b)
        else:
            # warm-up outside capture so that workspace allocation is not captured
            eng.workspace_for(batches[0])
            g = eng.capture(batches)
            g.replay()
        torch.cuda.synchronize()
        results.append((eng.ent.cpu().numpy().copy(), eng.rel.cpu().numpy().copy(),
                        eng.ent_state.cpu().numpy().copy(), np.array(eng.read_loss_sums())))
    for k in range(4):
        assert np.array_equal(results[0][k], results[1][k]), "run-to-run difference in output %d" % k
        assert np.array_equal(results[0][k], results[2][k]), "graph replay differs in output %d" % k
    assert np.isfinite(results[0][3]).all()


@pytest.mark.parametrize("model,gamma,dbl,D", [("TransE_l2", 19.9, False, 400), ("DistMult", 143.0, False, 400),
                                               ("ComplEx", 143.0, True, 200)])
def test_merged_forward_launch_equals_split_launches_bit_for_bit(model, gamma, dbl, D):
    """round 3: the strict step's first launch runs the forward GEMM tiles (pos-side fragments built from the table rows, raw
    products; the loss kernel applies the TransE_l2 distance) next to the edge-forward rows.  Same arithmetic as the two
    separate launches (KGE_FLAG_SPLIT_FWD): tables, states and loss sums must be bit-identical - cfg-T / cfg-D shape,
    both corruption modes (4 steps)."""
    from dglke_amd import plan, _lib
    from dglke_amd.engine import StepEngine
    rng = np.random.RandomState(11)
    n_ent, n_rel, B, N = 14951, 1345, 1000, 200
    plans = []
    for step in range(1, 5):
        bt = O.synth_batch(rng, n_ent, n_rel, B, N, N, step)
        plans.append(plan.build_plan(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"]))
    results = []
    # ... and the other launch variants of round 3 (direct-load forward tiles, dense copy of the negative rows for the backward):
    # the same products in the same order
    for flags in (0, _lib.FLAG_SPLIT_FWD, _lib.FLAG_FWD_DIRECT, _lib.FLAG_DENSE_BWD, _lib.FLAG_FWD_DIRECT | _lib.FLAG_DENSE_BWD):
        torch.manual_seed(0)
        eng = StepEngine(model, n_ent, n_rel, D, gamma, 0.25, DEV, dbl, dbl, True, 1.0, 1e-9, 3, flags=flags)
        batches = plan.upload(plans, DEV)
        scores = []
        for b in batches:
            want = dict(neg_score=torch.empty(b.C, b.chunk, b.N, device=DEV))
            eng.step(b, want)
            scores.append(want["neg_score"].cpu().numpy().copy())
        torch.cuda.synchronize()
        results.append((eng.ent.cpu().numpy().copy(), eng.rel.cpu().numpy().copy(), eng.ent_state.cpu().numpy().copy(),
                        eng.rel_state.cpu().numpy().copy(), np.array(eng.read_loss_sums()), np.stack(scores)))
    for v in range(1, len(results)):
        for k in range(6):
            assert np.array_equal(results[0][k], results[v][k]), "launch variant %d: output %d differs from the default" % (v, k)
    assert np.isfinite(results[0][4]).all()


def _run_steps(model, gamma, dbl, D, flags, plans, n_ent, n_rel, lr=0.25, graph=False, replays=1, scores=True, background=None):
    from dglke_amd import plan
    from dglke_amd.engine import StepEngine
    torch.manual_seed(0)
    eng = StepEngine(model, n_ent, n_rel, D, gamma, lr, DEV, dbl, dbl, True, 1.0, 1e-9, 3, flags=flags)
    batches = plan.upload(plans, DEV)
    sc = []
    if graph:
        eng.workspace_for(batches[0])
        g = eng.capture(batches)
        if background is not None:
            background(True)
        for _ in range(replays):
            g.replay()
        if background is not None:
            background(False)
    else:
        for b in batches:
            want = dict(neg_score=torch.empty(b.C, b.chunk, b.N, device=DEV)) if scores else None
            eng.step(b, want)
            if scores:
                sc.append(want["neg_score"].cpu().numpy().copy())
    torch.cuda.synchronize()
    assert int(eng.tickets.abs().max().item()) == 0, "hand-off tickets not returned to zero"
    return (eng.ent.cpu().numpy().copy(), eng.rel.cpu().numpy().copy(), eng.ent_state.cpu().numpy().copy(),
            eng.rel_state.cpu().numpy().copy(), np.array(eng.read_loss_sums()), np.stack(sc) if sc else np.zeros(1))


@pytest.mark.parametrize("model,gamma,dbl,D,B,N", [("TransE_l2", 19.9, False, 400, 1000, 200), ("DistMult", 143.0, False, 400, 1000, 200),
                                                   ("ComplEx", 143.0, True, 200, 1024, 256), ("TransE_l2", 19.9, False, 400, 1000, 40),
                                                   ("DistMult", 143.0, False, 100, 192, 96)])
def test_loss_rows_inside_the_first_launch_equal_the_loss_launch(model, gamma, dbl, D, B, N):
    """round 4, KGE_FLAG_LOSS_IN_FWD: the strict step as THREE launches - the forward tiles store final scores, every workgroup of
    a 16-row strip draws a ticket and the last one runs LossGenerator on the strip's rows (kge_neg_gemm.hip,
    neg_fwd_loss_edge_kernel) - against the default 4-launch sequence (same tiles, raw products, stand-alone loss kernel).
    DistMult / ComplEx: the same instructions on the same values - scores, tables, states and loss sums bit for bit.  TransE_l2:
    |a|^2 and |b|^2 of the distance are summed by the tiles in another order than by edge_fwd - equal within that rounding.  With
    and without per-step outputs (the LEAN and the generic instance), both corruption modes, the dense-backward variant.
    (Opt-in: correct, and slower than the loss launch - profiles/r04_loss_fold.txt.)"""
    from dglke_amd import plan, _lib
    rng = np.random.RandomState(19)
    n_ent, n_rel, lr = 14951, 1345, 0.25
    plans = []
    for step in range(1, 5):
        bt = O.synth_batch(rng, n_ent, n_rel, B, N, N, step)
        plans.append(plan.build_plan(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"]))
    LF = _lib.FLAG_LOSS_IN_FWD
    for with_scores in (True, False):
        ref = _run_steps(model, gamma, dbl, D, 0, plans, n_ent, n_rel, lr, scores=with_scores)
        for flags in (LF, LF | _lib.FLAG_DENSE_BWD):
            got = _run_steps(model, gamma, dbl, D, flags, plans, n_ent, n_rel, lr, scores=with_scores)
            for k in range(6):
                if model == "TransE_l2":
                    tol = 1e-4 * lr if k < 2 else 2e-5 * np.abs(ref[k]).max()       # (rows: the per-step error bound of the step itself)
                    err = np.abs(got[k] - ref[k]).max()
                    assert err <= tol, \
                        "TransE_l2 flags %d output %d: in-launch loss rows differ from the loss launch by %.3e" % (flags, k, err)
                else:
                    assert np.array_equal(got[k], ref[k]), "%s flags %d output %d differs from the loss launch" % (model, flags, k)
        assert np.isfinite(ref[4]).all() and np.abs(ref[0]).max() > 0


@pytest.mark.parametrize("model,gamma,dbl,D", [("TransE_l2", 19.9, False, 400), ("DistMult", 143.0, False, 400)])
def test_in_launch_hand_off_under_load_and_replay(model, gamma, dbl, D):
    """the hand-off of the 3-launch step (KGE_FLAG_LOSS_IN_FWD; write-through score stores -> ticket -> L1-bypassing loads by the last arriver) must
    hold on a busy chip and with warm caches: 40 graph replays of 6 steps (every replay re-uses the same score lines) while a
    second stream streams 256 MB copies, twice - bit-identical tables, states and loss sums run to run, and (DistMult: bit for
    bit; TransE_l2: within rounding) equal to the 4-launch sequence run without the load."""
    from dglke_amd import plan, _lib
    rng = np.random.RandomState(23)
    n_ent, n_rel, B, N, lr = 14951, 1345, 1000, 200, 0.25
    plans = []
    for step in range(1, 7):
        bt = O.synth_batch(rng, n_ent, n_rel, B, N, N, step)
        plans.append(plan.build_plan(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"]))
    side = torch.cuda.Stream()
    big_a = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
    big_b = torch.empty(64 << 20, dtype=torch.float32, device=DEV)

    def background(start):
        if start:
            with torch.cuda.stream(side):
                for _ in range(12):
                    big_b.copy_(big_a)
                    big_a.copy_(big_b)
        else:
            side.synchronize()
    ref = _run_steps(model, gamma, dbl, D, 0, plans, n_ent, n_rel, lr, graph=True, replays=40)
    runs = [_run_steps(model, gamma, dbl, D, _lib.FLAG_LOSS_IN_FWD, plans, n_ent, n_rel, lr, graph=True, replays=40, background=background)
            for _ in range(2)]
    for k in range(5):
        assert np.array_equal(runs[0][k], runs[1][k]), "3-launch step under load: run-to-run difference in output %d" % k
        if model == "DistMult":
            assert np.array_equal(runs[0][k], ref[k]), "3-launch step under load differs from the 4-launch step in output %d" % k
        else:
            tol = 1e-3 * lr if k < 2 else 1e-3 * np.abs(ref[k]).max()     # (240 steps: rounding differences compound)
            assert np.abs(runs[0][k] - ref[k]).max() <= tol, "output %d: %.3e" % (k, np.abs(runs[0][k] - ref[k]).max())
    assert np.isfinite(ref[4]).all()


@pytest.mark.parametrize("model,de_,B,N,hidden,gamma,lr", [("RotatE", True, 1024, 256, 200, 12.0, 0.009),
                                                           ("TransE_l1", False, 1000, 200, 400, 16.0, 0.01)])
def test_pairwise_fused_launches_equal_separate_launches_bit_for_bit(model, de_, B, N, hidden, gamma, lr):
    """the pairwise family's fused launches of round 3 against round 2's launch sequence (KGE_FLAG_SPLIT_FWD), bit for bit at the
    FB15k recipes' shapes.  RotatE: the sum of the shared-pair backward's GN partials runs as the second half of the edge_bwd
    launch.  TransE_l1: the forward tasks build their uniform rows x +/- r from the table rows and share the launch with the
    edge-forward rows (5 launches instead of 6)."""
    from dglke_amd import plan, _lib
    from dglke_amd.engine import StepEngine
    rng = np.random.RandomState(13)
    n_ent, n_rel = 14951, 1345
    plans = []
    for step in range(1, 4):
        bt = O.synth_batch(rng, n_ent, n_rel, B, N, N, step)
        plans.append(plan.build_plan(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"]))
    results = []
    for flags in (0, _lib.FLAG_SPLIT_FWD):
        torch.manual_seed(0)
        eng = StepEngine(model, n_ent, n_rel, hidden, gamma, lr, DEV, de_, False, True, 1.0, 1e-7, 3, flags=flags)
        for b in plan.upload(plans, DEV):
            eng.step(b)
        torch.cuda.synchronize()
        results.append((eng.ent.cpu().numpy().copy(), eng.rel.cpu().numpy().copy(), eng.ent_state.cpu().numpy().copy(),
                        eng.rel_state.cpu().numpy().copy(), np.array(eng.read_loss_sums())))
    for k in range(5):
        assert np.array_equal(results[0][k], results[1][k]), "%s fused vs separate launches: output %d differs" % (model, k)
    assert np.abs(results[0][0]).max() > 0 and np.isfinite(results[0][4]).all()


@pytest.mark.parametrize("model,de_,dr_,hidden,flags", [("TransE_l2", False, False, 64, 0), ("RotatE", True, False, 32, 0),
                                                        ("DistMult", False, False, 64, 0), ("TransR", False, False, 16, 0),
                                                        ("RESCAL", False, False, 16, 0), ("TransE_l2", False, False, 64, 32),
                                                        ("RotatE", True, False, 32, 32), ("TransE_l1", False, False, 64, 0),
                                                        ("TransE_l1", False, False, 64, -1)],
                         ids=["TransE_l2", "RotatE", "DistMult", "TransR", "RESCAL", "TransE_l2-neg_deg", "RotatE-neg_deg",
                              "TransE_l1", "TransE_l1-5-partials-3-parts"])
def test_four_phase_calls_equal_the_fused_step_bit_for_bit(model, de_, dr_, hidden, flags):
    """kge_step_phase (the reference's sample / forward / backward / update timers, train_pytorch.py:127-177; runs once per
    log interval in every dglke_train run): the four phase groups issued one after the other ARE kge_step_fused - same
    tables, states and loss sums, bit for bit, although the fused call merges launches that the phase calls keep apart."""
    import ctypes as C
    from dglke_amd import _lib, plan
    from dglke_amd.engine import StepEngine
    n_ent, n_rel, B, N = 800, 11, 96, 32
    if flags == -1:
        # TransE_l1 at the recipe's chunk shape: the fused call's update kernel sums the shared-pair backward's 5 GN partials and
        # 3 GA parts itself (UpdateArgs::gn_parts), the phase calls run the stand-alone reduction launch - same additions, same order
        n_ent, B, N, flags = 3000, 400, 200, 0
    rng = np.random.RandomState(17)
    plans = []
    for step in range(1, 4):
        bt = O.synth_batch(rng, n_ent, n_rel, B, N, N, step)
        plans.append(plan.build_plan(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"]))
    results = []
    for phased in (False, True):
        torch.manual_seed(0)
        eng = StepEngine(model, n_ent, n_rel, hidden, 10.0, 0.1, DEV, de_, dr_, True, 1.0, 1e-5, 3, flags=flags)
        for b in plan.upload(plans, DEV):
            if not phased:
                eng.step(b)
                continue
            ws = eng.workspace_for(b)
            out = _lib.KgeStepOut()
            out.loss_accum = _lib.ptr(eng.loss_accum)
            for ph in (_lib.PHASE_GATHER, _lib.PHASE_FORWARD, _lib.PHASE_BACKWARD, _lib.PHASE_UPDATE):
                _lib.check(_lib.lib().kge_step_phase(C.byref(eng.hp), C.byref(eng.tb), C.byref(b.c), C.byref(out), _lib.ptr(ws),
                                                     eng._ws_bytes, ph, _lib.stream_ptr()))
        torch.cuda.synchronize()
        res = [eng.ent.cpu().numpy().copy(), eng.rel.cpu().numpy().copy(), eng.ent_state.cpu().numpy().copy(),
               eng.rel_state.cpu().numpy().copy(), np.array(eng.read_loss_sums())]
        if eng.proj is not None:
            res.append(eng.proj.cpu().numpy().copy())
        results.append(res)
    for k in range(len(results[0])):
        assert np.array_equal(results[0][k], results[1][k]), "%s: output %d of the phase calls differs from the fused step" % (model, k)
    assert np.isfinite(results[0][4]).all() and np.abs(results[0][4]).sum() > 0


def test_row_error_trajectory_over_24_steps():
    """How far do the post-update rows drift from exact arithmetic, step after step, at the cfg-T shape?  Every step restarts the
    fp64 statement from the GPU's own fp32 tables, so the numbers are PER-STEP errors (they cannot accumulate):
      update:  fp64 Adagrad (trace order, oracle.adagrad_update) fed the GPU's OWN fp32 gradients - isolates the update kernel;
      full:    the whole fp64 oracle step - adds the gradient kernels' rounding, amplified by Adagrad's normalisation
               (delta = -lr * g / sqrt(state): with state ~ 0 in the first steps a relative error of the gradient IS a relative
               error of a step of size ~lr).
    The trajectory is written next to the test (gpurun_out/) and bounded: at this shape the rows of EVERY step agree with exact
    arithmetic to < 1e-4 * lr (update kernel alone: < 2e-6 * lr).  The 5e-3 * lr of the golden comparisons is a bound on the
    fp32 REFERENCE's own rounding at toy sizes (D = 16, 9 - 60 entities, duplicate-heavy batches), not on these kernels."""
    from dglke_amd import plan
    from dglke_amd.engine import StepEngine
    rng = np.random.RandomState(21)
    n_ent, n_rel, B, N, D, lr = 14951, 1345, 1000, 200, 400, 0.25
    cfg = O.Config("TransE_l2", 19.9, D, lr, adv=True, adv_temp=1.0, reg_coef=1e-9, reg_norm=3)
    torch.manual_seed(0)
    eng = StepEngine("TransE_l2", n_ent, n_rel, D, 19.9, lr, DEV, False, False, True, 1.0, 1e-9, 3)
    lines = ["step  update-only max|err|/lr   full-step max|err|/lr (entity rows)   relation rows full   rows touched"]
    upd_err, full_err = [], []
    for step in range(1, 25):
        ent64, rel64 = eng.ent.cpu().numpy().astype(np.float64), eng.rel.cpu().numpy().astype(np.float64)
        es64, rs64 = eng.ent_state.cpu().numpy().astype(np.float64), eng.rel_state.cpu().numpy().astype(np.float64)
        bt = O.synth_batch(rng, n_ent, n_rel, B, N, N, step)
        b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"], DEV)
        want = eng.alloc_outputs(b)
        eng.step(b, want)
        torch.cuda.synchronize()
        got_e, got_r = eng.ent.cpu().numpy().astype(np.float64), eng.rel.cpu().numpy().astype(np.float64)
        # (i) fp64 update from the GPU's own gradients
        ue, ur = ent64.copy(), rel64.copy()
        ues, urs = es64.copy(), rs64.copy()
        sel = np.searchsorted(b.p["ue_id"], bt["nid"])
        O.adagrad_update(ue, ues, bt["nid"], want["g_pos_ent"].cpu().numpy().astype(np.float64)[sel], lr)
        O.adagrad_update(ue, ues, bt["neg"], want["g_neg"].cpu().numpy().astype(np.float64), lr)
        O.adagrad_update(ur, urs, bt["r"], want["g_rel"].cpu().numpy().astype(np.float64), lr)
        # (ii) the whole step in fp64
        fe, fr, fes, frs = ent64.copy(), rel64.copy(), es64.copy(), rs64.copy()
        O.train_step(cfg, fe, fes, fr, frs, bt["nid"], bt["h_local"], bt["t_local"], bt["r"], bt["neg"], bt["neg_head"], N, N)
        touched = np.unique(np.concatenate([bt["nid"], bt["neg"]]))
        e_u = np.abs(got_e - ue).max() / lr
        e_f = np.abs(got_e - fe).max() / lr
        r_f = np.abs(got_r - fr).max() / lr
        upd_err.append(max(e_u, np.abs(got_r - ur).max() / lr))
        full_err.append(max(e_f, r_f))
        lines.append("%4d  %12.3e              %12.3e                         %12.3e        %d" % (step, e_u, e_f, r_f, len(touched)))
    try:
        import os
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        open(os.path.join(out, "row_error_trajectory.txt"), "w").write("\n".join(lines) + "\n")
    except OSError:
        pass
    print("\n".join(lines))
    # measured on MI355X (profiles/r03_row_error_trajectory.txt): update-only 4.0e-7 .. 7.1e-7, full step 6.6e-7 .. 3.7e-5 (x lr)
    assert max(upd_err) < 2e-6, "update kernel vs fp64 Adagrad on its own gradients: %.3e * lr" % max(upd_err)
    assert max(full_err) < 1e-4, "per-step row error: %r" % (full_err,)


def test_adagrad_scatter_duplicate_semantics():
    """ExternalEmbedding.update duplicate-index semantics (tensor_models.py:352-361) with the
    lock-free scatter kernels: a table of 5 rows, 4096 updates."""
    from dglke_amd import ops
    rng = np.random.RandomState(3)
    table = rng.randn(5, 64).astype(np.float32)
    state = rng.rand(5).astype(np.float32)
    idx = rng.randint(0, 5, size=4096).astype(np.int64)
    grad = (rng.randn(4096, 64) * 0.01).astype(np.float32)
    t_d, s_d = torch.from_numpy(table).to(DEV), torch.from_numpy(state).to(DEV)
    ops.adagrad_scatter(t_d, s_d, torch.from_numpy(idx).to(DEV), torch.from_numpy(grad).to(DEV), 0.3)
    t64, s64 = table.astype(np.float64), state.astype(np.float64)
    O.adagrad_update(t64, s64, idx, grad.astype(np.float64), 0.3)
    _close(s_d.cpu(), s64, 1e-5, 1e-7, "state")
    _close(t_d.cpu(), t64, 1e-5, 1e-5, "table")


def test_gather_rows_and_errors():
    from dglke_amd import ops, _lib
    t = torch.arange(0, 40, dtype=torch.float32, device=DEV).reshape(10, 4)
    idx = torch.tensor([9, 0, 0, 3], device=DEV)
    assert torch.equal(ops.gather_rows(t, idx), t[idx])
    assert ops.gather_rows(t, idx[:0]).shape == (0, 4)
    t5 = torch.arange(0, 50, dtype=torch.float32, device=DEV).reshape(10, 5)   # unaligned rows
    assert torch.equal(ops.gather_rows(t5, idx), t5[idx])
    with pytest.raises(_lib.KgeError):
        ops.gather_rows(t.cpu(), idx.cpu())      # no CPU fallback
    with pytest.raises(_lib.KgeError):
        ops.score_pos("ComplEx", torch.zeros(2, 6, device=DEV), torch.zeros(2, 4, device=DEV),
                      torch.zeros(2, 6, device=DEV), 1.0)


def test_sharded_engine_world1_equals_fused_step():
    """dglke_amd.dist.DistEngine (kge_route_build -> pull -> kge_step_grads -> push -> kge_adagrad_apply_merged) with a
    single rank must reproduce the fused single-GPU step: same kernels, same trace order."""
    import os
    import torch.distributed as dist
    from dglke_amd import dist as kd, plan
    from dglke_amd.engine import StepEngine
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        n_ent, n_rel, hidden, B, N = 5000, 40, 64, 128, 32
        rng = np.random.RandomState(11)
        for model, de_, dr_, direct in (("TransE_l2", False, False, False), ("RotatE", True, False, False),
                                        ("RotatE", True, False, True), ("TransE_l2", False, False, True)):
            a = StepEngine(model, n_ent, n_rel, hidden, 12.0, 0.1, DEV, de_, dr_, True, 1.0, 1e-6, 3)
            b = StepEngine(model, 1, n_rel, hidden, 12.0, 0.1, DEV, de_, dr_, True, 1.0, 1e-6, 3)
            b.rel.copy_(a.rel)
            ent = a.ent.clone()
            state = torch.zeros(n_ent, device=DEV)
            # RotatE: through the RCCL calls too (equal-split all_to_all_single / all_gather_into_tensor with one rank) and the
            # one-step pull pipeline's streams and events - with a single batch in flight the pipeline is the synchronous step
            # direct: the collectives through dist.RcclComm (librccl by ctypes on the step's streams, its own one-rank communicator,
            # the push exchanges as one grouped launch) instead of the c10d wrappers
            coll = model == "RotatE" or direct
            comm = kd.RcclComm() if direct else None
            assert comm is None or (comm.world, comm.rank) == (1, 0)
            deng = kd.DistEngine(b, kd.ShardSpec(n_ent, 1, 0), ent, state, always_collective=coll, comm=comm)
            for step in range(1, 4):
                bt = O.synth_batch(rng, n_ent, n_rel, B, N, N, step)
                a.step(plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"], DEV))
                gb = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"], DEV)      # GLOBAL ids
                gb.UE = 2 * B + (B // N) * N         # the engine sizes its buffers once, for the bound
                if coll:
                    deng.step_pipelined(gb, None)
                else:
                    deng.step(gb)                    # route on the device -> (world 1: no collective) -> grads -> merged apply
            assert deng.check_overflow() == 0
            torch.cuda.synchronize()
            # the two paths run different instantiations of the update code (in-place vs gradient-emitting +
            # owner-side apply): same formulas, fp32 contraction may differ, and Adagrad's first steps divide
            # by |g| - a few ulp of the gradient become ~1e-5 * lr in the row.  Still 100x tighter than the
            # 5e-3 * lr bar against the reference.
            _close(ent.cpu(), a.ent.cpu(), 1e-5, 5e-6, model + " sharded entity table")
            _close(state.cpu(), a.ent_state.cpu(), 1e-5, 1e-8, model + " sharded entity state")
            _close(b.rel.cpu(), a.rel.cpu(), 1e-5, 5e-6, model + " relation table")
            _close(b.rel_state.cpu(), a.rel_state.cpu(), 1e-5, 1e-8, model + " relation state")
            la, lb_ = a.read_loss_sums(), b.read_loss_sums()
            _close(lb_, la, 1e-5, 1e-6, model + " loss sums")
            if comm is not None:
                comm.close()
        # the pull pipeline (pull of step s+1 on the side stream while step s computes): through the c10d wrappers (side-stream
        # section under `with torch.cuda.stream`) and through RcclComm (stream handed to the calls, events reused) - same
        # one-step-stale schedule, same bits
        res = []
        for direct in (False, True):
            rng = np.random.RandomState(23)
            b = StepEngine("DistMult", 1, n_rel, hidden, 12.0, 0.1, DEV, False, False, True, 1.0, 1e-6, 3)
            torch.manual_seed(5)
            b.rel.uniform_(-0.2, 0.2)
            ent = torch.empty(n_ent, hidden, device=DEV).uniform_(-0.2, 0.2)
            state = torch.zeros(n_ent, device=DEV)
            comm = kd.RcclComm() if direct else None
            deng = kd.DistEngine(b, kd.ShardSpec(n_ent, 1, 0), ent, state, always_collective=True, comm=comm)
            gbs = []
            for step in range(1, 6):
                bt = O.synth_batch(rng, n_ent, n_rel, B, N, N, step)
                gb = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"], DEV)
                gb.UE = 2 * B + (B // N) * N
                gbs.append(gb)
            for k, gb in enumerate(gbs):
                deng.step_pipelined(gb, gbs[k + 1] if k + 1 < len(gbs) else None)
            torch.cuda.synchronize()
            assert deng.check_overflow() == 0
            res.append((ent.cpu(), state.cpu(), b.rel.cpu().clone(), b.rel_state.cpu().clone()))
            if comm is not None:
                comm.close()
        for x, y in zip(res[0], res[1]):
            assert torch.equal(x, y)
        assert not torch.equal(res[0][0], torch.empty(n_ent, hidden).fill_(0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model,hidden", [("TransR", 32), ("RESCAL", 32)])
def test_dist_engine_transr_rescal_relation_side_in_place(model, hidden):
    """round 6 (VERDICT r05 missing 1, second half): TransR and RESCAL through the all-to-all engine - the gradient-emitting step with the
    relation side applied IN PLACE (relation partitioning: emit.gr NULL; relation rows / matrices and TransR's projection rows belong
    to this rank), entity gradients as packed messages + owner-side apply.  World 1 with its RCCL exchanges kept, synchronous and
    overlapped; against the single-table step on the same batches (same kernels; the entity update runs the emitting instance +
    apply instead of the in-place one: the tolerance of test_sharded_engine_world1_equals_fused_step)."""
    from dglke_amd import dist as kd, plan
    from dglke_amd.engine import StepEngine
    n_ent, n_rel, B, N = 3000, 11, 128, 32
    rng0 = np.random.RandomState(17)
    batches = [O.synth_batch(rng0, n_ent, n_rel, B, N, N, s) for s in range(1, 5)]
    a = StepEngine(model, n_ent, n_rel, hidden, 8.0, 0.05, DEV, False, False, True, 1.0, 1e-6, 3)
    ent0, rel0 = a.ent.clone(), a.rel.clone()
    proj0 = a.proj.clone() if a.proj is not None else None
    for bt in batches:
        a.step(plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"], DEV))
    torch.cuda.synchronize()
    for sched in (False, "overlap"):
        b = StepEngine(model, 1, n_rel, hidden, 8.0, 0.05, DEV, False, False, True, 1.0, 1e-6, 3)
        b.rel.copy_(rel0); b.rel_state.zero_()
        if proj0 is not None:
            b.proj.copy_(proj0); b.proj_state.zero_()
        ent, state = ent0.clone(), torch.zeros(n_ent, device=DEV)
        comm = kd.RcclComm()
        de = kd.DistEngine(b, kd.ShardSpec(n_ent, 1, 0), ent, state, always_collective=True, comm=comm, rel_local=True)
        assert de._rel_inplace and de.packed
        gbs = []
        for bt in batches:
            gb = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"], DEV)
            gb.UE = 2 * B + (B // N) * N
            gbs.append(gb)
        de.ensure_capacity(gbs)
        if sched == "overlap":
            # (one-step-stale entity rows inside a group: groups of ONE step = the synchronous dataflow through the overlapped code path)
            for gb in gbs:
                de._steps([gb], "overlap")
        else:
            for gb in gbs:
                de.step(gb)
        torch.cuda.synchronize()
        assert de.check_overflow() == 0
        tag = "%s a2a %s" % (model, sched or "synchronous")
        _close(ent.cpu(), a.ent.cpu(), 1e-5, 5e-6, tag + " entity table")
        _close(state.cpu(), a.ent_state.cpu(), 1e-5, 1e-8, tag + " entity state")
        assert torch.equal(b.rel, a.rel) and torch.equal(b.rel_state, a.rel_state), tag + ": relation side (applied in place, same kernels)"
        if proj0 is not None:
            assert torch.equal(b.proj, a.proj) and torch.equal(b.proj_state, a.proj_state), tag + ": projection table"
        de.close()
    assert float((a.ent_state > 0).sum()) > 100


def _pair_neg_reference(model, neg_head, a, b, W, C, chunk, N, gamma):
    """fp64 torch autograd of the reference's pairwise create_neg forms (score_fun.py:36-38 cdist p=1;
    :526-531, 548-552 RotatE modulus of the broadcast difference) given the pos-side vectors a."""
    a = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    b = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    A = a.view(C, chunk, 1, -1)
    Bn = b.view(C, 1, N, -1)
    if model == "TransE_l1":
        s = gamma - (A - Bn).abs().sum(-1)
    else:
        K = a.shape[1] // 2
        dr = A[..., :K] - Bn[..., :K]
        di = A[..., K:] - Bn[..., K:]
        s = gamma - torch.sqrt(dr * dr + di * di).sum(-1)
    (s * torch.tensor(W, dtype=torch.float64)).sum().backward()
    return s.detach().numpy(), a.grad.numpy(), b.grad.numpy()


@pytest.mark.parametrize("model", ["TransE_l1", "RotatE"])
@pytest.mark.parametrize("shape", [(5, 200, 200, 400), (3, 37, 44, 48), (2, 16, 4, 16), (1, 129, 260, 112)])
def test_shared_pair_backward_matches_two_pass_and_fp64(model, shape):
    """TransE_l1 / RotatE negative-score backward: the kernel that evaluates every (positive, negative) pair once
    for both products (lane = column, LDS-staged) against the two-pass kernels (KGE_FLAG_TWO_PASS_PAIR) and an
    fp64 autograd reference - ragged row blocks, partial column slabs and a partial last group of negatives."""
    from dglke_amd import _lib, ops
    C, chunk, N, D = shape
    rng = np.random.RandomState(7)
    B = C * chunk
    de = D
    dr = D // 2 if model == "RotatE" else D
    x = rng.uniform(-1, 1, (B, de)).astype(np.float32)
    r = rng.uniform(-1, 1, (B, dr)).astype(np.float32)
    nb = rng.uniform(-1, 1, (C * N, de)).astype(np.float32)
    W = rng.uniform(-1, 1, (C, chunk, N)).astype(np.float32)
    outs = {}
    for name, fl in (("shared", 0), ("two_pass", _lib.FLAG_TWO_PASS_PAIR)):
        xt = torch.tensor(x, device=DEV, requires_grad=True)
        rt = torch.tensor(r, device=DEV, requires_grad=True)
        nt = torch.tensor(nb, device=DEV, requires_grad=True)
        s = ops.score_neg(model, False, xt, rt, nt, C, chunk, N, 12.0, emb_init=0.07, flags=fl)
        (s * torch.tensor(W, device=DEV)).sum().backward()
        outs[name] = (s.detach().cpu().numpy(), xt.grad.cpu().numpy(), rt.grad.cpu().numpy(), nt.grad.cpu().numpy())
    gscale = max(np.abs(outs["two_pass"][3]).max(), np.abs(outs["two_pass"][1]).max())
    for k, what in ((1, "g_pos_side"), (2, "g_rel"), (3, "g_neg")):
        _close(outs["shared"][k], outs["two_pass"][k], 0, 2e-5 * np.abs(outs["two_pass"][k]).max(),
               "%s shared vs two-pass %s" % (model, what))
    if model == "TransE_l1":       # a = h + r: compare the negative-row gradient with fp64 directly
        a = x.astype(np.float64) + r.astype(np.float64)
        s64, ga64, gn64 = _pair_neg_reference(model, False, a, nb, W, C, chunk, N, 12.0)
        _close(outs["shared"][0], s64, 1e-4, 1e-4, "TransE_l1 scores vs fp64")
        # sign() is discontinuous: an fp32 a_k - b_k of the other sign than fp64 flips one W_ij term
        bad = np.abs(outs["shared"][3] - gn64) > 3e-4 * gscale
        assert bad.mean() < 2e-3, "TransE_l1 g_neg vs fp64: %.4f of the entries differ" % bad.mean()
        bad = np.abs(outs["shared"][1] - ga64) > 3e-4 * gscale
        assert bad.mean() < 2e-3, "TransE_l1 g_pos_side vs fp64: %.4f of the entries differ" % bad.mean()


@pytest.mark.parametrize("flags", [32, 33], ids=["auto", "force_pairwise"])
@pytest.mark.parametrize("name", golden_names(nd=True, transr=False))      # (nd_transr_*: three tables, tests/test_gpu_transr.py)
def test_fused_step_neg_deg_sample_matches_reference(name, flags):
    """--neg_deg_sample on the fused step (KGE_FLAG_NEG_DEG_SAMPLE = 32): the chunk's own positives join the
    negatives, masked diagonal, their gradients in the positive trace - against scores / gradients / tables
    recorded from the reference running with args.neg_deg_sample = True (general_models.py:396-402, 424-432)."""
    z, case = load_golden(name)
    m = build_model(case, z)
    eng = m.engine
    eng.hp.flags = flags
    chunk, N = case["chunk"], case["N"]
    Np = chunk + N
    for s in range(1, case["steps"] + 1):
        p = "s%d_" % s
        b = golden_batch(z, case, s)
        want = eng.alloc_outputs(b)
        eng.step(b, want)
        torch.cuda.synchronize()
        _close(want["pos_score"].cpu(), z[p + "pos_score"], 1e-4, 1e-4, name + " pos_score")
        assert tuple(want["neg_score"].shape) == z[p + "neg_score"].shape
        _close(want["neg_score"].cpu(), z[p + "neg_score"], 1e-4, 1e-4, name + " neg_score")
        l4 = eng.read_loss()
        ref_log = z[p + "log"]
        _close(l4[0], ref_log[0], 1e-4, 1e-5, name + " pos_loss")
        _close(l4[1], ref_log[1], 1e-4, 1e-5, name + " neg_loss")
        _close(l4[2], ref_log[2], 1e-4, 1e-5, name + " loss")
        _close(l4[3], ref_log[3], 1e-4, 1e-7, name + " reg")
        sel = np.searchsorted(b.p["ue_id"], z[p + "nid"])
        _close(want["g_pos_ent"].cpu().numpy()[sel], z[p + "g_pos_ent"], 3e-4, grad_tol(z[p + "g_pos_ent"]), name + " g_pos_ent")
        # the sampled negatives are rows chunk.. of every chunk's N' block; the regulariser of those rows is added by
        # the update kernel in this mode (the reference's trace gradient includes it)
        gneg = want["g_neg"].cpu().numpy().reshape(-1, Np, want["g_neg"].shape[1])[:, chunk:].reshape(-1, want["g_neg"].shape[1])
        ref_gneg = z[p + "g_neg"]
        if case["reg_coef"] > 0:
            negrows = (z["init_entity"] if s == 1 else prev_ent)[z[p + "neg"]]
            ref_gneg = ref_gneg - O.reg_grad(negrows.astype(np.float64), case["reg_coef"], case["reg_norm"])
        _close(gneg, ref_gneg, 3e-4, grad_tol(z[p + "g_neg"]), name + " g_neg")
        _close(want["g_rel"].cpu(), z[p + "g_rel"], 3e-4, grad_tol(z[p + "g_rel"]), name + " g_rel")
        _close(eng.ent_state.cpu(), z[p + "entity_state"], 2e-3, 1e-9, name + " ent state")
        _close(eng.rel_state.cpu(), z[p + "relation_state"], 2e-3, 1e-9, name + " rel state")
        if (p + "entity") in z:
            rows_close(eng.ent.cpu(), z[p + "entity"], name, "entity", case["lr"], "entity rows")
            rows_close(eng.rel.cpu(), z[p + "relation"], name, "relation", case["lr"], "relation rows")
        prev_ent = eng.ent.cpu().numpy()
    rows_close(eng.ent.cpu(), z["final_entity"], name, "entity", case["lr"], "final entity")
    rows_close(eng.rel.cpu(), z["final_relation"], name, "relation", case["lr"], "final relation")


@pytest.mark.parametrize("name", [n for n in golden_names(nd=True, transr=False) if "rescal" not in n])   # (the emitting step has neither)
def test_dist_engine_neg_deg_sample_matches_reference(name):
    """round 4: --neg_deg_sample in the gradient-emitting step (kge_step_grads), i.e. in the north_star multi-GPU mode - the
    12 nd_* goldens through a world-1 DistEngine (route -> pull -> step against the row cache -> packed messages -> owner-side
    merged apply): both Adagrad states after every step and the final tables against the reference's."""
    from dglke_amd import dist as kd
    z, case = load_golden(name)
    m = build_model(case, z)
    eng = m.engine
    if eng.d_e % 4 or eng.d_r % 4:
        pytest.skip("the owner-side merged apply moves 16-byte packs: row widths must be multiples of 4 (every BASELINE recipe's are)")
    eng.hp.flags = 32
    ent, state = eng.ent, eng.ent_state
    deng = kd.DistEngine(eng, kd.ShardSpec(case["n_ent"], 1, 0), ent, state)
    ue_bound = None
    for s in range(1, case["steps"] + 1):
        p = "s%d_" % s
        b = golden_batch(z, case, s)
        if s == 1:
            ue_bound = 2 * b.B + b.C * b.N
        b.UE = ue_bound                       # the engine sizes its buffers once, for the bound
        deng.step(b)
        torch.cuda.synchronize()
        _close(state.cpu(), z[p + "entity_state"], 2e-3, 1e-9, name + " ent state")
        _close(eng.rel_state.cpu(), z[p + "relation_state"], 2e-3, 1e-9, name + " rel state")
        if (p + "entity") in z:
            rows_close(ent.cpu(), z[p + "entity"], name, "entity", case["lr"], "entity rows (DistEngine)")
            rows_close(eng.rel.cpu(), z[p + "relation"], name, "relation", case["lr"], "relation rows (DistEngine)")
    assert deng.check_overflow() == 0
    rows_close(ent.cpu(), z["final_entity"], name, "entity", case["lr"], "final entity (DistEngine)")
    rows_close(eng.rel.cpu(), z["final_relation"], name, "relation", case["lr"], "final relation (DistEngine)")


def _random_step_case(seed):
    """one random small configuration: model, shapes (ragged on purpose), loss options, kernel-path flags"""
    rng = np.random.RandomState(seed)
    model = ["TransE_l1", "TransE_l2", "DistMult", "ComplEx", "RotatE", "SimplE"][seed % 6]
    de = model in ("ComplEx", "RotatE", "SimplE")
    dr = model in ("ComplEx", "SimplE")
    hidden = int(rng.choice([8, 12, 16, 20, 24, 32, 40, 48, 64]))
    chunk = int(rng.choice([1, 3, 4, 7, 8, 16, 17, 24, 32, 40]))
    C = int(rng.randint(1, 5))
    N = int(rng.choice([1, 2, 4, 5, 8, 12, 16, 20, 32, 36, 64]))
    flags = int(rng.choice([0, 0, 0, 1, 2, 8, 10, 16, 32, 33, 128, 130, 512, 256, 768, 1024, 1024 + 256]))    # 1024: loss rows in the first launch
    return dict(model=model, de=de, dr=dr, hidden=hidden, chunk=chunk, C=C, N=N, flags=flags, impts=bool(rng.randint(4) == 0),
                n_ent=int(rng.choice([30, 200, 2000])), n_rel=int(rng.choice([3, 17])),
                adv=bool(rng.randint(2)), reg=float(rng.choice([0.0, 1e-4])), gamma=float(rng.choice([6.0, 12.0])),
                lr=float(rng.choice([0.05, 0.2])))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("KGE_FUZZ_N", "48"))))   # KGE_FUZZ_N=500 for a longer hunt
def test_fused_step_random_shapes_match_oracle(seed):
    """fuzz: 48 random (model, chunk, N, width, options, kernel-path flag) combinations - two fused steps (tail then head
    corruption) against the fp64 oracle started from the same tables: scores, loss, the three trace gradients, Adagrad
    states and rows.  Exercises ragged tiles of every negative-score kernel (matrix-core, pairwise, shared-pair), the
    TransE fast path on and off, and --neg_deg_sample."""
    from dglke_amd import plan
    from dglke_amd.engine import StepEngine
    k = _random_step_case(seed)
    nd = bool(k["flags"] & 32)
    cfg = O.Config(k["model"], k["gamma"], k["hidden"], k["lr"], adv=k["adv"], adv_temp=1.0, reg_coef=k["reg"], reg_norm=3,
                   double_ent=k["de"], double_rel=k["dr"], neg_deg=nd)
    rng = np.random.RandomState(1000 + seed)
    ent = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(k["n_ent"], cfg.ent_dim)).astype(np.float32)
    rel = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(k["n_rel"], cfg.rel_dim)).astype(np.float32)
    eng = StepEngine(k["model"], k["n_ent"], k["n_rel"], k["hidden"], k["gamma"], k["lr"], DEV, k["de"], k["dr"], k["adv"], 1.0,
                     k["reg"], 3, flags=k["flags"])
    eng.load_tables(ent, rel)
    B, chunk, N = k["C"] * k["chunk"], k["chunk"], k["N"]
    Np = chunk + N if nd else N
    tag0 = "seed %d %s" % (seed, k)
    for step in (1, 2):
        ent64 = eng.ent.cpu().numpy().astype(np.float64)
        rel64 = eng.rel.cpu().numpy().astype(np.float64)
        es64 = eng.ent_state.cpu().numpy().astype(np.float64)
        rs64 = eng.rel_state.cpu().numpy().astype(np.float64)
        bt = O.synth_batch(rng, k["n_ent"], k["n_rel"], B, N, chunk, step)
        # edge importance in a quarter of the cases (loss.py:72-83: the negative part per edge, the positive part by the batch mean)
        w = rng.uniform(0.5, 1.5, size=B).astype(np.float32) if k["impts"] else None
        b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], chunk, N, bt["neg_head"], DEV, w)
        want = eng.alloc_outputs(b)
        eng.step(b, want)
        torch.cuda.synchronize()
        negrows = ent64[bt["neg"]]
        out = O.train_step(cfg, ent64, es64, rel64, rs64, bt["nid"], bt["h_local"], bt["t_local"],
                           bt["r"], bt["neg"], bt["neg_head"], chunk, N, None if w is None else w.astype(np.float64))
        tag = "%s step %d" % (tag0, step)
        _close(want["pos_score"].cpu(), out["pos_score"], 1e-4, 1e-4, tag + " pos_score")
        _close(want["neg_score"].cpu(), out["neg_score"], 1e-4, 1e-4, tag + " neg_score")
        l4 = eng.read_loss()
        _close(l4[:3], out["log"][:3], 1e-4, 1e-5, tag + " loss")
        sel = np.searchsorted(b.p["ue_id"], bt["nid"])
        _close(want["g_pos_ent"].cpu().numpy()[sel], out["g_pos_ent"], 3e-4, grad_tol(out["g_pos_ent"]), tag + " g_pos_ent")
        gneg = want["g_neg"].cpu().numpy()
        ref_gneg = out["g_neg"]
        if nd:      # sampled rows only; the regulariser of those rows is added by the update kernel in this mode
            gneg = gneg.reshape(-1, Np, gneg.shape[1])[:, chunk:].reshape(-1, gneg.shape[1])
            if k["reg"] > 0:
                ref_gneg = ref_gneg - O.reg_grad(negrows, k["reg"], 3)
        _close(gneg, ref_gneg, 3e-4, grad_tol(out["g_neg"]), tag + " g_neg")
        _close(want["g_rel"].cpu(), out["g_rel"], 3e-4, grad_tol(out["g_rel"]), tag + " g_rel")
        _close(eng.ent_state.cpu(), es64, 2e-3, 1e-9, tag + " ent state")
        _close(eng.rel_state.cpu(), rs64, 2e-3, 1e-9, tag + " rel state")
        _close(eng.ent.cpu(), ent64, 1e-4, 5e-3 * k["lr"], tag + " entity rows")
        _close(eng.rel.cpu(), rel64, 1e-4, 5e-3 * k["lr"], tag + " relation rows")


def _skewed_batch(rng, n_ent, n_rel, B, N, chunk, step, hub_frac, rel_frac):
    """ids with hubs: entity 7 on `hub_frac` of the heads and tails and among the negatives, relation 1 on `rel_frac` of the
    edges - contribution lists of hundreds of entries (real graphs are heavy-tailed; uniform ids never produce them)."""
    C = B // chunk
    h = rng.randint(0, n_ent, size=B).astype(np.int64)
    t = rng.randint(0, n_ent, size=B).astype(np.int64)
    r = rng.randint(0, n_rel, size=B).astype(np.int64)
    neg = rng.randint(0, n_ent, size=C * N).astype(np.int64)
    h[rng.rand(B) < hub_frac] = 7
    t[rng.rand(B) < hub_frac] = 7
    r[rng.rand(B) < rel_frac] = 1
    neg[rng.rand(C * N) < hub_frac] = 7
    nid, inv = np.unique(np.concatenate([h, t]), return_inverse=True)
    return dict(h=h, t=t, r=r, neg=neg, neg_head=(step % 2 == 0), nid=nid.astype(np.int64),
                h_local=inv[:B].astype(np.int64), t_local=inv[B:].astype(np.int64), C=C)


HEAVY = [
    # model, hidden, de, dr, B, chunk, N, hub_frac, rel_frac, flags
    ("TransE_l2", 400, False, False, 1000, 200, 200, 0.02, 0.04, 0),     # FB15k's proportions: lists of 20 - 40 entries
    ("TransE_l2", 400, False, False, 1000, 200, 200, 0.15, 0.50, 0),     # lists longer than 64 entries (the serial remainder loops)
    ("TransE_l2", 100, False, False, 600, 100, 50, 0.15, 0.50, 0),       # one pack per lane
    ("TransE_l2", 400, False, False, 1000, 200, 200, 0.15, 0.50, 2),     # per-edge gradients instead of the TransE fast path
    ("TransE_l1", 400, False, False, 600, 100, 64, 0.15, 0.50, 0),       # TransE fast path with two source rows (P and GA)
    ("DistMult", 400, False, False, 600, 100, 64, 0.15, 0.50, 0),
    ("RotatE", 400, True, False, 512, 128, 64, 0.15, 0.50, 0),           # D_e = 800: four packs per lane
    ("ComplEx", 100, True, True, 600, 100, 50, 0.15, 0.50, 0),
]


@pytest.mark.parametrize("case", HEAVY, ids=lambda c: "%s-D%d-B%d-hub%g-rel%g-f%d" % (c[0], c[1], c[4], c[7], c[8], c[9]))
def test_fused_step_heavy_lists_match_oracle(case):
    """hub entities / a dominant relation: the update kernel's contribution lists (requested several entries at a time, the first
    64 from a register-resident index vector, the rest one by one) against the fp64 oracle - two steps, tail then head
    corruption, with and without the regulariser."""
    from dglke_amd import plan
    from dglke_amd.engine import StepEngine
    model, hidden, de, dr, B, chunk, N, hub, relf, flags = case
    n_ent, n_rel = 3000, 11
    for reg in (0.0, 1e-5):
        cfg = O.Config(model, 12.0, hidden, 0.1, adv=True, adv_temp=1.0, reg_coef=reg, reg_norm=3, double_ent=de, double_rel=dr)
        rng = np.random.RandomState(77)
        ent = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(n_ent, cfg.ent_dim)).astype(np.float32)
        rel = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(n_rel, cfg.rel_dim)).astype(np.float32)
        eng = StepEngine(model, n_ent, n_rel, hidden, 12.0, 0.1, DEV, de, dr, True, 1.0, reg, 3, flags=flags)
        eng.load_tables(ent, rel)
        for step in (1, 2):
            ent64 = eng.ent.cpu().numpy().astype(np.float64)
            rel64 = eng.rel.cpu().numpy().astype(np.float64)
            es64 = eng.ent_state.cpu().numpy().astype(np.float64)
            rs64 = eng.rel_state.cpu().numpy().astype(np.float64)
            bt = _skewed_batch(rng, n_ent, n_rel, B, N, chunk, step, hub, relf)
            b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], chunk, N, bt["neg_head"], DEV)
            want = eng.alloc_outputs(b)
            eng.step(b, want)
            torch.cuda.synchronize()
            out = O.train_step(cfg, ent64, es64, rel64, rs64, bt["nid"], bt["h_local"], bt["t_local"],
                               bt["r"], bt["neg"], bt["neg_head"], chunk, N)
            tag = "%s reg %g step %d" % (case, reg, step)
            _close(want["neg_score"].cpu(), out["neg_score"], 1e-4, 1e-4, tag + " neg_score")
            _close(eng.read_loss()[:3], out["log"][:3], 1e-4, 1e-5, tag + " loss")
            _close(eng.ent_state.cpu(), es64, 2e-3, 1e-9, tag + " ent state")
            _close(eng.rel_state.cpu(), rs64, 2e-3, 1e-9, tag + " rel state")
            _close(eng.ent.cpu(), ent64, 1e-4, 5e-3 * 0.1, tag + " entity rows")
            _close(eng.rel.cpu(), rel64, 1e-4, 5e-3 * 0.1, tag + " relation rows")


# ---------------------------------------------------------------------------------------------
# BASELINE configs at their REAL table sizes (cfg-C: 2.5 M entities = 4 GB, beyond Infinity Cache; cfg-R: the
# per-GPU step of the Freebase RotatE config, D_e = 800 / D_r = 400 over >= 1 M entities).  The oracle works on
# the compacted sub-table of the rows the step touches (same ids through a relabelling - the arithmetic does not
# depend on the id values); the untouched part of the 3-4 GB table must stay bit-identical.
# ---------------------------------------------------------------------------------------------
BIG = [
    # model, n_ent, n_rel, hidden, de, dr, B, N, gamma, lr, reg
    ("ComplEx", 2500604, 535, 200, True, True, 1024, 256, 143.0, 0.1, 2e-6),        # cfg-C, real table size
    ("RotatE", 1000003, 14824, 400, True, False, 1024, 256, 12.0, 0.01, 1e-7),      # cfg-R per-GPU step, D_e = 800
    ("TransE_l2", 1000003, 1345, 400, True, True, 1000, 200, 19.9, 0.25, 1e-9),     # D_e = D_r = 800 through the GEMM path
]


@pytest.mark.parametrize("shape", BIG, ids=lambda s: "%s-ent%d-D%d" % (s[0], s[1], s[3] * (2 if s[4] else 1)))
def test_fused_step_matches_oracle_at_real_table_sizes(shape):
    from dglke_amd import plan
    from dglke_amd.engine import StepEngine
    model, n_ent, n_rel, hidden, de, dr, B, N, gamma, lr, reg = shape
    cfg = O.Config(model, gamma, hidden, lr, adv=True, adv_temp=1.0, reg_coef=reg, reg_norm=3, double_ent=de, double_rel=dr)
    torch.manual_seed(5)
    eng = StepEngine(model, n_ent, n_rel, hidden, gamma, lr, DEV, de, dr, True, 1.0, reg, 3)
    rng = np.random.RandomState(99)
    for step in (1, 2):
        bt = O.synth_batch(rng, n_ent, n_rel, B, N, N, step)
        if step == 2:           # make the second step revisit rows the first one updated (state + row carried over)
            bt["h"][:64] = prev["h"][:64]
            bt["neg"][:64] = prev["neg"][:64]
            nid, inv = np.unique(np.concatenate([bt["h"], bt["t"]]), return_inverse=True)
            bt.update(nid=nid.astype(np.int64), h_local=inv[:B].astype(np.int64), t_local=inv[B:].astype(np.int64))
        prev = bt
        touched = np.unique(np.concatenate([bt["h"], bt["t"], bt["neg"]]))
        tix = torch.from_numpy(touched).to(DEV)
        before_ent, before_state = eng.ent.clone(), eng.ent_state.clone()
        ent_sub = eng.ent[tix].cpu().numpy().astype(np.float64)
        es_sub = eng.ent_state[tix].cpu().numpy().astype(np.float64)
        rel64 = eng.rel.cpu().numpy().astype(np.float64)
        rs64 = eng.rel_state.cpu().numpy().astype(np.float64)
        b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"], DEV)
        want = eng.alloc_outputs(b)
        eng.step(b, want)
        torch.cuda.synchronize()
        loc = lambda x: np.searchsorted(touched, x).astype(np.int64)
        out = O.train_step(cfg, ent_sub, es_sub, rel64, rs64, loc(bt["nid"]), bt["h_local"], bt["t_local"],
                           bt["r"], loc(bt["neg"]), bt["neg_head"], N, N)
        tag = "%s n_ent=%d step %d" % (model, n_ent, step)
        _close(want["pos_score"].cpu(), out["pos_score"], 1e-4, 1e-4, tag + " pos_score")
        _close(want["neg_score"].cpu(), out["neg_score"], 1e-4, 1e-4, tag + " neg_score")
        l4 = eng.read_loss()
        _close(l4[:3], out["log"][:3], 1e-4, 1e-5, tag + " loss")
        sel = np.searchsorted(b.p["ue_id"], bt["nid"])
        _close(want["g_pos_ent"].cpu().numpy()[sel], out["g_pos_ent"], 3e-4, grad_tol(out["g_pos_ent"]), tag + " g_pos_ent")
        _close(want["g_neg"].cpu(), out["g_neg"], 3e-4, grad_tol(out["g_neg"]), tag + " g_neg")
        _close(want["g_rel"].cpu(), out["g_rel"], 3e-4, grad_tol(out["g_rel"]), tag + " g_rel")
        _close(eng.ent_state[tix].cpu(), es_sub, 2e-3, 1e-9, tag + " ent state")
        _close(eng.rel_state.cpu(), rs64, 2e-3, 1e-9, tag + " rel state")
        _close(eng.ent[tix].cpu(), ent_sub, 1e-4, 1e-3 * lr, tag + " entity rows")      # (the bound of the config-shape test above)
        _close(eng.rel.cpu(), rel64, 1e-4, 1e-3 * lr, tag + " relation rows")
        # every row the step did not name is bit-identical
        keep = torch.ones(n_ent, dtype=torch.bool, device=DEV)
        keep[tix] = False
        assert torch.equal(eng.ent[keep], before_ent[keep]), tag + ": an untouched entity row changed"
        assert torch.equal(eng.ent_state[keep], before_state[keep]), tag + ": an untouched Adagrad state changed"
        del before_ent, before_state, keep


def _wide_step_case(seed):
    """random configuration at the BASELINE row widths: hidden in {100, 200, 400}, doubled rows where the recipes double them"""
    rng = np.random.RandomState(5000 + seed)
    model = ["TransE_l1", "TransE_l2", "DistMult", "ComplEx", "RotatE", "SimplE"][seed % 6]
    de = model in ("ComplEx", "RotatE", "SimplE") or bool(rng.randint(2))
    dr = de if model != "RotatE" else False
    hidden = int(rng.choice([100, 200, 400]))
    chunk = int(rng.choice([8, 16, 24, 40, 64]))
    C = int(rng.randint(1, 4))
    N = int(rng.choice([8, 16, 20, 32, 64, 72]))
    flags = int(rng.choice([0, 0, 0, 8, 8, 1, 2, 16, 32, 128, 512, 256]))
    return dict(model=model, de=de, dr=dr, hidden=hidden, chunk=chunk, C=C, N=N, flags=flags,
                n_ent=int(rng.choice([100, 3000])), n_rel=int(rng.choice([5, 40])),
                adv=bool(rng.randint(2)), reg=float(rng.choice([0.0, 1e-6])), gamma=float(rng.choice([12.0, 19.9])),
                lr=float(rng.choice([0.05, 0.25])))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("KGE_FUZZ_WIDE_N", "24"))))
def test_fused_step_wide_rows_match_oracle(seed):
    """fuzz at the BASELINE row widths (hidden 100 / 200 / 400, with -de [-dr]: rows of 100 .. 800 floats): selects the
    NIT = 1 / 2 / 4 instantiations of the update kernel, the multi-slab paths of the shared-pair backward and every
    d-tile count of the backward GEMM - two fused steps against the fp64 oracle from the same tables."""
    from dglke_amd import plan
    from dglke_amd.engine import StepEngine
    k = _wide_step_case(seed)
    nd = bool(k["flags"] & 32)
    cfg = O.Config(k["model"], k["gamma"], k["hidden"], k["lr"], adv=k["adv"], adv_temp=1.0, reg_coef=k["reg"], reg_norm=3,
                   double_ent=k["de"], double_rel=k["dr"], neg_deg=nd)
    rng = np.random.RandomState(7000 + seed)
    ent = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(k["n_ent"], cfg.ent_dim)).astype(np.float32)
    rel = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(k["n_rel"], cfg.rel_dim)).astype(np.float32)
    eng = StepEngine(k["model"], k["n_ent"], k["n_rel"], k["hidden"], k["gamma"], k["lr"], DEV, k["de"], k["dr"], k["adv"], 1.0,
                     k["reg"], 3, flags=k["flags"])
    eng.load_tables(ent, rel)
    B, chunk, N = k["C"] * k["chunk"], k["chunk"], k["N"]
    Np = chunk + N if nd else N
    tag0 = "wide seed %d %s" % (seed, k)
    for step in (1, 2):
        ent64 = eng.ent.cpu().numpy().astype(np.float64)
        rel64 = eng.rel.cpu().numpy().astype(np.float64)
        es64 = eng.ent_state.cpu().numpy().astype(np.float64)
        rs64 = eng.rel_state.cpu().numpy().astype(np.float64)
        bt = O.synth_batch(rng, k["n_ent"], k["n_rel"], B, N, chunk, step)
        b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], chunk, N, bt["neg_head"], DEV)
        want = eng.alloc_outputs(b)
        eng.step(b, want)
        torch.cuda.synchronize()
        negrows = ent64[bt["neg"]]
        out = O.train_step(cfg, ent64, es64, rel64, rs64, bt["nid"], bt["h_local"], bt["t_local"],
                           bt["r"], bt["neg"], bt["neg_head"], chunk, N)
        tag = "%s step %d" % (tag0, step)
        _close(want["pos_score"].cpu(), out["pos_score"], 1e-4, 1e-4, tag + " pos_score")
        _close(want["neg_score"].cpu(), out["neg_score"], 1e-4, 1e-4, tag + " neg_score")
        l4 = eng.read_loss()
        _close(l4[:3], out["log"][:3], 1e-4, 1e-5, tag + " loss")
        sel = np.searchsorted(b.p["ue_id"], bt["nid"])
        _close(want["g_pos_ent"].cpu().numpy()[sel], out["g_pos_ent"], 3e-4, grad_tol(out["g_pos_ent"]), tag + " g_pos_ent")
        gneg = want["g_neg"].cpu().numpy()
        ref_gneg = out["g_neg"]
        if nd:
            gneg = gneg.reshape(-1, Np, gneg.shape[1])[:, chunk:].reshape(-1, gneg.shape[1])
            if k["reg"] > 0:
                ref_gneg = ref_gneg - O.reg_grad(negrows, k["reg"], 3)
        _close(gneg, ref_gneg, 3e-4, grad_tol(out["g_neg"]), tag + " g_neg")
        _close(want["g_rel"].cpu(), out["g_rel"], 3e-4, grad_tol(out["g_rel"]), tag + " g_rel")
        _close(eng.ent_state.cpu(), es64, 2e-3, 1e-9, tag + " ent state")
        _close(eng.rel_state.cpu(), rs64, 2e-3, 1e-9, tag + " rel state")
        _close(eng.ent.cpu(), ent64, 1e-4, 5e-3 * k["lr"], tag + " entity rows")
        _close(eng.rel.cpu(), rel64, 1e-4, 5e-3 * k["lr"], tag + " relation rows")
