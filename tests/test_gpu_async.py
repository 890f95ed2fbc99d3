"""--async_update on the HIP path (kge_step_async: UPDATE(s-1) on a side stream under SCORE(s)) against the
oracle's deterministic restatement of the reference's async mode (oracle.kge_oracle.train_steps_async):
tables after a group of steps, bit-reproducibility, hipGraph replay == eager, and the strict step as control."""
import numpy as np
import pytest
import torch

from oracle import kge_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(a, b, rtol, atol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape and np.all(np.isfinite(a)), what
    err = np.abs(a - b)
    bad = err > atol + rtol * np.abs(b)
    assert not bad.any(), "%s: %d/%d out of tolerance, max err %.3e" % (what, bad.sum(), bad.size, err.max())


CASES = [
    # model, n_ent, n_rel, hidden, de, dr, chunk, C, N, gamma, lr, reg, flags
    ("TransE_l2", 60, 7, 32, False, False, 16, 2, 16, 10.0, 0.1, 0.0, 0),        # 60 entities: every step re-touches rows
    ("TransE_l2", 60, 7, 32, False, False, 16, 2, 16, 10.0, 0.1, 0.0, 64),       # relation trace deferred too
    ("TransE_l2", 400, 9, 64, False, False, 40, 2, 24, 12.0, 0.25, 1e-9, 0),
    ("TransE_l1", 80, 5, 32, False, False, 16, 2, 16, 10.0, 0.05, 0.0, 0),
    ("DistMult", 80, 5, 32, False, False, 16, 2, 16, 12.0, 0.1, 0.0, 0),         # edge_bwd reads the dense h / t / r copies
    ("ComplEx", 80, 5, 16, True, True, 16, 2, 16, 12.0, 0.1, 0.0, 64),
    ("RotatE", 80, 5, 16, True, False, 16, 2, 20, 10.0, 0.05, 0.0, 0),
    ("SimplE", 80, 5, 16, True, True, 16, 1, 16, 12.0, 0.1, 0.0, 0),
    ("TransE_l2", 80, 5, 32, False, False, 16, 2, 16, 10.0, 0.1, 0.0, 8),        # with the fused loss
    ("DistMult", 80, 5, 32, False, False, 16, 2, 16, 12.0, 0.1, 0.0, 32),        # --neg_deg_sample
    ("RotatE", 80, 5, 16, True, False, 16, 2, 20, 10.0, 0.05, 1e-4, 32),         # --neg_deg_sample + regulariser (FB15k RotatE recipe)
    ("TransE_l2", 80, 5, 32, False, False, 16, 2, 16, 10.0, 0.1, 1e-4, 32),
    ("ComplEx", 80, 5, 16, True, True, 16, 2, 16, 12.0, 0.1, 1e-4, 64),          # regulariser on the copies, relation deferred
]


def _setup(case, seed=3, steps=5):
    from dglke_amd import plan
    from dglke_amd.engine import StepEngine
    model, n_ent, n_rel, hidden, de, dr, chunk, C, N, gamma, lr, reg, flags = case
    cfg = O.Config(model, gamma, hidden, lr, adv=True, adv_temp=1.0, reg_coef=reg, reg_norm=3, double_ent=de, double_rel=dr,
                   neg_deg=bool(flags & 32))
    rng = np.random.RandomState(seed)
    ent = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(n_ent, cfg.ent_dim)).astype(np.float32)
    rel = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(n_rel, cfg.rel_dim)).astype(np.float32)
    bts = []
    for s in range(1, steps + 1):
        bt = O.synth_batch(rng, n_ent, n_rel, C * chunk, N, chunk, s)
        bt.update(chunk=chunk, N=N)
        bts.append(bt)

    def engine():
        e = StepEngine(model, n_ent, n_rel, hidden, gamma, lr, DEV, de, dr, True, 1.0, reg, 3, flags=flags)
        e.load_tables(ent, rel)
        return e
    batches = [plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], chunk, N, bt["neg_head"], DEV) for bt in bts]
    return cfg, ent, rel, bts, batches, engine


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-ent%d-flags%d" % (c[0], c[1], c[12]))
def test_async_pipeline_matches_the_stale_oracle(case):
    cfg, ent, rel, bts, batches, engine = _setup(case)
    lr, defer_rel = case[10], bool(case[12] & 64)
    # oracle, fp64: the reference's async semantics with the race fixed at its bound
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    es, rs = np.zeros(len(ent)), np.zeros(len(rel))
    O.train_steps_async(cfg, e64, es, r64, rs, bts, defer_rel=defer_rel)
    # control: the strict step gives DIFFERENT tables on these overlapping batches (the test is not vacuous)
    s64, sr64 = ent.astype(np.float64), rel.astype(np.float64)
    ses, srs = np.zeros(len(ent)), np.zeros(len(rel))
    for bt in bts:
        O.train_step(cfg, s64, ses, sr64, srs, bt["nid"], bt["h_local"], bt["t_local"], bt["r"], bt["neg"], bt["neg_head"],
                     bt["chunk"], bt["N"])
    assert np.abs(s64 - e64).max() > 1e-3 * lr, "strict and async oracles agree: the batches do not overlap"

    runs = []
    for mode in ("eager", "eager", "graph"):
        eng = engine()
        if mode == "eager":
            eng.steps_async(batches)
        else:
            g = eng.capture(batches, async_update=True)
            g.replay()
        torch.cuda.synchronize()
        runs.append([x.cpu().numpy().copy() for x in (eng.ent, eng.ent_state, eng.rel, eng.rel_state)])
    tag = "%s flags %d" % (case[0], case[12])
    _close(runs[0][1], es, 2e-3, 1e-9, tag + " entity state")
    _close(runs[0][3], rs, 2e-3, 1e-9, tag + " relation state")
    _close(runs[0][0], e64, 1e-4, 1e-2 * lr, tag + " entity table")
    _close(runs[0][2], r64, 1e-4, 1e-2 * lr, tag + " relation table")
    # and NOT the strict result
    assert np.abs(runs[0][0] - s64).max() > 1e-3 * lr
    for k in range(4):          # two streams, still bit-reproducible; graph replay == eager
        assert np.array_equal(runs[0][k], runs[1][k]), "%s: run-to-run difference in output %d" % (tag, k)
        assert np.array_equal(runs[0][k], runs[2][k]), "%s: graph replay differs in output %d" % (tag, k)


def test_async_group_of_one_step_is_the_strict_step():
    """a flush after every step leaves nothing in flight: identical to kge_step_fused, bit for bit"""
    cfg, ent, rel, bts, batches, engine = _setup(CASES[2], steps=3)
    a, b = engine(), engine()
    for bt in batches:
        a.step(bt)
        b.step_async(bt)
        b.flush_async()
    torch.cuda.synchronize()
    for x, y in ((a.ent, b.ent), (a.ent_state, b.ent_state), (a.rel, b.rel), (a.rel_state, b.rel_state)):
        assert torch.equal(x, y)


def test_async_at_cfg_t_shape_is_deterministic_and_close_to_oracle():
    """cfg-T shape (B 1000, N 200, D 400, 14 951 entities): a group of 6 steps, twice; the touched rows against the oracle"""
    case = ("TransE_l2", 14951, 1345, 400, False, False, 200, 5, 200, 19.9, 0.25, 1e-9, 0)
    cfg, ent, rel, bts, batches, engine = _setup(case, seed=11, steps=6)
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    es, rs = np.zeros(len(ent)), np.zeros(len(rel))
    O.train_steps_async(cfg, e64, es, r64, rs, bts)
    outs = []
    for _ in range(2):
        eng = engine()
        g = eng.capture(batches, async_update=True)
        g.replay()
        torch.cuda.synchronize()
        outs.append((eng.ent.cpu().numpy().copy(), eng.ent_state.cpu().numpy().copy(), eng.rel.cpu().numpy().copy()))
    assert all(np.array_equal(outs[0][k], outs[1][k]) for k in range(3))
    _close(outs[0][1], es, 2e-3, 1e-9, "entity state")
    _close(outs[0][0], e64, 1e-4, 1e-2 * 0.25, "entity table")
    _close(outs[0][2], r64, 1e-4, 1e-2 * 0.25, "relation table")


def test_dropin_model_async_update_matches_the_stale_oracle():
    """the reference loop with --async_update on the drop-in classes (train_pytorch.py:120-121, 141-152, 194-195):
    KEModel.create_async_update() -> [forward, backward, update] x n -> finish_async_update(); the entity updates land
    one step late, the relation updates at once - against oracle.train_steps_async."""
    from test_gpu_parity import make_args
    from dglke_amd import plan
    from dglke_amd.dataloader import NegGraph, PosGraph
    from dglke_amd.general_models import KEModel
    case = dict(model="DistMult", n_ent=70, n_rel=6, hidden=32, gamma=12.0, lr=0.1, reg_coef=1e-5, reg_norm=3, adv=True,
                adv_temp=1.0, de=False, dr=False)
    chunk, N, C = 16, 16, 2
    cfg = O.Config(case["model"], case["gamma"], case["hidden"], case["lr"], adv=True, adv_temp=1.0, reg_coef=case["reg_coef"],
                   reg_norm=3)
    rng = np.random.RandomState(2)
    ent = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(case["n_ent"], cfg.ent_dim)).astype(np.float32)
    rel = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(case["n_rel"], cfg.rel_dim)).astype(np.float32)
    bts = []
    for s in range(1, 6):
        bt = O.synth_batch(rng, case["n_ent"], case["n_rel"], C * chunk, N, chunk, s)
        bt.update(chunk=chunk, N=N)
        bts.append(bt)
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64)
    es, rs = np.zeros(len(ent)), np.zeros(len(rel))
    O.train_steps_async(cfg, e64, es, r64, rs, bts)
    m = KEModel(make_args(case), case["model"], case["n_ent"], case["n_rel"], case["hidden"], case["gamma"])
    m.entity_emb.emb.copy_(torch.from_numpy(ent)); m.relation_emb.emb.copy_(torch.from_numpy(rel))
    m.entity_emb.state_sum.zero_(); m.relation_emb.state_sum.zero_()
    m.create_async_update()
    for bt in bts:
        b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], chunk, N, bt["neg_head"], DEV)
        loss, log = m.forward(PosGraph(b), NegGraph(b), 0)
        loss.backward()
        m.update(0)
    m.finish_async_update()
    torch.cuda.synchronize()
    _close(m.entity_emb.state_sum.cpu(), es, 2e-3, 1e-9, "entity state")
    _close(m.entity_emb.emb.cpu(), e64, 1e-4, 1e-2 * case["lr"], "entity table")
    _close(m.relation_emb.emb.cpu(), r64, 1e-4, 1e-2 * case["lr"], "relation table")


def _async_goldens():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from golden_util import async_golden_names
    return async_golden_names()


@pytest.mark.parametrize("name", _async_goldens())
def test_async_pipeline_matches_the_reference_async_update(name):
    """kge_step_async against goldens recorded from the REFERENCE's own --async_update code path (the unmodified async_update
    loop body applying every step's entity traces one step late; tests/golden/gen_golden.py): final tables and states."""
    from golden_util import load_golden
    from dglke_amd import plan
    from dglke_amd.engine import StepEngine
    z, case = load_golden(name)
    eng = StepEngine(case["model"], case["n_ent"], case["n_rel"], case["hidden"], case["gamma"], case["lr"], DEV, case["de"],
                     case["dr"], case["adv"], case["adv_temp"], case["reg_coef"], case["reg_norm"])
    eng.load_tables(z["init_entity"], z["init_relation"])
    batches = []
    for s in range(1, case["steps"] + 1):
        p = "s%d_" % s
        batches.append(plan.make_batch(z[p + "h"], z[p + "t"], z[p + "r"], z[p + "neg"], case["chunk"], case["N"],
                                       bool(z[p + "neg_head"]), DEV))
    eng.steps_async(batches)
    torch.cuda.synchronize()
    lr = case["lr"]
    _close(eng.ent_state.cpu().numpy(), z["final_entity_state"], 2e-3, 1e-9, name + " entity state")
    _close(eng.rel_state.cpu().numpy(), z["final_relation_state"], 2e-3, 1e-9, name + " relation state")
    _close(eng.ent.cpu().numpy(), z["final_entity"], 1e-4, 1e-2 * lr, name + " entity table")
    _close(eng.rel.cpu().numpy(), z["final_relation"], 1e-4, 1e-2 * lr, name + " relation table")
