"""bench.py's CPU baseline of kind "reference" (oracle/make_ref.py + oracle/ref_baseline.py): the reference's own six hot-path
files, compiled unmodified (py_compile: sourceless .pyc modules) into oracle/_ref at build() time, drive the timed CPU step.  These tests pin that what is staged IS the
reference (one golden case reproduced through the staged package) and that the baseline harness runs.  Skipped where neither
/root/reference nor a staged copy exists."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dgl-ke_amd"))


def _staged():
    from oracle import make_ref
    return make_ref.stage()


def test_make_ref_recipe_names_the_six_hot_path_files_and_keeps_them_out_of_git():
    from oracle import make_ref
    assert len(make_ref.FILES) == 6 and all(f.endswith(".py") for f in make_ref.FILES)
    ign = open(os.path.join(ROOT, ".gitignore")).read().split()
    assert "oracle/_ref/" in ign                                     # never committed ...
    gpi = open(os.path.join(ROOT, ".gpurunignore")).read().split()
    assert not any(x.startswith("oracle") for x in gpi)             # ... but it travels to the GPU box
    if make_ref.stage():                                            # binaries only: no reference source text in the tree
        staged = [f for d, _, fs in os.walk(make_ref.DST) if "__pycache__" not in d for f in fs]
        assert sum(f.endswith(".pyc") for f in staged) == 6
        own = {"__init__.py", "utils.py"}
        assert all(f.endswith(".pyc") or f in own for f in staged), staged


def test_staged_reference_reproduces_a_golden_step():
    """the staged package is the unmodified reference: forward -> backward -> update on a golden's batches gives the golden's
    scores, loss and tables bit for bit (same torch build, same op sequence)"""
    if not _staged():
        pytest.skip("no reference here and nothing staged")
    import torch as th
    from golden_util import load_golden
    from oracle import ref_baseline as RB
    from oracle.ref_stub import NegG, PosG
    z, case = load_golden("transe_l2_small")
    w = dict(model=case["model"], n_ent=case["n_ent"], n_rel=case["n_rel"], hidden=case["hidden"], gamma=case["gamma"],
             lr=case["lr"], de=case["de"], dr=case["dr"], adv=case["adv"], adv_temp=case["adv_temp"],
             reg_coef=case["reg_coef"], reg_norm=case["reg_norm"], B=case["B"], N=case["N"])
    args = RB.make_args(w, 1)
    model = RB.make_model(w, args, seed=case["seed"])
    assert np.array_equal(model.entity_emb.emb.numpy(), z["init_entity"])
    for s in range(1, case["steps"] + 1):
        p = "s%d_" % s
        pos_g = PosG(th.from_numpy(z[p + "nid"]), th.from_numpy(z[p + "h_local"]), th.from_numpy(z[p + "t_local"]),
                     th.from_numpy(z[p + "r"]))
        neg_g = NegG(th.from_numpy(z[p + "neg"]), case["B"] // case["chunk"], case["chunk"], case["N"], bool(z[p + "neg_head"]))
        loss, log = model.forward(pos_g, neg_g, -1)
        loss.backward()
        assert np.array_equal(pos_g.edata["score"].detach().numpy(), z[p + "pos_score"])
        assert float(loss.item()) == float(z[p + "loss_total"])
        model.update(-1)
    assert np.array_equal(model.entity_emb.emb.numpy(), z["final_entity"])
    assert np.array_equal(model.relation_emb.emb.numpy(), z["final_relation"])


@pytest.mark.timeout(300)
def test_reference_baseline_harness_runs_the_reference_train_loop():
    if not _staged():
        pytest.skip("no reference here and nothing staged")
    from oracle import kge_oracle as O, ref_baseline as RB
    w = dict(model="TransE_l2", n_ent=500, n_rel=20, hidden=32, gamma=12.0, lr=0.1, de=False, dr=False, adv=True, adv_temp=1.0,
             reg_coef=1e-6, reg_norm=3, B=64, N=16)
    rng = np.random.RandomState(0)
    plans = [O.synth_batch(rng, w["n_ent"], w["n_rel"], w["B"], w["N"], w["N"], s) for s in range(1, 5)]
    args = RB.make_args(w, 6)
    model = RB.make_model(w, args)
    before = model.entity_emb.emb.clone()
    dt = RB.run_train(model, args, RB.graphs_of(plans, w))
    assert dt > 0 and not np.array_equal(before.numpy(), model.entity_emb.emb.numpy())
    assert float(model.entity_emb.state_sum.sum()) > 0
