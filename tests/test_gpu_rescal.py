"""RESCAL on the HIP path (kge_rescal.hip) beyond the goldens' toy widths: the fused step against the fp64 oracle at widths that
exercise every instance of the per-unique-relation passes (one / two / four 16-byte column chunks per lane), relations carried by
many edges of the batch (register group + the edges before it), the regulariser, and the per-edge fallback (width not a
multiple of 4).  Reference: models/pytorch/score_fun.py:378-449 (RESCALScore), tensor_models.py:330-361 (sparse Adagrad)."""
import numpy as np
import pytest
import torch

from oracle import kge_oracle as O
from test_gpu_parity import DEV, _close, grad_tol

pytestmark = pytest.mark.gpu

CASES = [
    # n_ent, n_rel, hidden, B, chunk, N, reg, adv
    (2000, 300, 500, 256, 64, 64, 0.0, True),      # the FB15k recipe's width (two chunks per lane), mostly one edge per relation
    (300, 3, 36, 96, 32, 16, 1e-4, True),          # ~32 edges per relation: the edges before the register group, regulariser
    (300, 7, 260, 64, 32, 32, 0.0, False),         # just past one chunk per lane
    (200, 5, 516, 32, 16, 16, 1e-5, True),         # four chunks per lane (D > 512)
    (200, 5, 18, 48, 16, 8, 1e-4, True),           # D % 4 != 0: the per-edge passes
    (200, 40, 16, 40, 8, 4, 0.0, True),            # fewer rows than row blocks
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "D%d-rel%d-B%d" % (c[2], c[1], c[3]))
def test_rescal_fused_step_matches_oracle(case):
    from dglke_amd import plan
    from dglke_amd.engine import StepEngine
    n_ent, n_rel, hidden, B, chunk, N, reg, adv = case
    lr, gamma = 0.05, 6.0
    cfg = O.Config("RESCAL", gamma, hidden, lr, adv=adv, adv_temp=1.0, reg_coef=reg, reg_norm=3)
    rng = np.random.RandomState(hidden * 7 + n_rel)
    ent = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(n_ent, cfg.ent_dim)).astype(np.float32)
    rel = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(n_rel, hidden * hidden)).astype(np.float32)   # general_models.py:232-236
    eng = StepEngine("RESCAL", n_ent, n_rel, hidden, gamma, lr, DEV, False, False, adv, 1.0, reg, 3)
    assert eng.rel.shape == (n_rel, hidden * hidden)
    eng.load_tables(ent, rel)
    for step in (1, 2, 3):                          # tail, head, tail corruption
        ent64, rel64 = eng.ent.cpu().numpy().astype(np.float64), eng.rel.cpu().numpy().astype(np.float64)
        es64, rs64 = eng.ent_state.cpu().numpy().astype(np.float64), eng.rel_state.cpu().numpy().astype(np.float64)
        bt = O.synth_batch(rng, n_ent, n_rel, B, N, chunk, step)
        b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], chunk, N, bt["neg_head"], DEV)
        want = eng.alloc_outputs(b)
        eng.step(b, want)
        torch.cuda.synchronize()
        out = O.train_step(cfg, ent64, es64, rel64, rs64, bt["nid"], bt["h_local"], bt["t_local"], bt["r"], bt["neg"],
                           bt["neg_head"], chunk, N)
        tag = "RESCAL %r step %d" % (case, step)
        _close(want["pos_score"].cpu(), out["pos_score"], 1e-4, 1e-4, tag + " pos_score")
        _close(want["neg_score"].cpu(), out["neg_score"], 1e-4, 1e-4, tag + " neg_score")
        l4 = eng.read_loss()
        _close(l4[:3], out["log"][:3], 1e-4, 1e-5, tag + " loss")
        _close(l4[3], out["log"][3], 1e-3, 1e-7, tag + " reg")
        sel = np.searchsorted(b.p["ue_id"], bt["nid"])
        _close(want["g_pos_ent"].cpu().numpy()[sel], out["g_pos_ent"], 3e-4, grad_tol(out["g_pos_ent"]), tag + " g_pos_ent")
        _close(want["g_neg"].cpu(), out["g_neg"], 3e-4, grad_tol(out["g_neg"]), tag + " g_neg")
        _close(want["g_rel"].cpu(), out["g_rel"], 3e-4, grad_tol(out["g_rel"]), tag + " g_rel")
        _close(eng.ent_state.cpu(), es64, 2e-3, 1e-9, tag + " ent state")
        _close(eng.rel_state.cpu(), rs64, 2e-3, 1e-9, tag + " rel state")
        _close(eng.ent.cpu(), ent64, 1e-4, 1e-3 * lr, tag + " entity rows")
        _close(eng.rel.cpu(), rel64, 1e-4, 1e-3 * lr, tag + " relation rows")
