"""where do the LDS-slab backward and the direct backward differ? (developer probe)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dgl-ke_amd")]
import numpy as np, torch
from oracle import kge_oracle as O
from dglke_amd import plan
from dglke_amd.engine import StepEngine
DEV = "cuda:0"
model, n_ent, n_rel, hidden, B, N, chunk = "TransE_l2", 5000, 50, 64, 96, 250, 48
res = {}
for flags in (0, 1024):
    rng = np.random.RandomState(1234)
    cfg = O.Config(model, 12.0, hidden, 0.1, adv=True, adv_temp=1.0, reg_coef=1e-6, reg_norm=3)
    ent = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(n_ent, hidden)).astype(np.float32)
    rel = rng.uniform(-cfg.emb_init, cfg.emb_init, size=(n_rel, hidden)).astype(np.float32)
    eng = StepEngine(model, n_ent, n_rel, hidden, 12.0, 0.1, DEV, False, False, True, 1.0, 1e-6, 3, flags=flags)
    eng.load_tables(ent, rel)
    bt = O.synth_batch(rng, n_ent, n_rel, B, N, chunk, 1)
    b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], chunk, N, bt["neg_head"], DEV)
    want = eng.alloc_outputs(b)
    eng.step(b, want)
    torch.cuda.synchronize()
    res[flags] = {k: v.cpu().numpy() for k, v in want.items()}
for k in ("g_neg", "g_rel", "g_pos_ent"):
    a, b_ = res[0][k], res[1024][k]
    bad = np.argwhere(a != b_)
    print(k, a.shape, "differing elements:", len(bad))
    if len(bad):
        rows = np.unique(bad[:, 0]); cols = np.unique(bad[:, 1])
        print("  rows", rows[:40], "... cols", cols[:70])
        r, c = bad[0]
        print("  first", r, c, a[r, c], b_[r, c])
