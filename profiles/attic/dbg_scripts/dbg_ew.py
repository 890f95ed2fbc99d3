import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dgl-ke_amd")]
from dglke_amd import plan
from dglke_amd.engine import StepEngine
from oracle import kge_oracle as O
dev = "cuda:0"
import itertools
for (model, (n_ent, n_rel, hidden, B, N)) in itertools.product(("SimplE", "ComplEx"), ((500, 20, 64, 64, 16), (3000, 40, 100, 256, 64), (3000, 40, 200, 200, 200))):
    rng = np.random.RandomState(0)
    bt = O.synth_batch(rng, n_ent, n_rel, B, N, N, 1)
    outs = []
    cx = model in ("ComplEx", "SimplE")
    for flags in (0, 2):
        torch.manual_seed(0)
        eng = StepEngine(model, n_ent, n_rel, hidden, 12.0, 0.1, dev, cx, cx, True, 1.0, 1e-6, 3, flags=flags)
        b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"], dev)
        want = eng.alloc_outputs(b)
        eng.step(b, want)
        torch.cuda.synchronize()
        outs.append({k: v.cpu().numpy() for k, v in want.items()})
    for k in ("g_pos_ent", "g_rel", "g_neg"):
        a, c = outs[0][k], outs[1][k]
        bad = ~np.isclose(a, c, rtol=1e-4, atol=1e-6)
        print(model, hidden, B, N, k, a.shape, "bad", bad.sum())
        if bad.any():
            rows, cols = np.nonzero(bad)
            print("   rows", np.unique(rows)[:20], "cols", np.unique(cols)[:40])
            print("   sample", a[rows[0], cols[0]], c[rows[0], cols[0]])
