import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/dgl-ke_amd"); sys.path.insert(0, "/root/repo/tests")
from oracle import kge_oracle as O
import test_gpu_async as T
for flags in (0, 64, 2, 4):
    case = ("TransE_l2", 14951, 1345, 400, False, False, 200, 5, 200, 19.9, 0.25, 1e-9, flags)
    cfg, ent, rel, bts, batches, engine = T._setup(case, seed=11, steps=2)
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64); es, rs = np.zeros(len(ent)), np.zeros(len(rel))
    outs = O.train_steps_async(cfg, e64, es, r64, rs, bts, defer_rel=bool(flags & 64))
    eng = engine()
    for b in batches: eng.step_async(b)
    eng.flush_async(); torch.cuda.synchronize()
    g = eng.ent.cpu().numpy()
    d = np.abs(g - e64).max(1)
    bad = np.where(d > 2.5e-3)[0]
    print("flags", flags, "bad rows", len(bad), "max", d.max(), "rel err", np.abs(eng.rel.cpu().numpy() - r64).max())
    if flags == 0 and len(bad):
        b1, b2 = bts
        for r_ in bad[:6]:
            roles = []
            for k, bt in enumerate(bts):
                roles.append(("h%d" % (bt["h"] == r_).sum(), "t%d" % (bt["t"] == r_).sum(), "n%d" % (bt["neg"] == r_).sum()))
            # candidates
            x0 = ent[r_].astype(np.float64)
            print("  row", r_, roles, "|gpu-oracle|", d[r_], "|gpu-x0|", np.abs(g[r_] - x0).max(), "|oracle-x0|", np.abs(e64[r_] - x0).max())
        # aggregate roles
        import collections
        c = collections.Counter()
        for r_ in bad:
            c[tuple(((bt["h"] == r_).any() or (bt["t"] == r_).any(), (bt["neg"] == r_).any()) for bt in bts)] += 1
        print("  roles (pos, neg) per step:", c.most_common())
