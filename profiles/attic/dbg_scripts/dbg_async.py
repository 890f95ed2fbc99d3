import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/dgl-ke_amd"); sys.path.insert(0, "/root/repo/tests")
from oracle import kge_oracle as O
import test_gpu_async as T
case = ("TransE_l2", 14951, 1345, 400, False, False, 200, 5, 200, 19.9, 0.25, 1e-9, 0)
for steps in (2, 3, 6):
    cfg, ent, rel, bts, batches, engine = T._setup(case, seed=11, steps=steps)
    # strict control
    s64, sr64 = ent.astype(np.float64), rel.astype(np.float64); ses, srs = np.zeros(len(ent)), np.zeros(len(rel))
    for bt in bts:
        O.train_step(cfg, s64, ses, sr64, srs, bt["nid"], bt["h_local"], bt["t_local"], bt["r"], bt["neg"], bt["neg_head"], bt["chunk"], bt["N"])
    eng = engine()
    for b in batches: eng.step(b)
    torch.cuda.synchronize()
    d = np.abs(eng.ent.cpu().numpy() - s64).max(1)
    print("steps", steps, "STRICT max err", d.max(), "rows>2.5e-3:", (d > 2.5e-3).sum())
    e64, r64 = ent.astype(np.float64), rel.astype(np.float64); es, rs = np.zeros(len(ent)), np.zeros(len(rel))
    O.train_steps_async(cfg, e64, es, r64, rs, bts)
    eng = engine()
    for b in batches: eng.step_async(b)
    eng.flush_async(); torch.cuda.synchronize()
    d = np.abs(eng.ent.cpu().numpy() - e64).max(1)
    bad = np.where(d > 2.5e-3)[0]
    print("steps", steps, "ASYNC  max err", d.max(), "rows>2.5e-3:", len(bad), "state err", np.abs(eng.ent_state.cpu().numpy()-es).max())
    # classify the bad rows: in how many steps were they touched, consecutive?
    touched = [set(np.concatenate([bt["h"], bt["t"], bt["neg"]]).tolist()) for bt in bts]
    cnt = {}
    for r_ in bad[:2000]:
        key = tuple(int(r_ in t) for t in touched)
        cnt[key] = cnt.get(key, 0) + 1
    print("   bad rows by touch pattern:", sorted(cnt.items(), key=lambda kv: -kv[1])[:8])
    # compare against strict oracle too
    d2 = np.abs(eng.ent.cpu().numpy() - s64).max(1)
    print("   async GPU vs STRICT oracle: rows>2.5e-3:", (d2 > 2.5e-3).sum())
