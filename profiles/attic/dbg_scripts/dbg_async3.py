import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/dgl-ke_amd"); sys.path.insert(0, "/root/repo/tests")
from oracle import kge_oracle as O
import test_gpu_async as T
case = ("TransE_l2", 14951, 1345, 400, False, False, 200, 5, 200, 19.9, 0.25, 1e-9, 0)
cfg, ent, rel, bts, batches, engine = T._setup(case, seed=11, steps=2)
e64, r64 = ent.astype(np.float64), rel.astype(np.float64); es, rs = np.zeros(len(ent)), np.zeros(len(rel))
outs = O.train_steps_async(cfg, e64, es, r64, rs, bts)
s64, sr64 = ent.astype(np.float64), rel.astype(np.float64); ses, srs = np.zeros(len(ent)), np.zeros(len(rel))
souts = [O.train_step(cfg, s64, ses, sr64, srs, bt["nid"], bt["h_local"], bt["t_local"], bt["r"], bt["neg"], bt["neg_head"], bt["chunk"], bt["N"]) for bt in bts]
eng = engine()
eng.step_async(batches[0])
want = eng.alloc_outputs(batches[1])
eng.step_async(batches[1], want)
eng.flush_async(); torch.cuda.synchronize()
ps = want["pos_score"].cpu().numpy()
print("step2 pos_score: |gpu - stale oracle| max", np.abs(ps - outs[1]["pos_score"]).max(), " |gpu - strict oracle| max", np.abs(ps - souts[1]["pos_score"]).max())
ns = want["neg_score"].cpu().numpy()
print("step2 neg_score: |gpu - stale oracle| max", np.abs(ns - outs[1]["neg_score"]).max(), " |gpu - strict oracle| max", np.abs(ns - souts[1]["neg_score"]).max())
b = batches[1]
sel = np.searchsorted(b.p["ue_id"], bts[1]["nid"])
gp = want["g_pos_ent"].cpu().numpy()[sel]
ref = outs[1]["g_pos_ent"]
err = np.abs(gp - ref).max(1)
print("step2 g_pos_ent err max", err.max(), "ref max", np.abs(ref).max(), "rows with rel err > 1e-2:", (err > 1e-2 * np.abs(ref).max(1)).sum())
g = eng.ent.cpu().numpy(); d = np.abs(g - e64).max(1); bad = np.where(d > 2.5e-3)[0]
# for bad rows: compare the gradient row
pos = {int(n): k for k, n in enumerate(bts[1]["nid"])}
for r_ in bad[:5]:
    k = pos[int(r_)]
    print(" row", r_, "gpu g max", np.abs(gp[k]).max(), "oracle g max", np.abs(ref[k]).max(), "ratio", np.abs(gp[k]).max() / np.abs(ref[k]).max(),
          "state gpu", eng.ent_state[int(r_)].item(), "state oracle", es[r_])
