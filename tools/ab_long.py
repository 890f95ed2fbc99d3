#!/usr/bin/env python3
"""tools/ab_long.py [G] [groups] [rounds]: back-to-back groups of G steps, launch-based sampler vs sampler tail on every step, alternating
in ONE process: us/step of `groups` groups, `rounds` times each (median)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dgl-ke_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench
from dglke_amd.dataloader import DeviceSampler, PrefetchedGroups
from dglke_amd.engine import StepEngine

G = int(sys.argv[1]) if len(sys.argv) > 1 else 120
NG = int(sys.argv[2]) if len(sys.argv) > 2 else 10
R = int(sys.argv[3]) if len(sys.argv) > 3 else 7
w = dict(bench.WORKLOADS["transe_l2_fb15k"])
dev = torch.device("cuda", 0)
h, r, t = bench.synth_triples(w, 0)
runs = {}
for mode in ("serial", "fused"):
    torch.manual_seed(0)
    eng = StepEngine(w["model"], w["n_ent"], w["n_rel"], w["hidden"], w["gamma"], w["lr"], dev, w["de"], w["dr"], w["adv"], w["adv_temp"],
                     w["reg_coef"], w["reg_norm"])
    smp = DeviceSampler(h, r, t, w["n_ent"], w["B"], w["N"], dev, n_slots=2 * G, seed=0)
    eng.workspace_for(smp.sample(1)[0])
    pg = PrefetchedGroups(smp, eng.step, group_max=G, mode=mode, fused_max=100000)
    pg.buf, pg.ready = 0, None
    pg.prefill(G)
    for _ in range(4):
        pg.run(G, graph=True)
    torch.cuda.synchronize()
    runs[mode] = (eng, smp, pg)
res = {"serial": [], "fused": []}
for rep in range(R):
    for mode in ("serial", "fused"):
        pg = runs[mode][2]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(NG):
            pg.run(G, graph=True)
        torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) * 1e6 / (NG * G))
for mode in ("serial", "fused"):
    a = np.array(res[mode])
    print("G=%d %-6s us/step: median %.3f  min %.3f  max %.3f" % (G, mode, np.median(a), a.min(), a.max()), flush=True)
