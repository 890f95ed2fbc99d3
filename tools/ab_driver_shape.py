#!/usr/bin/env python3
"""tools/ab_driver_shape.py [R]: the driver's shape - [120 untimed steps][synchronise][20 timed steps][synchronise] - R times in ONE process,
for the launch-based sampler ('serial') and the sampler tail ('fused'): median / min of the 20-step wall and event times.  A single
run of `bench.py --steps 20` carries +-1 us/step of host jitter - too much to tell a 1-us difference."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dgl-ke_amd")):
    sys.path.insert(0, p)
import numpy as np
if os.environ.get("AB_SPIN"):      # experiment: spin-wait synchronise (hipDeviceScheduleSpin = 1, Yield = 2, BlockingSync = 4) set before the context exists
    import ctypes
    _hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(%s) ->" % os.environ["AB_SPIN"], _hip.hipSetDeviceFlags(ctypes.c_uint(int(os.environ["AB_SPIN"]))), flush=True)
import torch
import bench
from dglke_amd.dataloader import DeviceSampler, PrefetchedGroups
from dglke_amd.engine import StepEngine

R = int(sys.argv[1]) if len(sys.argv) > 1 else 30
w = dict(bench.WORKLOADS["transe_l2_fb15k"])
dev = torch.device("cuda", 0)
h, r, t = bench.synth_triples(w, 0)
G = 120
HEAD = [int(x) for x in os.environ.get("AB_HEADS", "0").split(",")]
for mode, head in [(m, hd) for hd in HEAD for m in ("serial", "fused")] * 2:
    torch.manual_seed(0)
    eng = StepEngine(w["model"], w["n_ent"], w["n_rel"], w["hidden"], w["gamma"], w["lr"], dev, w["de"], w["dr"], w["adv"], w["adv_temp"],
                     w["reg_coef"], w["reg_norm"])
    smp = DeviceSampler(h, r, t, w["n_ent"], w["B"], w["N"], dev, n_slots=2 * G, seed=0)
    eng.workspace_for(smp.sample(1)[0])
    pg = PrefetchedGroups(smp, eng.step, group_max=G, mode=mode, head=head)
    seq = [120, 20, 20] * (R + 2)
    pg.buf, pg.ready = 0, None
    pg.prefill(seq[0])
    walls, evs = [], []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(len(seq) - 1):
        timed = seq[i] == 20 and seq[i - 1] == 120 and i > 3          # (the first rounds capture the graphs)
        if timed:
            torch.cuda.synchronize()
            ev0.record()
            t0 = time.perf_counter()
        pg.run(seq[i + 1], graph=True)
        if timed:
            ev1.record()
            torch.cuda.synchronize()
            walls.append((time.perf_counter() - t0) * 1e6 / 20)
            evs.append(ev0.elapsed_time(ev1) * 1e3 / 20)
    torch.cuda.synchronize()
    walls, evs = np.array(walls), np.array(evs)
    print("%-6s head=%d  wall us/step: median %.2f  min %.2f  p90 %.2f   events: median %.2f  min %.2f   (%d timed 20-step groups)"
          % (mode, head, np.median(walls), walls.min(), np.percentile(walls, 90), np.median(evs), evs.min(), len(walls)), flush=True)
    del pg, smp, eng
