#!/usr/bin/env python3
"""tools/sampler_timeline.py [n_slots]: wavefront timeline of sample_plan_kernel (library built with -DKGE_TIMELINE
[-DKGE_TL_MARKS], KGE_LIB pointing at it): per workgroup kind the wave life and the phase marks."""
import ctypes as C
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dgl-ke_amd")]
import bench
from dglke_amd import _lib
from dglke_amd.dataloader import DeviceSampler
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
w = dict(bench.WORKLOADS["transe_l2_fb15k"])
dev = torch.device("cuda", 0)
h, r, t = bench.synth_triples(w, 0)
lib = _lib.lib()
fn = lib.kge_tl_set_sampler; fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
buf = torch.zeros(8 * 8192 * 8, dtype=torch.int64, device=dev)
assert fn(buf.data_ptr()) == 0
smp = DeviceSampler(h, r, t, w["n_ent"], w["B"], w["N"], dev, n_slots=max(n, 2), seed=0)
for _ in range(3):
    smp.sample(n)
torch.cuda.synchronize()
buf.zero_()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); smp.sample(n); ev1.record(); torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 8); a = a[a[:, 3] > 0]
print("launch of %d slots: %.1f us (events)" % (n, 1e3 * ev0.elapsed_time(ev1)))
for kid, name in ((0, "entity plan"), (1, "relation plan")):
    m = a[:, 3] - 1 == kid
    t0, t1 = a[m, 1] * 0.01, a[m, 2] * 0.01
    base = (a[:, 1] * 0.01).min()
    print("%-14s waves %4d  start %.2f..%.2f  end %.2f..%.2f  life p50 %.2f max %.2f" % (
        name, m.sum(), (t0 - base).min(), (t0 - base).max(), (t1 - base).min(), (t1 - base).max(),
        np.percentile(t1 - t0, 50), (t1 - t0).max()))
    mk = a[m][:, 4:8] * 0.01
    for j in range(4):
        ok = mk[:, j] > 0
        if ok.any():
            print("    mark%d since wave start p50 %.2f  max %.2f" % (j, np.percentile(mk[ok, j] - t0[ok], 50), (mk[ok, j] - t0[ok]).max()))
