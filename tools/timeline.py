#!/usr/bin/env python3
"""tools/timeline.py - per-wavefront timeline of ONE strict training step (developer tool).

Needs the library variant built with -DKGE_TIMELINE (tools/build_variant.sh tl -DKGE_TIMELINE) selected through
KGE_LIB.  Every wavefront of the step's kernels writes {hardware id, start, end} on the 100 MHz wall clock; the
buffers are read after a hipGraph replay of G steps (later steps overwrite earlier ones, so the records show the
LAST step of the group) and summarised: when each kernel's first / median / last wavefront started and ended,
how long the wavefronts lived, and the gaps between the kernels.

    KGE_LIB=dgl-ke_amd/variants/libkge_tl.so python tools/timeline.py [--workload transe_l2_fb15k] [--flags F]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dgl-ke_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch

NAMES = {0: "edge_fwd", 1: "neg_fwd_gemm", 2: "loss", 3: "neg_bwd_gemm", 4: "update(ent)", 5: "edge_bwd", 6: "aux(gn_red/smp_tail)", 7: "update(rel)"}
PER = 8192
NK = 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="transe_l2_fb15k")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--graph-steps", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--async-update", action="store_true")
    ap.add_argument("--skew", action="store_true", help="heavy-tailed ids (bench.py --skew)")
    ap.add_argument("--tail", action="store_true", help="every step carries a sampler job (kge_step_fused_sampling): the tail workgroups are kernel id 6")
    ap.add_argument("--per-cu", type=int, default=-1, help="kernel id: wavefronts per CU / SIMD and their life by occupancy class")
    args = ap.parse_args()
    import bench
    from dglke_amd import _lib
    _kge_lib = _lib
    from dglke_amd.dataloader import DeviceSampler
    from dglke_amd.engine import StepEngine
    w = dict(bench.WORKLOADS[args.workload])
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    h, r, t = bench.synth_triples(w, 0, args.skew)
    eng = StepEngine(w["model"], w["n_ent"], w["n_rel"], w["hidden"], w["gamma"], w["lr"], dev, w["de"], w["dr"],
                     w["adv"], w["adv_temp"], w["reg_coef"], w["reg_norm"], flags=args.flags)
    lib = _lib.lib()
    bufs = {}
    for tu in ("rowwise", "gemm", "bcast"):
        fn = getattr(lib, "kge_tl_set_" + tu, None)
        if fn is None:
            raise SystemExit("library has no timeline hooks: build with -DKGE_TIMELINE and set KGE_LIB")
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p]
        bufs[tu] = torch.zeros(NK * PER * 8, dtype=torch.int64, device=dev)
        assert fn(bufs[tu].data_ptr()) == 0
    G = args.graph_steps
    smp = DeviceSampler(h, r, t, w["n_ent"], w["B"], w["N"], dev, n_slots=2 * G if args.tail else G, seed=0)
    cur = smp.sample(G) if args.tail else None
    def run_group():
        if args.tail:
            jobs, _ = smp.tail_jobs(G, slot0=G)
            for k, b in enumerate(cur):
                eng.step(b, sample_job=jobs[k])
        elif args.async_update:
            eng.steps_async(smp.sample())
        else:
            for b in smp.sample():
                eng.step(b)
    run_group()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with _kge_lib.graph_capture(g):
        run_group()
    for _ in range(args.reps):
        g.replay()
    torch.cuda.synchronize()
    rec = []
    for tu, bf in bufs.items():
        a = bf.cpu().numpy().reshape(NK * PER, 8)
        a = a[a[:, 3] > 0]
        rec.append(a)
    a = np.concatenate(rec, 0)
    kid = a[:, 3] - 1
    t0 = a[:, 1].astype(np.float64) * 0.01      # us
    t1 = a[:, 2].astype(np.float64) * 0.01
    xcc = (a[:, 0] >> 32) & 0xf
    order = sorted(set(kid.tolist()), key=lambda k: t0[kid == k].min())
    # only the records of the LAST step: a kernel that did not run in the last step keeps stale records - drop
    # everything that ended before the first kernel's earliest start
    base = t0[kid == order[0]].min() if order else 0.0
    print("# %s flags=%d  (times in us relative to the first wavefront of the step's first kernel; 10 ns clock)" % (args.workload, args.flags))
    print("%-14s %6s | %7s %7s %7s | %7s %7s %7s | %6s %6s %6s | waves per XCD" % (
        "kernel", "waves", "start0", "start50", "start99", "end1", "end50", "end100", "dur50", "dur90", "durmax"))
    prev_end = None
    for k in order:
        m = kid == k
        s, e = t0[m] - base, t1[m] - base
        d = e - s
        per = np.bincount(xcc[m].astype(int), minlength=8)
        gap = "" if prev_end is None else "  gap_after_prev_end %.2f" % (s.min() - prev_end)
        print("%-14s %6d | %7.2f %7.2f %7.2f | %7.2f %7.2f %7.2f | %6.2f %6.2f %6.2f | %s%s" % (
            NAMES.get(int(k), str(k)), m.sum(), s.min(), np.percentile(s, 50), np.percentile(s, 99),
            np.percentile(e, 1), np.percentile(e, 50), e.max(), np.percentile(d, 50), np.percentile(d, 90), d.max(),
            " ".join(str(x) for x in per), gap))
        prev_end = e.max()
        mk = a[m][:, 4:8].astype(np.float64) * 0.01
        if (mk > 0).any():          # phase marks (KGE_TL_MARKS build): median time of each mark since the wave start
            parts = []
            for j in range(4):
                ok = mk[:, j] > 0
                if ok.any():
                    rel = mk[ok, j] - t0[m][ok]
                    parts.append("mark%d p50 %.2f p90 %.2f" % (j, np.percentile(rel, 50), np.percentile(rel, 90)))
            print("               marks since wave start: " + " | ".join(parts))
    if order:
        print("step span (first start -> last end): %.2f us" % (t1.max() - base))
    if args.per_cu >= 0:
        # HW_ID (gfx9 layout): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
        m = kid == args.per_cu
        hw = a[m][:, 0] & 0xffffffff
        simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
        cuid = ((xcc[m] * 8 + se) * 2 + sh) * 16 + cu
        sid = cuid * 4 + simd
        d = (t1 - t0)[m]
        e = (t1 - base)[m]
        ncu, nsimd = len(set(cuid.tolist())), len(set(sid.tolist()))
        print("# kernel %d: %d wavefronts on %d CUs / %d SIMDs" % (args.per_cu, m.sum(), ncu, nsimd))
        cnt = {}
        for x in sid.tolist():
            cnt[x] = cnt.get(x, 0) + 1
        occ = np.array([cnt[x] for x in sid.tolist()])
        for c in sorted(set(occ.tolist())):
            mm = occ == c
            print("#   SIMDs holding %d wavefronts: %4d SIMDs, wavefront life p50 %.2f max %.2f, last end %.2f us" % (
                c, len(set(sid[mm].tolist())), np.percentile(d[mm], 50), d[mm].max(), e[mm].max()))


if __name__ == "__main__":
    main()
