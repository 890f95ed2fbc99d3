#!/usr/bin/env python3
"""bench.py - positive edges/sec of the fused KGE training step on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` (for N>1 launched under
torch.distributed.run, one rank per GPU).  W untimed warm-up steps, then EXACTLY K timed steps
bracketed by barrier + synchronize, max over ranks, ONE JSON line on rank 0.

A "step" = one pass of the hot path over one batch: gather -> positive score -> chunked negative
score -> loss -> analytic gradients -> row-sparse Adagrad update (reference:
train_pytorch.py:141-152), on synthetic FB15k-shaped triples whose id batches (and their
duplicate-grouping plan) are pre-staged in HBM before the timed region.  Workload at N=1 =
BASELINE.json configs[1]: TransE_l2, n_ent 14951, n_rel 1345, batch 1000, neg 200, dim 400,
-adv, lr 0.25, regularization_coef 1e-9, gamma 19.9.

Extra objects on the JSON line:
  roofline     - algorithmic HBM bytes of one step (12*(R_e*D_e+R_r*D_r)+16*(R_e+R_r))
                 divided by the average step duration measured with HIP events on the launch stream
  cpu_baseline - the CPU oracle port (oracle/torch_port.py, the reference's torch ops on the host
                 cores) timed on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dgl-ke_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

from dglke_amd import _lib as _kge_lib

WORKLOADS = {
    # BASELINE.json configs[1]
    "transe_l2_fb15k": dict(model="TransE_l2", n_ent=14951, n_rel=1345, n_train=483142, hidden=400,
                            de=False, dr=False, B=1000, N=200, gamma=19.9, lr=0.25, adv=True,
                            adv_temp=1.0, reg_coef=1e-9, reg_norm=3),
    # configs[2]
    "distmult_fb15k": dict(model="DistMult", n_ent=14951, n_rel=1345, n_train=483142, hidden=400,
                           de=False, dr=False, B=1000, N=200, gamma=143.0, lr=0.08, adv=True,
                           adv_temp=1.0, reg_coef=2e-6, reg_norm=3),
    # configs[3] (row = 2*hidden floats with -de -dr)
    "complex_wikikg2": dict(model="ComplEx", n_ent=2500604, n_rel=535, n_train=16109182, hidden=200,
                            de=True, dr=True, B=1024, N=256, gamma=143.0, lr=0.1, adv=True,
                            adv_temp=1.0, reg_coef=2e-6, reg_norm=3),
    "rotate_fb15k": dict(model="RotatE", n_ent=14951, n_rel=1345, n_train=483142, hidden=200,
                         de=True, dr=False, B=1024, N=256, gamma=12.0, lr=0.009, adv=True,
                         adv_temp=1.0, reg_coef=1e-7, reg_norm=3),
    # the reference's FB15k recipes for the remaining score functions (examples/fb15k/multi_gpu.sh)
    "simple_fb15k": dict(model="SimplE", n_ent=14951, n_rel=1345, n_train=483142, hidden=400,
                         de=True, dr=True, B=1000, N=200, gamma=143.0, lr=0.1, adv=True,
                         adv_temp=1.0, reg_coef=2e-6, reg_norm=3),
    "rescal_fb15k": dict(model="RESCAL", n_ent=14951, n_rel=1345, n_train=483142, hidden=500,
                         de=False, dr=False, B=1024, N=256, gamma=24.0, lr=0.03, adv=True,
                         adv_temp=1.0, reg_coef=0.0, reg_norm=3),
    "transr_fb15k": dict(model="TransR", n_ent=14951, n_rel=1345, n_train=483142, hidden=200,
                         de=False, dr=False, B=1024, N=256, gamma=8.0, lr=0.015, adv=True,
                         adv_temp=1.0, reg_coef=5e-8, reg_norm=3),
    # cfg-R's per-GPU step on ONE local table (BASELINE configs[4] without its exchange: D_e = 800 / D_r = 400, batch 1024, neg 256, a
    # 1 M-row shard stand-in = 3.2 GB): the shape the pairwise kernels are profiled at (tools/timeline.py, rocprofv3)
    "rotate_wide": dict(model="RotatE", n_ent=1000003, n_rel=14824, n_train=2000000, hidden=400,
                        de=True, dr=False, B=1024, N=256, gamma=12.0, lr=0.01, adv=True,
                        adv_temp=1.0, reg_coef=1e-7, reg_norm=3),
    "transe_l1_fb15k": dict(model="TransE_l1", n_ent=14951, n_rel=1345, n_train=483142, hidden=400,
                            de=False, dr=False, B=1000, N=200, gamma=16.0, lr=0.01, adv=True,
                            adv_temp=1.0, reg_coef=1e-7, reg_norm=3),
}


def algorithmic_bytes(plans, d_e, d_r):
    """every traced row is read once in forward and read+written once in the
    update (12 B per float), plus 4 B state read + 4 B state write + 8 B id per traced row."""
    tot = 0.0
    for p in plans:
        r_e = p["U"] + p["C"] * p["N"]
        r_r = p["B"]
        tot += 12.0 * (r_e * d_e + r_r * d_r) + 16.0 * (r_e + r_r)
    return tot / len(plans)


def measured_traffic(workload):
    """HBM bytes per step from the rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
    separate passes, tools/pmc_cycle.sh), recorded in profiles/latest_traffic.json for the build that
    produced it; bench.py cannot collect PMC counters itself."""
    for name in ("latest_traffic_%s.json" % workload, "latest_traffic.json"):      # (per-workload files: the other BASELINE configs' legs)
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            if d.get("workload") == workload:
                _PROFILE_HASHES["traffic"] = d.get("source_hash")
                return d["hbm_bytes_per_step"]
        except Exception:
            pass
    return None


_PROFILE_HASHES = {}      # source hashes of the committed profile summaries this line quotes (measured_traffic, profiled_kernels)


def profile_matches_build():
    """True when every committed profile summary quoted in the line (roofline.traffic, dominant_kernel_us, profiled_step_us) was
    taken from the source set the running libkge_hip.so was built from (__graft_entry__.source_hash, written into
    profiles/latest_*.json by tools/kernel_stats_json.py / traffic_json.py); False when one of them is older or carries no hash."""
    import __graft_entry__
    cur = __graft_entry__.source_hash()
    return {"matches": bool(_PROFILE_HASHES) and all(v == cur for v in _PROFILE_HASHES.values()), "build_source_hash": cur,
            "profile_source_hashes": dict(_PROFILE_HASHES)}


def profiled_kernels(workload):
    """per-kernel average launch durations of the committed rocprofv3 --kernel-trace --stats summary of this workload
    (profiles/latest_kernel_stats.json, written by tools/kernel_stats_json.py); None when the profile is of another one."""
    try:
        with open(os.path.join(ROOT, "profiles", "latest_kernel_stats.json")) as f:
            d = json.load(f)
        if d.get("workload") != workload:
            return None
        _PROFILE_HASHES["kernel_stats"] = d.get("source_hash")
        return d
    except Exception:
        return None


def synth_triples(w, seed, skew=False):
    """FB15k-shaped synthetic triples: h,t ~ U[0,n_ent), r ~ U[0,n_rel).
    skew=True: heavy-tailed ids instead (entity k with weight 1/(k+10)^0.9, relation k with 1/(k+5): the most frequent
    entity is ~1 % of the heads and of the tails, the most frequent relation ~3.6 % of the edges - FB15k's own proportions,
    /m/09c7w0 and award_nominee): the rows with long contribution lists that uniform ids never produce."""
    rng = np.random.RandomState(seed)
    n = min(w["n_train"], 2_000_000)
    if skew:
        pe = 1.0 / (np.arange(w["n_ent"]) + 10.0) ** 0.9
        pr = 1.0 / (np.arange(w["n_rel"]) + 5.0)
        perm_e, perm_r = rng.permutation(w["n_ent"]), rng.permutation(w["n_rel"])
        return (perm_e[rng.choice(w["n_ent"], n, p=pe / pe.sum())].astype(np.int64),
                perm_r[rng.choice(w["n_rel"], n, p=pr / pr.sum())].astype(np.int64),
                perm_e[rng.choice(w["n_ent"], n, p=pe / pe.sum())].astype(np.int64))
    return (rng.randint(0, w["n_ent"], n).astype(np.int64),
            rng.randint(0, w["n_rel"], n).astype(np.int64),
            rng.randint(0, w["n_ent"], n).astype(np.int64))


def step_kernels(model, force_pairwise=False):
    """the launches of one strict training step per model family and the dominant one."""
    if model == "TransR":
        return ("one training step = projections (matvec), transr_pos, transr_fwd_wide, loss, transr_gn_wide / gn_reduce / "
                "gp_wide (dq, GR and the gradient's sum of squares in its sweep), projection Adagrad, update; dominant: the three "
                "64 x 208-tile fp32-MFMA products (kge_transr_wide.hpp)")
    if model == "RESCAL":
        return ("one training step = rescal_rel_fwd (one pass over M per UNIQUE relation), neg_fwd_gemm, loss, neg_bwd_gemm, "
                "rescal_rel_bwd_apply (backward products + Adagrad in one pass per unique relation), combine, update; dominant: "
                "rescal_rel_bwd_apply / rescal_rel_fwd (HBM streaming of the relation matrices; the algorithmic bytes count every "
                "traced row, the kernels read a relation carried by several edges once)")
    if model == "TransE_l1" and not force_pairwise:
        return ("one training step = [neg_fwd_bcast tasks || edge-forward rows], loss, neg_bwd_lc, gn_reduce, update; "
                "dominant: neg_bwd_lc_kernel (VALU, packed fp32)")
    if model in ("TransE_l1", "RotatE") or force_pairwise:
        return ("one training step = edge_fwd, neg_fwd_bcast, loss, neg_bwd_lc, [edge_bwd || gn_reduce], "
                "update; dominant: neg_bwd_lc_kernel (VALU, packed fp32)")
    if model == "TransE_l2":
        return ("one training step = 4 dependent launches ([forward GEMM tiles || edge-forward rows], loss, neg_bwd_gemm, "
                "update); the first and the third take ~9 us each (roofline.dominant_kernel: the committed profile's)")
    if model in ("DistMult", "ComplEx"):
        return ("one training step = 4 dependent launches ([forward GEMM tiles || edge-forward rows], loss, neg_bwd_gemm with the "
                "per-edge gradient rows written by the GA tiles' epilogue, update); dominant: the first launch / neg_bwd_gemm_kernel")
    return ("one training step = 5 dependent kernels (edge_fwd, neg_fwd_gemm, loss, neg_bwd_gemm with the per-edge gradient rows "
            "written by the GA tiles' epilogue, update); dominant: neg_bwd_gemm_kernel")


def cpu_baseline(w, plans, budget_s=12.0, max_steps=200):
    """the CPU baseline on this host's cores: the REFERENCE ITSELF when oracle/make_ref.py staged its hot-path files into
    oracle/_ref at build() time (`kind: "reference"`: dglke.train_pytorch.train() on the unmodified files, DGL stubbed), else the
    torch-CPU port of the reference step (oracle/torch_port.py, `kind: "port"`)."""
    if len(plans) < 32:          # device-sampler mode keeps only a few host plans: make a host sample
        from dglke_amd.dataloader import UniformChunkedSampler
        hh, rr, tt = synth_triples(w, 0)
        plans = UniformChunkedSampler(hh, rr, tt, w["n_ent"], w["B"], w["N"], "cpu", seed=0).next_plans(max_steps + 15)
    try:
        from oracle import ref_baseline
        if os.environ.get("KGE_CPU_BASELINE", "reference") != "port" and ref_baseline.available():
            return cpu_baseline_reference(w, plans, budget_s)
    except Exception as e:       # noqa: BLE001 - the staged reference must never hide the number: fall back to the port, say why
        why = repr(e)
    else:
        why = "oracle/_ref not staged (build() ran without /root/reference)"
    out = cpu_baseline_port(w, plans, budget_s, max_steps)
    out["reference_unavailable"] = why
    return out


def cpu_baseline_reference(w, plans, budget_s=12.0):
    """`kind: "reference"`: the reference's own train() loop (train_pytorch.py:95-197) on its own KEModel, both ways the reference
    uses a many-core host; `value` is the better of the two."""
    from oracle import ref_baseline as RB
    nthreads = torch.get_num_threads()
    one = RB.single(w, plans, budget_s=budget_s)
    shape = "%s, B=%d N=%d D=%d" % (w["model"], w["B"], w["N"], w["hidden"])
    out = {"value": one["value"], "unit": "edges/s", "cores": one["threads"], "kind": "reference",
           "sample": "%d steps of the same workload (%s) by the reference's own train() / KEModel.forward / backward / update "
                     "(the six files of SURVEY 8(a), compiled unmodified into oracle/_ref; DGL stubbed: the CPU side runs "
                     "WITHOUT its sampler, the GPU `value` INCLUDES sampling and plan construction - conservative for the GPU); intra-op threads = the fastest of the probed counts on this %d-core host"
                     % (one["steps"], shape, nthreads),
           "ms_per_step": round(1e3 * w["B"] / one["value"], 3), "edges_per_s_by_threads": one["edges_per_s_by_threads"],
           "single_process": {"value": one["value"], "threads": one["threads"]}}
    try:
        procs = max(2, min(64, (os.cpu_count() or 2)))
        hv, hsteps, hwall = RB.hogwild(w, procs, seconds=5.0, rate_1=one["edges_per_s_by_threads"].get(1))
        out["num_proc"] = {"value": round(hv, 1), "procs": procs, "steps": hsteps, "seconds": round(hwall, 2),
                           "semantics": "P single-thread trainer processes on one shared-memory KEModel (reference --num_proc P)"}
        if hv > one["value"]:
            out["value"], out["cores"] = round(hv, 1), procs
            out["ms_per_step"] = round(1e3 * w["B"] / hv, 3)          # aggregate: one step of ANY process every ... ms
            out["sample"] = ("%d steps in %.1f s by %d single-thread processes sharing one KEModel (reference --num_proc %d; %s): "
                             "the reference's own train() / KEModel.forward / backward / update on the six files of SURVEY 8(a), "
                             "compiled unmodified into oracle/_ref (DGL stubbed: the CPU side runs WITHOUT its sampler, the GPU `value` "
                             "INCLUDES sampling and plan construction - conservative for the GPU); the "
                             "single-process run with %d intra-op threads: %.0f edges/s"
                             % (hsteps, hwall, procs, procs, shape, one["threads"], one["value"]))
    except Exception as e:  # noqa: BLE001 - the multi-process leg must never hide the single-process number
        out["num_proc"] = {"error": repr(e)}
    return out


def cpu_baseline_port(w, plans, budget_s=12.0, max_steps=200):
    """time the CPU port of the reference step (oracle/torch_port.py) on the host cores."""
    from oracle import torch_port
    th = torch
    nthreads = th.get_num_threads()
    model = torch_port.TorchPort(w["model"], w["n_ent"], w["n_rel"], w["hidden"], w["gamma"], w["lr"],
                                 w["de"], w["dr"], w["adv"], w["adv_temp"], w["reg_coef"], w["reg_norm"])
    # warm-up
    for p in plans[:2]:
        model.step(p)
    # the intra-op thread count that is fastest on THIS host (all cores oversubscribe a step this
    # small): probe a few counts on 12 steps each, then time the sample with the best one
    tried = {}
    for nt in sorted({1, 4, 8, 16, 32, 64, nthreads}):
        if nt > nthreads:
            continue
        th.set_num_threads(nt)
        model.step(plans[2])
        t0 = time.perf_counter()
        for p in plans[3:15]:
            model.step(p)
        tried[nt] = round(12 * w["B"] / (time.perf_counter() - t0), 1)
    best = max(tried, key=tried.get)
    th.set_num_threads(best)
    t0 = time.perf_counter()
    n = 0
    for p in plans[15:15 + max_steps]:
        model.step(p)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    th.set_num_threads(nthreads)
    single = n * w["B"] / dt
    out = {"value": round(single, 1), "unit": "edges/s", "cores": best, "kind": "port",
           "sample": "%d steps of the same workload (%s, B=%d N=%d D=%d), torch-CPU port of the "
                     "reference ops (oracle/torch_port.py); the CPU side runs WITHOUT a sampler, the GPU `value` INCLUDES sampling; intra-op "
                     "threads = the fastest of the probed counts on this %d-core host"
                     % (n, w["model"], w["B"], w["N"], w["hidden"], nthreads),
           "ms_per_step": round(1e3 * dt / max(n, 1), 3), "edges_per_s_by_threads": tried,
           "single_process": {"value": round(single, 1), "threads": best}}
    # the reference's own way to use a many-core host: --num_proc P single-thread trainer processes, lock-free on
    # one shared-memory table (train.py:298-317).  `value` is the better of the two configurations.
    try:
        procs = max(2, min(64, (os.cpu_count() or 2)))
        hv, hsteps = torch_port.hogwild_cpu(w, procs, seconds=5.0)
        out["num_proc"] = {"value": round(hv, 1), "procs": procs, "steps": hsteps, "seconds": 5.0,
                           "semantics": "P single-thread processes, Hogwild on shared-memory tables (reference --num_proc P)"}
        if hv > single:
            out["value"], out["cores"] = round(hv, 1), procs
            out["ms_per_step"] = round(1e3 * w["B"] / hv, 3)          # aggregate: one step of ANY process every ... ms
            out["sample"] = ("%d steps in 5 s by %d single-thread processes sharing the tables (reference --num_proc %d; "
                             "%s, B=%d N=%d D=%d), torch-CPU port of the reference ops (oracle/torch_port.py); the CPU side runs "
                             "WITHOUT a sampler, the GPU `value` INCLUDES sampling; the single-process run with %d intra-op threads: %.0f edges/s"
                             % (hsteps, procs, procs, w["model"], w["B"], w["N"], w["hidden"], best, single))
    except Exception as e:  # noqa: BLE001 - the multi-process leg must never hide the single-process number
        out["num_proc"] = {"error": repr(e)}
    return out


def hogwild_measure(w, eng, dev, trainers, steps, G, flags, skew=False):
    """K concurrent trainers on ONE GPU (K HIP streams, K graph chains) updating the same tables
    lock-free - the reference's multi-process Hogwild mode (`--num_proc K`, train.py:298-317,
    docs/source/train.rst:138) with the processes replaced by streams.  Results are
    order-nondeterministic exactly like the reference's racy index_add_; reported next to the
    strict single-trainer number, never instead of it."""
    from dglke_amd import plan
    from dglke_amd.dataloader import UniformChunkedSampler
    from dglke_amd.engine import StepEngine
    tables = (eng.ent, eng.ent_state, eng.rel, eng.rel_state)
    per = max(G, (steps // trainers // G) * G)
    engines, graphs, streams = [], [], []
    for k in range(trainers):
        h, r, t = synth_triples(w, 100 + k, skew)
        smp = UniformChunkedSampler(h, r, t, w["n_ent"], w["B"], w["N"], dev, seed=100 + k)
        batches = plan.upload(smp.next_plans(G * 2), dev)
        e = StepEngine(w["model"], w["n_ent"], w["n_rel"], w["hidden"], w["gamma"], w["lr"], dev, w["de"],
                       w["dr"], w["adv"], w["adv_temp"], w["reg_coef"], w["reg_norm"], flags=flags,
                       tables=tables)
        for b in batches:
            e.workspace_for(b)
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            e.step(batches[0])
        torch.cuda.synchronize()
        gs = [e.capture(batches[:G], stream=st), e.capture(batches[G:], stream=st)]
        engines.append(e); graphs.append(gs); streams.append(st)
    torch.cuda.synchronize()

    def run(nrep):
        for rep in range(nrep):
            for k in range(trainers):
                with torch.cuda.stream(streams[k]):
                    graphs[k][rep % 2].replay()
    run(2)
    torch.cuda.synchronize()
    nrep = per // G
    t0 = time.perf_counter()
    run(nrep)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    total = nrep * G * trainers
    sums = [e.read_loss_sums() for e in engines]
    ok = all(np.isfinite(s_[2]) for s_ in sums)
    return {"trainers": trainers, "steps_total": total, "value": round(total * w["B"] / wall, 1),
            "unit": "edges/s", "us_per_step_aggregate": round(1e6 * wall / total, 3), "finite_loss": bool(ok),
            "semantics": "Hogwild: concurrent lock-free trainers on shared tables (reference --num_proc K)"}


def async_measure(w, dev, steps, G, flags, seed=0, skew=False):
    """the --async_update pipeline (kge_step_async: the entity update of step s-1 runs on a side stream under the
    scoring of step s; exact one-step staleness inside a group of G steps, flushed at the group's end) on its own
    tables, same workload, sampler inside the timed region.  Reported next to the strict number, never as `value`."""
    from dglke_amd.dataloader import DeviceSampler
    from dglke_amd.engine import StepEngine
    h, r, t = synth_triples(w, seed, skew)
    eng = StepEngine(w["model"], w["n_ent"], w["n_rel"], w["hidden"], w["gamma"], w["lr"], dev, w["de"], w["dr"], w["adv"],
                     w["adv_temp"], w["reg_coef"], w["reg_norm"], flags=flags)
    smp = DeviceSampler(h, r, t, w["n_ent"], w["B"], w["N"], dev, n_slots=G, seed=seed)
    eng.steps_async(smp.sample())
    torch.cuda.synchronize()
    eng.reset_parameters()
    g = torch.cuda.CUDAGraph()
    with _kge_lib.graph_capture(g):
        eng.steps_async(smp.sample())
    nrep = max(1, steps // G)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    eng.loss_accum.zero_()
    t0 = time.perf_counter()
    for _ in range(nrep):
        g.replay()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    sums = eng.read_loss_sums()
    return {"value": round(nrep * G * w["B"] / wall, 1), "unit": "edges/s", "steps": nrep * G,
            "us_per_step": round(1e6 * wall / (nrep * G), 3), "group": G, "mean_loss": round(sums[2] / (nrep * G), 6),
            "relation_trace": "deferred too" if flags & 64 else "synchronous (reference: entity table only)",
            "semantics": "--async_update: step s gathers rows that hold every update up to s-2 (exact one-step staleness, "
                         "bit-reproducible); groups of %d steps, flushed at the end of each group" % G}


def skew_measure(w, dev, steps, G, flags, seed=0, skew=True):
    """the strict step on heavy-tailed ids (synth_triples(skew=True): FB15k's hub proportions) on its own tables, sampler inside
    the timed region: what uniform synthetic ids hide - rows with 20 - 40 contributions per batch.  Reported next to the
    headline (uniform ids, the shape BASELINE.json's synthetic metric is defined on), never as `value`."""
    from dglke_amd.dataloader import DeviceSampler
    from dglke_amd.engine import StepEngine
    h, r, t = synth_triples(w, seed, skew)
    eng = StepEngine(w["model"], w["n_ent"], w["n_rel"], w["hidden"], w["gamma"], w["lr"], dev, w["de"], w["dr"], w["adv"],
                     w["adv_temp"], w["reg_coef"], w["reg_norm"], flags=flags)
    smp = DeviceSampler(h, r, t, w["n_ent"], w["B"], w["N"], dev, n_slots=G, seed=seed)
    for b in smp.sample():
        eng.step(b)
    torch.cuda.synchronize()
    eng.reset_parameters()
    g = torch.cuda.CUDAGraph()
    with _kge_lib.graph_capture(g):
        for b in smp.sample():
            eng.step(b)
    nrep = max(1, steps // G)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    eng.loss_accum.zero_()
    t0 = time.perf_counter()
    for _ in range(nrep):
        g.replay()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    sums = eng.read_loss_sums()
    return {"value": round(nrep * G * w["B"] / wall, 1), "unit": "edges/s", "steps": nrep * G,
            "us_per_step": round(1e6 * wall / (nrep * G), 3), "mean_loss": round(sums[2] / (nrep * G), 6),
            "ids": ("entity k with weight 1/(k+10)^0.9, relation k with 1/(k+5): most frequent relation ~3.6 % of the edges, "
                    "hub entity ~1 % of the heads / tails (FB15k's proportions)") if skew else "uniform"}


def other_configs(steps=600, timeout_s=150.0, only=None):
    """BASELINE.json configs[2..4] at N = 1, each as a bounded leg in its OWN process (a leg that fails or hangs can never hide
    `value`; its tables - 4 GB for wikikg2, 34 GB for the Freebase shard - are gone when it returns): the strict step of
    DistMult FB15k, ComplEx on the real 2 500 604-entity wikikg2 table, and cfg-R's per-GPU step (RotatE hidden 400 -de over one
    10 756 769-row shard of the Freebase entity table) through the all-to-all engine at world 1 and through the peer-to-peer shard
    map.  Every entry: us_per_step, edges_per_s, the algorithmic bytes of its step and the fraction of the 8 TB/s roofline."""
    import subprocess
    legs = [("distmult_fb15k", ["--workload", "distmult_fb15k"], {}),
            ("complex_wikikg2", ["--workload", "complex_wikikg2"], {}),
            ("rotate_freebase_a2a", ["--workload", "rotate_freebase"], {"KGE_DIST_MODE": "a2a"}),
            ("rotate_freebase_p2p", ["--workload", "rotate_freebase"], {"KGE_DIST_MODE": "p2p"}),
            # the N > 1 code path on one GPU: the SAME a2a step with its RCCL collectives kept at world 1 (ids once per group, rows
            # and gradient messages per step, relation all-gather), kernels + collectives replayed from one hipGraph per group
            ("rotate_freebase_a2a_forced_exchange", ["--workload", "rotate_freebase"],
             {"KGE_DIST_MODE": "a2a", "KGE_DIST_FORCE_COLL": "1", "KGE_DIST_PIPELINE": "0"}),
            # ... and as `bench.py --gpus N` runs it at N > 1: triples partitioned by relation (no relation exchange), synchronous, then
            # with every exchange off the compute stream (DistEngine._steps_overlapped: push + owner-side apply of step s and the
            # pull of step s+2 next to step s+1; one-step-stale entity rows - the reference's --async_update licence, which its own
            # recipe for this config passes)
            ("rotate_freebase_a2a_forced_exchange_relpart", ["--workload", "rotate_freebase"],
             {"KGE_DIST_MODE": "a2a", "KGE_DIST_FORCE_COLL": "1", "KGE_DIST_PIPELINE": "0", "KGE_DIST_REL_PART": "force"}),
            ("rotate_freebase_a2a_forced_exchange_relpart_overlapped", ["--workload", "rotate_freebase"],
             {"KGE_DIST_MODE": "a2a", "KGE_DIST_FORCE_COLL": "1", "KGE_DIST_PIPELINE": "overlap", "KGE_DIST_REL_PART": "force"})]
    # BASELINE configs[1]'s own graph through the N > 1 code path (north_star: FB15k-shaped triples at 1/2/4/8 GPUs): the a2a engine
    # with its RCCL collectives kept at world 1, relation partitioning like the reference's 8-GPU FB15k recipe
    # (examples/fb15k/multi_gpu.sh:111-126), synchronous and with every exchange off the compute stream.  A 31-us step against
    # three collectives per step: what the exchange costs when nothing is big enough to hide it
    legs += [("transe_l2_fb15k_a2a_forced_exchange_relpart", ["--workload", "transe_l2_fb15k"],
              {"KGE_FORCE_DIST": "1", "KGE_DIST_MODE": "a2a", "KGE_DIST_FORCE_COLL": "1", "KGE_DIST_PIPELINE": "0",
               "KGE_DIST_REL_PART": "force"}),
             ("transe_l2_fb15k_a2a_forced_exchange_relpart_overlapped", ["--workload", "transe_l2_fb15k"],
              {"KGE_FORCE_DIST": "1", "KGE_DIST_MODE": "a2a", "KGE_DIST_FORCE_COLL": "1", "KGE_DIST_PIPELINE": "overlap",
               "KGE_DIST_REL_PART": "force"})]
    res = {}
    for name, extra, env_extra in legs:
        if only is not None and name not in only:
            continue
        env = dict(os.environ)
        env.update(env_extra)
        env.update({"KGE_DIST_OTHER_LEG": "0", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0",
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(29600 + (os.getpid() + len(res)) % 300)})
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(min(120, steps)),
               "--no-cpu-baseline", "--hogwild", "0", "--no-async-update", "--no-configs"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                res[name] = {"error": "exit %d: %s" % (r.returncode, (r.stderr or "")[-300:])}
                continue
            d = json.loads(line[-1])
            rf = d.get("roofline", {})
            res[name] = {"us_per_step": round(1e3 * d["ms_per_step"], 3), "edges_per_s": d["value"], "steps": d["steps"],
                         "algorithmic_bytes_per_step": rf.get("algorithmic_bytes_per_step", rf.get("algorithmic_bytes_per_rank_step")),
                         "frac": rf.get("frac"), "mean_loss": d.get("mean_loss"),
                         "leg_wall_s": round(time.perf_counter() - t0, 1)}
            # measured fabric bytes per step of this leg (committed PMC passes, profiles/latest_traffic_<leg>.json) and their ratio
            # to the algorithmic bytes
            tr = rf.get("traffic") or measured_traffic(name)
            if tr and res[name]["algorithmic_bytes_per_step"]:
                res[name]["traffic"] = tr
                res[name]["traffic_ratio"] = round(tr / res[name]["algorithmic_bytes_per_step"], 3)
            if "mode" in d.get("config", {}):
                res[name]["mode"] = d["config"]["mode"]
            if len(str(d.get("config", {}).get("launch") or "")) in range(1, 40):      # (a2a engine: "graph" / "eager")
                res[name]["launch"] = d["config"]["launch"]
            for k in ("schedule", "relation_partition"):
                if k in d.get("config", {}):
                    res[name][k] = d["config"][k]
        except subprocess.TimeoutExpired:
            res[name] = {"error": "leg exceeded %.0f s" % timeout_s}
        except Exception as e:  # noqa: BLE001 - a leg must never hide the headline
            res[name] = {"error": repr(e)}
    return res


def rank_eval_leg(model="TransE_l2", n_test=50000, batch=4096):
    """one filtered evaluation of the FB15k-shaped graph - 2 x n_test test triples against all 14 951 entities, 592 213 known triples -
    through dglke_amd.eval.evaluate as the trainers call it (filter lists built on the device and cached; the ranking of the
    matrix-form models is a tiled fp32-MFMA GEMM, csrc/kge_rank_gemm.hip): seconds of the first call (lists + workspace) and of a
    cached call, and the cached call's rate against the dense fp32-MFMA peak (2 x rows x candidates x dim flops per side)."""
    from dglke_amd import eval as kev
    w = WORKLOADS["transe_l2_fb15k"]
    n_ent, n_rel, D, gamma = w["n_ent"], w["n_rel"], w["hidden"], w["gamma"]
    rng = np.random.RandomState(0)
    known = tuple(rng.randint(0, n, 592213) for n in (n_ent, n_rel, n_ent))
    test = tuple(k[:n_test] for k in known)
    emb_init = (gamma + 2.0) / D
    torch.manual_seed(0)
    ent = torch.empty(n_ent, D, device="cuda").uniform_(-emb_init, emb_init)
    rel = torch.empty(n_rel, D, device="cuda").uniform_(-emb_init, emb_init)
    cache, secs = {}, []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = kev.evaluate(model, ent, rel, gamma, emb_init, test, known, batch=batch, cache=cache)
        torch.cuda.synchronize()
        secs.append(time.perf_counter() - t0)
    best = min(secs[1:])
    flops = 2.0 * 2 * n_test * n_ent * D
    return {"model": model, "triples": 2 * n_test, "candidates": n_ent, "dim": D, "known_triples": 592213,
            "first_call_s": round(secs[0], 4), "cached_call_s": round(best, 4),
            "tflops": round(flops / best / 1e12, 1), "frac_fp32_mfma_peak": round(flops / best / 157.3e12, 3),
            "mrr_of_random_tables": round(m["MRR"], 5),
            "what": "whole call incl. pos-side vectors, the tiled ranking GEMM (rank_gemm_kernel: 128 x 128 tiles, one comparison bit per "
                    "pair), ranks from the mask, metrics; wall clock around a synchronised call"}


def time_to_mrr(timeout_s=240.0, target=0.65):
    """the second half of BASELINE.json's metric as a bounded leg in its own process: `dglke_train` with the reference's FB15k
    TransE_l2 recipe (examples/fb15k/multi_gpu.sh:83-95: batch 1000, neg 200, dim 400, gamma 19.9, lr 0.25, -adv, rc 1e-9,
    max_step 24000), filtered validation every 500 steps, stop at validation MRR >= 0.65.  Real FB15k files are used when an
    operator supplied them (FB15K_DIR, $KGE_DATA_PATH/FB15k, data/FB15k); there is no network here, so otherwise the graph is the
    FB15k-shaped PLANTED one of tools/make_planted_fb15k.py and the entry says so (`graph`)."""
    import re
    import subprocess
    import tempfile
    real = os.environ.get("FB15K_DIR") or os.path.join(os.environ.get("KGE_DATA_PATH", os.path.join(ROOT, "data")), "FB15k")
    common = ["--model_name", "TransE_l2", "--no_save_emb", "--gpu", "0", "--batch_size", "1000", "--neg_sample_size", "200",
              "--hidden_dim", "400", "--gamma", "19.9", "--lr", "0.25", "-adv", "--regularization_coef", "1e-9", "--max_step", "24000",
              "--log_interval", "1000", "--eval_interval", "500", "--valid", "--test", "--target_mrr", str(target),
              "--graph_steps", "100", "--batch_size_eval", "16"]
    cli = os.path.join(ROOT, "dgl-ke_amd", "dglke_train")
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(prefix="kge_ttm_") as tmp:
        if os.path.isfile(os.path.join(real, "train.txt")) and os.path.isfile(os.path.join(real, "entities.dict")):
            graph = "fb15k"
            data = ["--dataset", "FB15k", "--data_path", os.path.dirname(os.path.abspath(real))]
        else:
            graph = "planted"
            d = os.path.join(tmp, "fb15k_planted")
            g = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_planted_fb15k.py"), d], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, text=True, timeout=timeout_s)
            if g.returncode != 0:
                return {"error": "planted graph: %s" % (g.stderr or "")[-300:]}
            data = ["--format", "udd_hrt", "--dataset", "fb15k_planted", "--data_path", d,
                    "--data_files", "entities.dict", "relations.dict", "train.txt", "valid.txt", "test.txt"]
        try:
            r = subprocess.run([sys.executable, cli] + data + ["--save_path", os.path.join(tmp, "ckpts")] + common,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                               timeout=max(10.0, timeout_s - (time.perf_counter() - t0)))
        except subprocess.TimeoutExpired:
            return {"graph": graph, "error": "leg exceeded %.0f s" % timeout_s}
    if r.returncode != 0:
        return {"graph": graph, "error": "exit %d: %s" % (r.returncode, (r.stderr or "")[-300:])}
    txt = r.stdout
    hit = re.search(r"validation MRR ([0-9.]+) >= [0-9.]+ after (\d+) steps, ([0-9.]+) s of training", txt)
    valid_mrr = [float(x) for x in re.findall(r"\]Valid average MRR: ([0-9.eE+-]+)", txt)]
    evals = [float(x) for x in re.findall(r"validation take ([0-9.]+) seconds", txt)]
    test_mrr = re.search(r"\]Test average MRR: ([0-9.eE+-]+)", txt)
    test_s = re.search(r"testing takes ([0-9.]+) seconds", txt)
    out = {"graph": graph, "target_mrr": target, "reached": bool(hit),
           "mrr": float(hit.group(1)) if hit else (valid_mrr[-1] if valid_mrr else None),
           "steps": int(hit.group(2)) if hit else 24000,
           "train_seconds": float(hit.group(3)) if hit else None,
           "eval_seconds": round(sum(evals), 3), "validations": len(evals),
           "test_mrr": float(test_mrr.group(1)) if test_mrr else None,
           "test_seconds": float(test_s.group(1)) if test_s else None,
           "leg_wall_s": round(time.perf_counter() - t0, 1),
           "recipe": "dglke_train TransE_l2, the reference's FB15k recipe (examples/fb15k/multi_gpu.sh:83-95), --valid every 500 steps "
                     "(filtered ranking of every validation triple against all entities, both sides), stop at MRR >= %g; "
                     "train_seconds = the training steps only, eval_seconds = the validations up to the stop" % target,
           "note": ("REAL FB15k (operator-supplied files): this is BASELINE.json's time-to-MRR metric" if graph == "fb15k" else
                    "PLANTED FB15k-shaped graph (14 951 entities, 1 345 relations, 483 142 / 50 000 / 59 071 triples, 10 % noisy tails; "
                    "tools/make_planted_fb15k.py) - real FB15k is not available offline, so the MRR trajectory is NOT FB15k's: only the step "
                    "and evaluation rates carry over (reference: MRR 0.649 after 24 000 steps = 167 s on one V100)")}
    if not hit and not valid_mrr:
        out["error"] = "no validation line in the CLI output: %s" % txt[-300:]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--workload", default="transe_l2_fb15k",
                    choices=sorted(WORKLOADS) + ["transe_l2_freebase", "rotate_freebase"])
    ap.add_argument("--pool", type=int, default=480, help="pre-staged batches (cycled)")
    ap.add_argument("--graph-steps", type=int, default=120, help="steps per captured HIP graph")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skew", action="store_true", help="heavy-tailed entity / relation ids (FB15k's proportions) instead of uniform")
    ap.add_argument("--force-pairwise", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="extra kge_hparams.flags bits (tuning)")
    ap.add_argument("--no-adv", action="store_true", help="tuning: disable -adv")
    ap.add_argument("--reg-coef", type=float, default=None, help="tuning: override the regularisation coefficient")
    ap.add_argument("--hidden", type=int, default=None,
                    help="tuning: override the workload's hidden dimension (the line's config.workload names the dimension that ran)")
    ap.add_argument("--host-plan", action="store_true",
                    help="pre-stage host-built batches instead of sampling on the device inside the timed region")
    ap.add_argument("--sampler-mode", default="auto", choices=["auto", "fused", "streams", "fork", "fork_tail", "serial"],
                    help="device sampler: the next group's batches are built by one sampler launch in front of every group on the "
                         "same stream (default), concurrently on a second stream, or on a forked branch of the group's hipGraph - the "
                         "concurrent modes make the steps 3.5 us slower each on ROCm 7.0 (profiles/r03_merged_fwd.txt)")
    ap.add_argument("--hogwild", type=int, default=4, help="also measure K concurrent Hogwild trainers (0 = skip)")
    ap.add_argument("--no-configs", dest="configs", action="store_false",
                    help="skip the bounded legs of BASELINE configs[2..4] (the `configs` object of the default line)")
    ap.add_argument("--no-time-to-mrr", dest="time_to_mrr", action="store_false",
                    help="skip the bounded time-to-MRR@0.65 leg (dglke_train on real FB15k when supplied, else on the planted graph)")
    ap.add_argument("--min-untimed", type=int, default=120,
                    help="steps run back to back right in front of the timed region: max(--warmup, this).  The first ~250 us of step "
                         "kernels after a pause of the queue run ~50 us late (clock ramp of the idle GPU, "
                         "profiles/r03_v2_driver_shape.txt); 0 = exactly --warmup steps")
    ap.add_argument("--no-async-update", dest="async_update", action="store_false",
                    help="skip the --async_update pipeline measurement (reported as its own object)")
    args = ap.parse_args()
    # (bench_dist.dist_workload_name: `transe_l2_fb15k` on the sharded engines only when it was asked for; without the flag the
    #  N > 1 headline is BASELINE configs[4])
    args.workload_explicit = any(a == "--workload" or a.startswith("--workload=") for a in sys.argv[1:])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("KGE_FORCE_DIST") or args.workload in ("transe_l2_freebase", "rotate_freebase"):
        import bench_dist
        return bench_dist.main(args, world, rank, local_rank)
    if args.gpus != 1:
        raise SystemExit("--gpus %d needs torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))

    import __graft_entry__
    __graft_entry__.build()
    from dglke_amd import _lib, plan
    from dglke_amd.dataloader import UniformChunkedSampler
    from dglke_amd.engine import StepEngine

    w = dict(WORKLOADS[args.workload])
    if args.no_adv:
        w["adv"] = False
    if args.reg_coef is not None:
        w["reg_coef"] = args.reg_coef
    if args.hidden is not None:
        w["hidden"] = args.hidden
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.manual_seed(0)
    h, r, t = synth_triples(w, 0, args.skew)
    import math
    eng = StepEngine(w["model"], w["n_ent"], w["n_rel"], w["hidden"], w["gamma"], w["lr"], dev, w["de"],
                     w["dr"], w["adv"], w["adv_temp"], w["reg_coef"], w["reg_norm"],
                     flags=(_lib.FLAG_FORCE_PAIRWISE if args.force_pairwise else 0) | args.flags)
    C = w["B"] // w["N"]
    use_graph = not args.no_graph
    G = args.graph_steps
    dev_sampler = (not args.host_plan) and 2 * w["B"] + C * w["N"] <= 4096
    if dev_sampler:
        G = max(2, G - G % 2)            # even group: slot parity = head / tail corruption (sampler.py:853-859)
    if dev_sampler:
        # ---- sampling + plan ON THE DEVICE, inside the timed region: one sampler launch per group of <= G steps over
        # double-buffered slots (dataloader.PrefetchedGroups): every timed group trains on batches built during the group before
        # it and builds the batches of the group after it - the timed region contains exactly K steps and the sampling of K
        # batches.  --sampler-mode serial (default): the launch follows the group's steps on the same stream ----
        from dglke_amd.dataloader import DeviceSampler, PrefetchedGroups
        smp = DeviceSampler(h, r, t, w["n_ent"], w["B"], w["N"], dev, n_slots=2 * G, seed=0)
        dbs = smp.sample(G)
        for b in dbs[:1]:
            eng.workspace_for(b)
        torch.cuda.synchronize()
        plans = []
        for k in range(min(8, G)):       # host copies of a few batches: only to count traced rows (bytes)
            a_ = smp.slot_arrays(k)
            plans.append(plan.build_plan(a_["h_gid"], a_["t_gid"], a_["rel_ids"], a_["neg_ids"], w["N"], w["N"],
                                         dbs[k].neg_head))
        for b in dbs:                    # eager warm-up of every kernel before capture
            eng.step(b)
        torch.cuda.synchronize()

        def sizes(count):
            return [G] * (count // G) + ([count % G] if count % G else [])
        n_warm = max(args.warmup, args.min_untimed)
        seq_w, seq_t = sizes(n_warm), sizes(args.steps)
        seq = seq_w + seq_t
        if args.sampler_mode == "auto":
            # round 5: the step's own launches build the next group's batches (tail workgroups, kge_step_fused_sampling) where the step
            # is the 4-launch strict step of the matrix-core family; everything else keeps the launch behind the group
            plain = args.flags == 0 and not args.force_pairwise
            args.sampler_mode = "fused" if (w["model"] in ("TransE_l2", "DistMult", "ComplEx") and plain and
                                            eng.d_e % (8 if w["model"] == "ComplEx" else 4) == 0 and eng.d_r == eng.d_e) else "serial"
        auto_mode = args.sampler_mode == "fused" and "fused" not in sys.argv
        pg = PrefetchedGroups(smp, eng.step, group_max=G, mode=args.sampler_mode, fused_max=int(os.environ.get("KGE_FUSED_MAX", "64")))

        def run_groups(lo, hi):          # groups seq[lo:hi]; group i builds the batches of group i + 1 (the last one: of a
            for i in range(lo, hi):      # group like the first timed one, so that K batches are sampled per K timed steps)
                pg.run(seq[i + 1] if i + 1 < len(seq) else seq_t[0], graph=use_graph)

        def start():
            pg.buf, pg.ready = 0, None
            pg.prefill(seq[0])
        if use_graph:                    # dry run of the whole schedule: captures one graph per distinct
            start()                      # (group size, next size, buffer half); then start again from fresh parameters
            try:
                run_groups(0, len(seq))
                torch.cuda.synchronize() if auto_mode else None      # (a failure of the tail path must show up HERE, not in the timed region)
            except Exception as e:       # noqa: BLE001 - the automatically chosen tail path must never cost the line: fall back to the launch
                if not auto_mode:
                    raise
                print("sampler tail unavailable (%r): falling back to the sampler launch" % (e,), file=sys.stderr)
                torch.cuda.synchronize()
                args.sampler_mode = "serial"
                pg = PrefetchedGroups(smp, eng.step, group_max=G, mode="serial")
                start()
                run_groups(0, len(seq))
        # (no synchronise here or behind the warm-up's preparation: everything is stream-ordered, and a queue that pauses for the
        #  host's bookkeeping starts its next burst with a ~50-us stall, profiles/r03_v2_driver_shape.txt - the one synchronise
        #  that opens the timed region is the contract's)
        eng.reset_parameters()
        start()
        run_w = lambda: run_groups(0, len(seq_w))
        run_t = lambda: run_groups(len(seq_w), len(seq))
        def launch_desc():
            # what the TIMED groups actually did (PrefetchedGroups.stats, reset behind the warm-up): a group's successor is built by
            # tail workgroups of the group's own launches only when 0 < next <= current <= fused_max steps, else by a sampler launch
            st = pg.stats
            ng = st["fused"] + st["launch"] + st["none"]
            how = {"streams": "a sampler launch on a second stream next to the group", "fork": "a sampler launch on a forked branch of the graph",
                   "fork_tail": "a sampler launch on a branch of the graph forked in front of the group's last step",
                   "serial": "ONE sampler launch behind the group's steps on the same stream",
                   "fused": "ONE sampler launch behind the group's steps on the same stream"}[args.sampler_mode]
            parts = []
            if st["fused"]:
                parts.append("%d of the %d timed groups: NO sampler launch - step k of the group builds batch k of the next group with 4 + 5 + 5 "
                             "tail workgroups on its own first / backward / update launches (kge_step_fused_sampling; groups of <= %d steps)"
                             % (st["fused"], ng, pg.fused_max))
            if st["launch"]:
                parts.append("%d of the %d timed groups: %s" % (st["launch"], ng, how))
            d = (("hipGraph per group of <= %d steps (timed groups: %s); the next group's batches: %s"
                  % (G, "+".join(str(x) for x in seq_t), "; ".join(parts) or "none built"))
                 if use_graph else "eager, sampler mode " + args.sampler_mode)
            d += ("; untimed before the timed region: %s%d warm-up steps (max(--warmup %d, --min-untimed %d): the GPU's clocks "
                  "have ramped when the one synchronise opens the timed region; the warm-up groups of more than %d steps use the sampler launch)"
                  % (("a dry run of the whole schedule (%d steps, graph capture), then fresh parameters and " % (n_warm + args.steps))
                     if use_graph else "", n_warm, args.warmup, args.min_untimed, pg.fused_max))
            return d
        data_desc = ("triples in HBM; batch ids, negatives and plan built ON THE DEVICE inside the timed region "
                     "(double-buffered: group g+1 is sampled while group g trains)")
    else:
        # ---- host-built batches + plans pre-staged in HBM (sampler outside the timed region) ----
        sampler = UniformChunkedSampler(h, r, t, w["n_ent"], w["B"], w["N"], dev, seed=0)
        pool = max(G, (args.pool // G) * G)
        plans = sampler.next_plans(pool)
        batches = plan.upload(plans, dev)
        for b in batches:
            eng.workspace_for(b)
        graphs = {}

        def get_graph(start, count):
            key = (start, count)
            if key not in graphs:
                graphs[key] = eng.capture([batches[(start + k) % pool] for k in range(count)])
            return graphs[key]

        def schedule(pos, count):
            items = []
            while count > 0:
                seg_off = pos % G
                n = min(G - seg_off, count)
                items.append((pos % pool, n))
                pos += n
                count -= n
            return items, pos

        n_warm = max(args.warmup, args.min_untimed)
        warm_items, pos = schedule(0, n_warm)
        timed_items, pos = schedule(pos, args.steps)
        if use_graph:
            eng.step(batches[0])
            torch.cuda.synchronize()
            eng.reset_parameters()
            for it in warm_items + timed_items:
                get_graph(*it)
            torch.cuda.synchronize()

        def run(items):
            if use_graph:
                for it in items:
                    graphs[it].replay()
            else:
                for start, n in items:
                    for k in range(n):
                        eng.step(batches[(start + k) % pool])
        run_w = lambda: run(warm_items)
        run_t = lambda: run(timed_items)
        launch_desc = lambda: "hipGraph of %d steps" % G if use_graph else "eager"
        pg = None
        data_desc = "id batches + plan built on the host and pre-staged in HBM, sampler excluded"

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run_w()
    if pg is not None:
        pg.reset_stats()              # (host-side counters: which sampler path the TIMED groups take)
    eng.loss_accum.zero_()            # (stream order: behind the warm-up steps) the sums of the timed steps only
    torch.cuda.synchronize()
    ev0.record()                      # (in front of the clock: the GPU is idle, its time stamp precedes the first kernel)
    t0 = time.perf_counter()
    run_t()
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    K = args.steps
    accum = eng.read_loss_sums()
    assert all(np.isfinite(accum[:3])), "non-finite loss: %r" % (accum,)

    bytes_step = algorithmic_bytes(plans, eng.d_e, eng.d_r)
    step_s = (ev_ms * 1e-3) / K
    achieved = bytes_step / step_s / 1e9
    out = {
        "metric": "positive edges/sec (whole node)",
        "value": round(K * w["B"] / wall, 1),
        "unit": "edges/s",
        "n_gpus": 1, "steps": K, "warmup": args.warmup,
        # what actually ran untimed right in front of the timed region: max(--warmup, --min-untimed) steps (config.launch says why)
        "warmup_effective": n_warm,
        "ms_per_step": round(1e3 * wall / K, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s synthetic FB15k-shaped (%s ids): n_ent=%d n_rel=%d batch=%d neg=%d dim=%d "
                               "gamma=%g lr=%g adv=%s rc=%g, full tables in HBM, %s" % (
                                   w["model"], "heavy-tailed" if args.skew else "uniform", w["n_ent"], w["n_rel"], w["B"],
                                   w["N"], w["hidden"], w["gamma"], w["lr"], w["adv"], w["reg_coef"], data_desc),
                   "global_batch": w["B"], "parallelism": "1 GPU",
                   "launch": launch_desc(),
                   "sampler_groups_timed": dict(pg.stats) if pg is not None else None,
                   "neg_kernels": "pairwise" if args.force_pairwise else "auto"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(achieved / 8000.0, 5), "traffic": measured_traffic(args.workload),
                     "kernel": step_kernels(w["model"], args.force_pairwise),
                     "algorithmic_bytes_per_step": round(bytes_step, 1),
                     "event_ms_per_step": round(ev_ms / K, 6)},
        "mean_loss": round(accum[2] / K, 6),
    }
    if out["roofline"]["traffic"]:
        # measured fabric bytes per step (PMC passes of the committed profile) over the algorithmic bytes: re-read factor
        out["roofline"]["traffic_ratio"] = round(out["roofline"]["traffic"] / bytes_step, 3)
    prof = profiled_kernels(args.workload)
    if prof:
        # self-check: the dominant kernel of the committed profile, its share of the step and its own roofline fraction
        top = max(v["calls"] for v in prof["kernels"].values())
        ks = {k: v for k, v in prof["kernels"].items() if v["calls"] * 2 >= top}       # the once-per-step kernels
        dom = max(ks, key=lambda k: ks[k]["avg_us"])
        out["roofline"]["dominant_kernel"] = dom.split("(")[0].replace("void ", "")
        out["roofline"]["dominant_kernel_us"] = ks[dom]["avg_us"]
        out["roofline"]["profiled_step_us"] = round(sum(v["avg_us"] for v in ks.values()), 3)
        out["roofline"]["profile_build"] = prof.get("build")
        if "neg_bwd_gemm" in dom:
            d_row = w["hidden"] * (2 if w["de"] else 1)
            tf = 4.0 * w["B"] * w["N"] * d_row / (ks[dom]["avg_us"] * 1e-6) / 1e12
            out["roofline"]["mfma_dominant_frac"] = round(tf / 157.3, 5)
    pm = profile_matches_build()
    out["roofline"]["profile_matches_build"] = pm["matches"]
    out["roofline"]["build_source_hash"] = pm["build_source_hash"]
    out["roofline"]["profile_source_hashes"] = pm["profile_source_hashes"]
    if w["model"] in ("TransE_l2", "DistMult", "ComplEx", "SimplE") and not args.force_pairwise:
        # second bound of SURVEY 8(d): the chunked negative score and its two gradient products on the
        # fp32 matrix cores, 2·B·N·D forward + 4·B·N·D backward, against the dense fp32 MFMA peak
        d_row = w["hidden"] * (2 if w["de"] else 1)
        flops = 6.0 * w["B"] * w["N"] * d_row
        tf = flops / (ev_ms / K * 1e-3) / 1e12
        out["roofline"]["mfma"] = {"flops_per_step": flops, "achieved": round(tf, 3), "peak": 157.3,
                                   "unit": "TFLOP/s", "frac": round(tf / 157.3, 5),
                                   "note": "whole step; the two GEMM kernels alone: see profiles/*kernel_stats*"}
    if args.hogwild > 1:
        try:
            out["hogwild"] = hogwild_measure(w, eng, dev, args.hogwild, K, max(G, 60), eng.hp.flags, args.skew)
        except Exception as e:
            out["hogwild"] = {"error": repr(e)}
    if args.async_update and dev_sampler and w["model"] not in ("RESCAL", "TransR"):
        try:
            out["async_update"] = async_measure(w, dev, K, G, eng.hp.flags, skew=args.skew)
            out["async_update_rel"] = async_measure(w, dev, K, G, eng.hp.flags | 64, skew=args.skew)
            # which of these the CLI runs: `dglke_train --async_update` (entity table only, the reference's flag) maps onto the
            # STRICT step = `value` (zero staleness is within the flag's licence); these two legs are the one-step-stale pipeline
            out["async_update"]["cli_flag"] = "--async_update --async_update_pipeline"
            out["async_update"]["cli_maps_to"] = "plain --async_update runs the strict step (`value`), not this pipeline"
            out["async_update_rel"]["cli_flag"] = "--async_update --async_update_rel"
        except Exception as e:
            out["async_update"] = {"error": repr(e)}
    if args.async_update and dev_sampler and not args.skew and w["model"] not in ("RESCAL", "TransR"):
        try:
            out["heavy_tailed_ids"] = skew_measure(w, dev, min(K, 1200), G, eng.hp.flags)
        except Exception as e:
            out["heavy_tailed_ids"] = {"error": repr(e)}
    if args.async_update and dev_sampler and not args.skew and not args.flags and w["model"] in ("TransE_l2", "DistMult", "ComplEx"):
        # the 3-launch form of the strict step (KGE_FLAG_LOSS_IN_FWD: LossGenerator inside the first launch through a
        # last-arriver hand-off; built in round 4, parity-tested, SLOWER - profiles/r04_loss_fold.txt): measured beside the
        # default 4-launch step on every run so that the record stays current
        try:
            m3 = skew_measure(w, dev, min(K, 1200), G, eng.hp.flags | 1024, skew=False)
            m3["what"] = "strict step as 3 launches (KGE_FLAG_LOSS_IN_FWD, opt-in); `value` is the default 4-launch step"
            out["strict_step_3_launches"] = m3
        except Exception as e:
            out["strict_step_3_launches"] = {"error": repr(e)}
    if args.configs and args.workload == "transe_l2_fb15k" and not args.skew and not args.flags:
        del eng                        # (the legs run in their own processes; the 34-GB shard needs the HBM this one holds)
        torch.cuda.empty_cache()
        out["configs"] = other_configs()
    if args.time_to_mrr and args.configs and args.workload == "transe_l2_fb15k" and not args.skew and not args.flags:
        try:
            eng = None                 # (the leg runs in its own process)
            torch.cuda.empty_cache()
            out["time_to_mrr"] = time_to_mrr()
        except Exception as e:  # noqa: BLE001 - a leg must never hide the headline
            out["time_to_mrr"] = {"error": repr(e)}
    if args.time_to_mrr and args.configs and args.workload == "transe_l2_fb15k" and not args.skew and not args.flags:
        try:
            out["rank_eval"] = rank_eval_leg()
        except Exception as e:  # noqa: BLE001
            out["rank_eval"] = {"error": repr(e)}
    if not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(w, plans)
        except Exception as e:  # the baseline must never hide the GPU number
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
