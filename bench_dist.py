"""bench_dist.py - N>1 leg of bench.py: weak-scaled, one process per GPU (torch.distributed, backend
nccl = RCCL), entity AND relation tables range-sharded over the ranks' HBM.

Default workload = BASELINE.json configs[4]: RotatE on a synthetic Freebase-sized graph (hidden 400 -de: D_e = 800,
D_r = 400; per-GPU batch 1024, neg 256; 86 054 151 entities / 14 824 relations at 8 GPUs - examples/README.md:11 of
the reference), WEAK-scaled in both dimensions: every GPU runs the same step and holds the same 1/8 of the Freebase
entity table (10 756 769 rows = 34.4 GB), so N GPUs train a graph of N/8 x Freebase and N = 8 is configs[4]
itself.  The SAME workload runs at N = 1 (`python bench.py --gpus 1 --workload rotate_freebase`: one shard,
kge_step_sharded through a 1-entry shard map), so 1 / 2 / 4 / 8 is one curve; the N = 1 DEFAULT of bench.py stays
configs[1] (the configuration BASELINE.json's metric is quoted on).  `--workload transe_l2_freebase`: the N=1
bench's step on the same tables.

Two multi-GPU modes (KGE_DIST_MODE):
  p2p (default)  the shared-table Hogwild mode of the reference's multi-GPU trainer with the shared
                 table living in the union of the GPUs' HBM: every rank maps all peer shards
                 (hipIpc) and kge_step_sharded reads / updates remote rows directly over xGMI.  No
                 collective and no host work per step; [1 sampler launch + G steps] per hipGraph.
  a2a            the parameter-server semantics (pull -> compute -> push, owner applies) as RCCL
                 all-to-all collectives (dglke_amd/dist.py).  Also the automatic fall-back when the
                 peer mappings cannot be established.
value = (steps x batch x N) / max-over-ranks wall time.
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

DIST_WORKLOADS = {
    "transe_l2_freebase": dict(model="TransE_l2", n_ent=86054151, n_rel=14824, hidden=400, de=False,
                               dr=False, B=1000, N=200, gamma=10.0, lr=0.1, adv=True, adv_temp=1.0,
                               reg_coef=1e-9, reg_norm=3),
    "rotate_freebase": dict(model="RotatE", n_ent=86054151, n_rel=14824, hidden=400, de=True, dr=False,
                            B=1024, N=256, gamma=12.0, lr=0.01, adv=True, adv_temp=1.0,
                            reg_coef=1e-7, reg_norm=3),
}


def _p2p_setup(args, world, rank, dev, w, n_ent, d_e, d_r, emb_init):
    """shared tables over the peers' HBM + on-device sampler + one hipGraph per G steps."""
    import math
    from dglke_amd import p2p
    from dglke_amd.dataloader import DeviceSampler
    from dglke_amd.engine import StepEngine
    tabs = p2p.ShardedTables(n_ent, w["n_rel"], d_e, d_r, dev, world, rank)
    if not tabs.probe():
        raise RuntimeError("peer mappings do not reach the other GPUs' memory")
    tabs.init_uniform(emb_init, 1234)
    eng = StepEngine(w["model"], n_ent, w["n_rel"], w["hidden"], w["gamma"], w["lr"], dev, w["de"], w["dr"],
                     w["adv"], w["adv_temp"], w["reg_coef"], w["reg_norm"], shards=tabs)
    # this rank's edge shard (reference: RandomPartition of the training triples, sampler.py:256-290):
    # synthetic uniform triples over the GLOBAL id space, generated in HBM
    n_train = int(os.environ.get("KGE_DIST_TRIPLES", min(338586276 // world, 48_000_000)))
    g = torch.Generator(device=dev)
    g.manual_seed(777 + rank)
    H = torch.randint(0, n_ent, (n_train,), device=dev, generator=g)
    T = torch.randint(0, n_ent, (n_train,), device=dev, generator=g)
    R = torch.randint(0, w["n_rel"], (n_train,), device=dev, generator=g)
    G = max(2, min(120, args.graph_steps) // 2 * 2)      # even group: slot parity = head / tail corruption
    smp = DeviceSampler(H, R, T, n_ent, w["B"], w["N"], dev, n_slots=G, seed=rank + 1)
    dbs = smp.sample()
    eng.workspace_for(dbs[0])
    for b in dbs:                       # eager warm-up of every kernel before capture
        eng.step(b)
    torch.cuda.synchronize()

    def group():
        for b in smp.sample():
            eng.step(b)
    gr = None
    if not args.no_graph:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            group()
        torch.cuda.synchronize()

    def partial(n):                     # one sampler launch + the first n < G steps of the group
        for b in smp.sample()[:n]:
            eng.step(b)
    rem_graphs = {}
    if gr is not None:
        for n in {args.warmup % G, args.steps % G} - {0}:
            rem_graphs[n] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(rem_graphs[n]):
                partial(n)
        torch.cuda.synchronize()

    def run(count):                     # EXACTLY count steps: full groups, then one partial group
        for _ in range(count // G):
            if gr is not None:
                gr.replay()
            else:
                group()
        if count % G:
            if gr is not None:
                rem_graphs[count % G].replay()
            else:
                partial(count % G)
    C = w["B"] // w["N"]
    # traced rows per step for the byte accounting: count them on one sampled batch
    a_ = smp.slot_arrays(0)
    ue = int(np.unique(np.concatenate([a_["h_gid"], a_["t_gid"], a_["neg_ids"]])).shape[0])
    u_pos = int(np.unique(np.concatenate([a_["h_gid"], a_["t_gid"]])).shape[0])
    rows = dict(UE=ue, R_e=u_pos + C * w["N"], B=w["B"])
    desc = ("tables sharded over the GPUs' HBM and mapped peer-to-peer (hipIpc): remote rows read and "
            "updated directly over xGMI, Hogwild across ranks (reference --num_proc shared-table semantics), "
            "no collective in the step; %s; sampling + plan on the device inside the timed region"
            % (("hipGraph of [1 sampler launch + %d steps]" % G) if gr is not None else "eager launches"))
    return eng, run, rows, desc, tabs


def _a2a_setup(args, world, rank, dev, w, n_ent, d_e, emb_init):
    """parameter-server semantics over RCCL all-to-all (dglke_amd/dist.py)."""
    from dglke_amd import plan
    from dglke_amd import dist as kd
    from dglke_amd.engine import StepEngine
    spec = kd.ShardSpec(n_ent, world, rank)
    torch.manual_seed(1234 + rank)
    ent = torch.empty(spec.n_local, d_e, dtype=torch.float32, device=dev).uniform_(-emb_init, emb_init)
    ent_state = torch.zeros(spec.n_local, dtype=torch.float32, device=dev)
    torch.manual_seed(99)          # identical relation replicas on every rank
    eng = StepEngine(w["model"], 1, w["n_rel"], w["hidden"], w["gamma"], w["lr"], dev, w["de"], w["dr"],
                     w["adv"], w["adv_temp"], w["reg_coef"], w["reg_norm"])
    de = kd.DistEngine(eng, spec, ent, ent_state)
    # pre-stage a pool of batches: ids (uniform, like the N=1 bench), plans, routes
    pool = max(8, min(args.pool, 64))
    rng = np.random.RandomState(1000 + rank)
    C = w["B"] // w["N"]
    plans, ues = [], []
    for s in range(pool):
        h = rng.randint(0, n_ent, w["B"]).astype(np.int64)
        t = rng.randint(0, n_ent, w["B"]).astype(np.int64)
        r = rng.randint(0, w["n_rel"], w["B"]).astype(np.int64)
        neg = rng.randint(0, n_ent, C * w["N"]).astype(np.int64)
        ue, p = kd.localize_plan(h, t, r, neg, w["N"], w["N"], (s + 1) % 2 == 0)
        plans.append(p)
        ues.append(ue)
    batches = plan.upload(plans, dev)
    routes = [de.prepare_route(ue) for ue in ues]
    for b in batches:
        eng.workspace_for(b)
    de.max_rows = int(max(max(r.UE for r in routes), max(r.n_recv for r in routes)) * 1.25) + 64
    de.step(batches[0], routes[0])     # allocates every persistent buffer
    torch.cuda.synchronize()
    state = {"pos": 0}

    def run(count):
        for _ in range(count):
            i = state["pos"] % pool
            de.step(batches[i], routes[i])
            state["pos"] += 1
    rows = dict(UE=float(np.mean([p["UE"] for p in plans])),
                R_e=float(np.mean([p["U"] + C * w["N"] for p in plans])), B=w["B"])
    desc = ("entity table range-sharded, relation table replicated, RCCL all-to-all pull/push with owner-side "
            "Adagrad (parameter-server semantics), eager launches, host-built batches pre-staged")
    return eng, run, rows, desc


def main(args, world, rank, local_rank):
    import __graft_entry__
    __graft_entry__.build()      # serialised by a file lock; a no-op when the .so is current

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if not dist.is_initialized():
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    name = args.workload if args.workload in DIST_WORKLOADS else "rotate_freebase"
    w = dict(DIST_WORKLOADS[name])
    # weak scaling of the table as well: every GPU holds 1/8 of the Freebase entity table, N = 8 is the full graph
    n_ent = int(os.environ.get("KGE_DIST_ENTITIES", (w["n_ent"] + 7) // 8 * world))
    d_e = 2 * w["hidden"] if w["de"] else w["hidden"]
    d_r = 2 * w["hidden"] if w["dr"] else w["hidden"]
    emb_init = (w["gamma"] + 2.0) / w["hidden"]

    mode = os.environ.get("KGE_DIST_MODE", "p2p")
    eng = run = rows = desc = tabs = None
    why = ""
    if mode == "p2p":
        ok = 1
        try:
            eng, run, rows, desc, tabs = _p2p_setup(args, world, rank, dev, w, n_ent, d_e, d_r, emb_init)
        except Exception as e:           # noqa: BLE001 - any set-up problem means: use the collectives
            ok, why = 0, repr(e)
            print("[rank %d] peer-to-peer set-up failed (%s): falling back to all-to-all" % (rank, why),
                  file=sys.stderr)
        flag = torch.tensor([ok], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            mode, eng, run, tabs = "a2a", None, None, None
            torch.cuda.empty_cache()
    if mode != "p2p":
        eng, run, rows, desc = _a2a_setup(args, world, rank, dev, w, n_ent, d_e, emb_init)

    run(args.warmup)
    torch.cuda.synchronize()
    dist.barrier()
    eng.loss_accum.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    dist.barrier()
    wall = time.perf_counter() - t0
    tw = torch.tensor([wall], dtype=torch.float64, device=dev)
    dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    wall = float(tw.item())
    sums = eng.read_loss_sums()
    K = args.steps
    # secondary leg at N > 1: the parameter-server semantics over RCCL all-to-all (the partitioning north_star names:
    # entity table range-sharded, relation table replicated, pull / push collectives), eager launches, a bounded
    # number of steps, under a watchdog - a hung collective must never cost the headline line
    a2a_leg = {"result": None}
    want_a2a = mode == "p2p" and world > 1 and os.environ.get("KGE_DIST_A2A_LEG", "1") != "0"

    def emit(a2a):
        if rank != 0:
            return
        line = _result_line(args, w, n_ent, world, wall, K, rows, d_e, eng.d_r, desc, mode, why, sums, a2a)
        print(json.dumps(line), flush=True)

    if want_a2a:
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(float(os.environ.get("KGE_DIST_A2A_TIMEOUT", "90"))):
                emit({"error": "all-to-all leg did not finish in time (watchdog)"})
                os._exit(0)
        th = threading.Thread(target=watchdog, daemon=True)
        th.start()
        try:
            a_steps = max(20, min(K, 200))
            aeng, arun, arows, adesc = _a2a_setup(args, world, rank, dev, w, n_ent, d_e, emb_init)
            arun(min(20, a_steps))
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            arun(a_steps)
            torch.cuda.synchronize(); dist.barrier()
            aw = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(aw, op=dist.ReduceOp.MAX)
            a2a_leg["result"] = {"value": round(a_steps * w["B"] * world / float(aw.item()), 1), "unit": "edges/s",
                                 "steps": a_steps, "us_per_step": round(1e6 * float(aw.item()) / a_steps, 2),
                                 "collectives_per_step": "2 all_to_all_single (rows, packed gradients) + 1 all_gather (relation "
                                                         "gradients); ids routed ahead", "launch": "eager", "desc": adesc}
        except Exception as e:          # noqa: BLE001
            a2a_leg["result"] = {"error": repr(e)}
        done.set()
    emit(a2a_leg["result"])
    dist.barrier()
    if tabs is not None:
        tabs.close()
    dist.destroy_process_group()


def _result_line(args, w, n_ent, world, wall, K, rows, d_e, d_r, desc, mode, why, sums, a2a):
    if True:
        # algorithmic bytes per rank-step (SURVEY 8d formula on the sampled batches)
        bytes_step = 12.0 * (rows["R_e"] * d_e + rows["B"] * d_r) + 16.0 * (rows["R_e"] + rows["B"])
        # rows crossing xGMI per rank-step: every traced row is read once and read-modify-written once
        xgmi_step = 3.0 * (rows["R_e"] * d_e + rows["B"] * d_r) * 4 * (world - 1) / world
        out = {
            "metric": "positive edges/sec (whole node)",
            "value": round(K * w["B"] * world / wall, 1), "unit": "edges/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(1e3 * wall / K, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s synthetic Freebase-scale (BASELINE configs[4], weak-scaled: %d/8 of the 86 054 151-entity "
                                   "graph): n_ent=%d n_rel=%d, per-GPU batch=%d neg=%d hidden=%d (D_e=%d) over %d GPU%s "
                                   "(%.1f GB of entity rows per GPU); %s"
                                   % (w["model"], world, n_ent, w["n_rel"], w["B"], w["N"], w["hidden"], d_e, world,
                                      "" if world == 1 else "s", (n_ent + world - 1) // world * d_e * 4 / 1e9, desc),
                       "global_batch": w["B"] * world,
                       "parallelism": ("shared tables over %d GPUs' HBM, peer-to-peer xGMI (Hogwild)" % world)
                       if mode == "p2p" else ("entity-shard x%d (RCCL all-to-all)" % world),
                       "mode": mode, "fallback_reason": why or None},
            "roofline": {"bound": "hbm", "achieved": round(bytes_step * world / (wall / K) / 1e9, 2),
                         "peak": 8000.0 * world, "unit": "GB/s",
                         "frac": round(bytes_step / (wall / K) / 1e9 / 8000.0, 5), "traffic": None,
                         "algorithmic_bytes_per_rank_step": round(bytes_step, 1),
                         "xgmi_bytes_per_rank_step": round(xgmi_step, 1),
                         "xgmi_GBps_per_gpu": round(xgmi_step / (wall / K) / 1e9, 2),
                         "xgmi_peak_GBps_per_gpu": 1071.0},
            "mean_loss": round(sums[2] / K, 6),
            "n1_same_workload": "python bench.py --gpus 1 --workload %s" % args.workload if args.workload in DIST_WORKLOADS
                                else "python bench.py --gpus 1 --workload rotate_freebase",
        }
        if a2a is not None:
            out["a2a"] = a2a
        return out
