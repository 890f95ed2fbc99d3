"""bench_dist.py - N>1 leg of bench.py: weak-scaled, one process per GPU (torch.distributed, backend
nccl = RCCL), the entity table range-sharded over the ranks' HBM.

Default workload = BASELINE.json configs[4]: RotatE on a synthetic Freebase-sized graph (hidden 400 -de: D_e = 800,
D_r = 400; per-GPU batch 1024, neg 256; 86 054 151 entities / 14 824 relations at 8 GPUs - examples/README.md:11 of
the reference), WEAK-scaled in both dimensions: every GPU runs the same step and holds the same 1/8 of the Freebase
entity table (10 756 769 rows = 34.4 GB), so N GPUs train a graph of N/8 x Freebase and N = 8 is configs[4]
itself.  The SAME workload runs at N = 1 (`python bench.py --gpus 1 --workload rotate_freebase`: one shard, the same
route -> pull -> step -> push -> apply schedule without collectives), so 1 / 2 / 4 / 8 is one curve; the N = 1 DEFAULT of bench.py stays
configs[1] (the configuration BASELINE.json's metric is quoted on).  `--workload transe_l2_freebase`: the N=1
bench's step on the same tables.

Two multi-GPU modes (KGE_DIST_MODE), one is the headline, the other a bounded secondary leg on the same line:
  a2a (default)  BASELINE.json's north_star partitioning: entity table range-sharded, relation table replicated, the
                 parameter-server semantics (pull -> compute -> push, owner applies) as RCCL all-to-all collectives
                 (dglke_amd/dist.py): routing on the device, fixed-size messages, no host work in the step.  Per sampled group
                 one sampler launch + the bucket-capacity check, then ONE hipGraph of [routing of the group, one all-to-all of
                 the group's request ids, the steps with their row / gradient exchanges] (DistEngine.run_group, round 5: RCCL
                 collectives DO replay from hipGraphs - profiles/r05_rccl_capture_diagnosis.txt): the synchronous schedule is
                 timed and delivered, then the same K steps with every exchange on a side stream (DistEngine._steps_overlapped:
                 one-step-stale entity rows, the reference's --async_update licence) - the faster one is the line's value.  The
                 supervisor falls back to eager launches (overlapped), then to the synchronous eager step, the c10d wrappers, p2p
                 and independent replicas.
  p2p            the shared-table Hogwild mode of the reference's multi-GPU trainer with the shared table living in the
                 union of the GPUs' HBM: every rank maps all peer shards (hipIpc) and kge_step_sharded reads / updates
                 remote rows directly over xGMI (BOTH tables sharded).  No collective and no host work per step.
value = (steps x batch x N) / max-over-ranks wall time.
"""
import json
import os
import sys
import time

import numpy as np
import torch

from dglke_amd import _lib as _kge_lib
import torch.distributed as dist

DIST_WORKLOADS = {
    "transe_l2_freebase": dict(model="TransE_l2", n_ent=86054151, n_rel=14824, hidden=400, de=False,
                               dr=False, B=1000, N=200, gamma=10.0, lr=0.1, adv=True, adv_temp=1.0,
                               reg_coef=1e-9, reg_norm=3),
    "rotate_freebase": dict(model="RotatE", n_ent=86054151, n_rel=14824, hidden=400, de=True, dr=False,
                            B=1024, N=256, gamma=12.0, lr=0.01, adv=True, adv_temp=1.0,
                            reg_coef=1e-7, reg_norm=3),
    # BASELINE configs[1]'s graph on N GPUs (north_star: "edges/sec on synthetic FB15k-shaped triples reported at 1/2/4/8 GPUs"; the
    # reference's own 8-GPU FB15k recipe, examples/fb15k/multi_gpu.sh:111-126: every trainer its own batches of 1000 over ONE
    # 14 951-entity table, --rel_part --async_update): the graph stays FB15k-sized at every N - the table is NOT weak-scaled, each
    # rank holds 1/N of its rows and 1/N of the 483 142 triples.  A 31-us step against three collectives: latency-bound on purpose
    "transe_l2_fb15k": dict(model="TransE_l2", n_ent=14951, n_rel=1345, hidden=400, de=False, dr=False,
                            B=1000, N=200, gamma=19.9, lr=0.25, adv=True, adv_temp=1.0,
                            reg_coef=1e-9, reg_norm=3, n_train=483142, fixed_graph=True),
}


def dist_workload_name(args):
    """the sharded workload of this invocation: `--workload` when it names one of DIST_WORKLOADS - for `transe_l2_fb15k` only when it
    was passed EXPLICITLY (it is bench.py's argparse default = the N = 1 headline, configs[1]; without the flag the N > 1 headline is
    BASELINE configs[4], the Freebase-scale RotatE config the north_star's scaling target is stated on)."""
    if args.workload in DIST_WORKLOADS and (args.workload != "transe_l2_fb15k" or getattr(args, "workload_explicit", False)):
        return args.workload
    return "rotate_freebase"


def dist_entities(w, world):
    """entities of the sharded graph at this world size: the graph itself (fixed_graph) or world / 8 of it (weak-scaled table:
    every GPU holds 1/8 of the Freebase entity table, N = 8 is the full graph); KGE_DIST_ENTITIES overrides"""
    if os.environ.get("KGE_DIST_ENTITIES"):
        return int(os.environ["KGE_DIST_ENTITIES"])
    return w["n_ent"] if w.get("fixed_graph") else (w["n_ent"] + 7) // 8 * world


def dist_triples(w, world):
    """this rank's edge shard (reference: RandomPartition of the training triples, sampler.py:256-290); KGE_DIST_TRIPLES overrides"""
    if os.environ.get("KGE_DIST_TRIPLES"):
        return int(os.environ["KGE_DIST_TRIPLES"])
    return w["n_train"] // world if w.get("fixed_graph") else min(338586276 // world, 48_000_000)


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
    n_train = dist_triples(w, world)
    g = torch.Generator(device=dev)
    g.manual_seed(777 + rank)
    H = torch.randint(0, n_ent, (n_train,), device=dev, generator=g)
    T = torch.randint(0, n_ent, (n_train,), device=dev, generator=g)
    R = torch.randint(0, w["n_rel"], (n_train,), device=dev, generator=g)
    G = max(2, min(120, args.graph_steps) // 2 * 2)      # even group: slot parity = head / tail corruption
    smp = DeviceSampler(H, R, T, n_ent, w["B"], w["N"], dev, n_slots=G, seed=rank + 1)
    dbs = smp.sample()
    eng.workspace_for(dbs[0])
    torch.cuda.synchronize()
    _progress("tables")
    for b in dbs:                       # eager warm-up of every kernel before capture
        eng.step(b)
    torch.cuda.synchronize()

    def group():
        for b in smp.sample():
            eng.step(b)
    gr = None
    if not args.no_graph:
        gr = torch.cuda.CUDAGraph()
        with _kge_lib.graph_capture(gr):
            group()
        torch.cuda.synchronize()

    def partial(n):                     # one sampler launch + the first n < G steps of the group
        for b in smp.sample()[:n]:
            eng.step(b)
    rem_graphs = {}
    if gr is not None:
        for n in {args.warmup % G, args.steps % G} - {0}:
            rem_graphs[n] = torch.cuda.CUDAGraph()
            with _kge_lib.graph_capture(rem_graphs[n]):
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
                if count % G not in rem_graphs:         # (a size nobody announced: captured on first use)
                    rem_graphs[count % G] = torch.cuda.CUDAGraph()
                    with _kge_lib.graph_capture(rem_graphs[count % G]):
                        partial(count % G)
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


def _a2a_setup(args, world, rank, dev, w, n_ent, d_e, emb_init, allow_force_coll=True):
    """parameter-server semantics over RCCL all-to-all (dglke_amd/dist.py): entity table range-sharded, relation table
    replicated, device-side routing, fixed-size messages, on-device sampler inside the timed region."""
    from dglke_amd import dist as kd
    from dglke_amd.dataloader import DeviceSampler
    from dglke_amd.engine import StepEngine
    spec = kd.ShardSpec(n_ent, world, rank)
    torch.manual_seed(1234 + rank)
    ent = torch.empty(spec.n_local, d_e, dtype=torch.float32, device=dev).uniform_(-emb_init, emb_init)
    ent_state = torch.zeros(spec.n_local, dtype=torch.float32, device=dev)
    torch.manual_seed(99)          # identical relation replicas on every rank
    eng = StepEngine(w["model"], 1, w["n_rel"], w["hidden"], w["gamma"], w["lr"], dev, w["de"], w["dr"],
                     w["adv"], w["adv_temp"], w["reg_coef"], w["reg_norm"])
    # KGE_DIST_FORCE_COLL=1: keep the RCCL calls and the pull pipeline at world 1 too (smoke test of the N > 1 code path on one GPU)
    force_coll = allow_force_coll and os.environ.get("KGE_DIST_FORCE_COLL", "0") == "1"
    # collectives: librccl called directly on the step's streams (dist.RcclComm; KGE_DIST_COMM=torch: the c10d wrappers)
    t_comm = time.perf_counter()
    comm = kd.make_comm() if (world > 1 or force_coll) else None
    t_comm = time.perf_counter() - t_comm        # (communicator creation: ncclCommInitRank over the id broadcast - the first contact with the peers)
    # relation partitioning (the reference's multi-GPU Freebase recipe passes --rel_part, examples/freebase/multi_gpu.sh:100-116): every
    # rank's triples use its own relations (r = rank mod world), relation rows are updated where their edges are - no relation
    # exchange.  KGE_DIST_REL_PART=0: uniform relations on every rank, relation messages all-gathered and applied by everyone
    # (KGE_DIST_REL_PART=force: also at world 1 - the forced-exchange proxy of the N > 1 default on one GPU)
    rel_part = (world > 1 and os.environ.get("KGE_DIST_REL_PART", "1") != "0") or os.environ.get("KGE_DIST_REL_PART") == "force"
    # owner buckets start at 1.25 x the mean share: every exchange moves whole buckets, pads included, and the uniform synthetic ids
    # fill a bucket to mean + 3.5 sigma = 1.17 x the mean at world 8 (384 +- 18 of 3 072 rows per owner); a group that needs more
    # grows them before it runs (ensure_capacity; config.bucket_growth says so).  The CLI's default stays 1.5 (real ids are skewed)
    de = kd.DistEngine(eng, spec, ent, ent_state, comm=comm, slack=float(os.environ.get("KGE_DIST_SLACK", "1.25")),
                       always_collective=force_coll, rel_local=rel_part)
    # this rank's edge shard: synthetic uniform triples over the GLOBAL id space, generated in HBM
    n_train = dist_triples(w, world)
    g = torch.Generator(device=dev)
    g.manual_seed(777 + rank)
    H = torch.randint(0, n_ent, (n_train,), device=dev, generator=g)
    T = torch.randint(0, n_ent, (n_train,), device=dev, generator=g)
    R = torch.randint(0, w["n_rel"], (n_train,), device=dev, generator=g)
    if rel_part:
        R = torch.clamp((R // world) * world + rank, max=(w["n_rel"] - 1 - rank) // world * world + rank)
    G = max(2, min(120, args.graph_steps) // 2 * 2)
    smp = DeviceSampler(H, R, T, n_ent, w["B"], w["N"], dev, n_slots=G, seed=rank + 1)
    de.bench_sampler = smp
    de.comm_create_s = round(t_comm, 3) if comm is not None else None
    # KGE_DIST_PIPELINE: 0 = synchronous steps; 1 = the pull of step s+1 next to step s (step_pipelined); overlap = push, owner-side
    # apply and pull all on the side stream, the compute stream runs the steps' kernels back to back (DistEngine._steps_overlapped:
    # the same one-step-stale dataflow, bit-identical tables)
    pipelined = (world > 1 or force_coll) and os.environ.get("KGE_DIST_PIPELINE", "1") != "0"
    if os.environ.get("KGE_DIST_PIPELINE") == "overlap":      # (also without collectives: owner-side apply + gather next to the steps)
        pipelined = "overlap"

    def steps(dbs):
        de._steps(dbs, pipelined)
    dbs = smp.sample()
    eng.workspace_for(dbs[0])
    torch.cuda.synchronize()
    _progress("tables")                 # everything local is in place; what follows is the first contact with the peers
    de.prepare_group(dbs)               # bucket capacity + the routing of the whole group in one launch (allocates the route pool)
    steps(dbs[:4])                      # eager warm-up: allocates every persistent buffer
    n_cg = 0
    # with collectives (N > 1, or forced at world 1): every group = [sampler launch, capacity check (the group's one device read)]
    # eagerly, then [routing of the group + ONE id exchange + its steps with their collectives] replayed from one hipGraph
    # (DistEngine.run_group; round 5 - captured RCCL collectives replay fine, see dist.RcclComm.close).  KGE_DIST_GRAPH=0 / a
    # communicator that cannot be recorded: the same calls launched one by one (host-bound)
    graph_coll = (de.coll and not args.no_graph and os.environ.get("KGE_DIST_GRAPH", "1") != "0" and
                  getattr(de.comm, "capturable", False))
    if de.coll and not graph_coll:      # eager step (collectives): its kernels between pull and push may replay from small hipGraphs
        torch.cuda.synchronize()
        n_cg = de.precapture(smp)
    torch.cuda.synchronize()
    graphs = {}
    use_graph = world == 1 and not force_coll and not args.no_graph      # no collective at world 1: the group replays from a hipGraph

    def group_(n):                      # one sampler launch, the group's buckets checked and ALL its batches routed in one launch,
        dbs_ = smp.sample(n)            # then the steps (world 1: no device read in prepare_group, so the whole thing is capturable)
        if de.coll:
            de.run_group(dbs_, graph=graph_coll, pipelined=pipelined)
        else:
            de.prepare_group(dbs_)
            steps(dbs_)

    def sizes(count):
        return [G] * (count // G) + ([count % G] if count % G else [])

    def run(count):                     # EXACTLY count steps: groups of G, then one partial group (one sampler launch each)
        for n in sizes(count):
            if use_graph:
                if n not in graphs:
                    graphs[n] = torch.cuda.CUDAGraph()
                    with _kge_lib.graph_capture(graphs[n]):
                        group_(n)
                graphs[n].replay()
            else:
                group_(n)
    if use_graph:                       # capture outside the timed region
        for n in set(sizes(args.warmup) + sizes(args.steps)):
            graphs[n] = torch.cuda.CUDAGraph()
            with _kge_lib.graph_capture(graphs[n]):
                group_(n)
        torch.cuda.synchronize()
    elif graph_coll:                    # the first group of every size runs eagerly and is recorded behind it: outside the timed region
        for n in sorted(set(sizes(args.warmup) + sizes(args.steps))):
            group_(n)
        torch.cuda.synchronize()
    C = w["B"] // w["N"]
    a_ = smp.slot_arrays(0)
    u_pos = int(np.unique(np.concatenate([a_["h_gid"], a_["t_gid"]])).shape[0])
    ue = int(np.unique(np.concatenate([a_["h_gid"], a_["t_gid"], a_["neg_ids"]])).shape[0])
    rows = dict(UE=ue, R_e=u_pos + C * w["N"], B=w["B"], cap=de.cap, cap2=getattr(de, "cap2", 0), rel_part=rel_part,
                launch="graph" if (use_graph or graph_coll) else "eager")
    rel_desc = ("triples partitioned by relation (--rel_part of the reference's recipe): relation rows updated on their owner rank, "
                "no relation exchange" if rel_part else "relation gradients all-gathered")
    comm_desc = ((": librccl called directly" + ("" if rel_part else ", push + relation exchange grouped"))
                 if type(de.comm).__name__ == "RcclComm" else ": torch.distributed wrappers") if de.coll else ""
    def describe(sched):
        """the line's description of this set-up for a schedule: False (synchronous), True (pull pipeline), "overlap" """
        pipe_desc = ("every exchange off the compute stream: push + owner-side apply of step s and the pull of step s+2 run next to step "
                     "s+1 (one-step-stale entity rows, --async_update licence)" if sched == "overlap" else
                     "pull of step s+1 overlapped with step s (one-step-stale rows, --async_update licence)")
        launch_desc = ("hipGraph of [1 sampler launch + %d steps]" % G if use_graph else
                       ("per group of <= %d steps: sampler launch + bucket-capacity check (one device read), then ONE hipGraph of [routing of "
                        "the group, one id all-to-all for the group, the steps with their RCCL collectives]%s" % (G, ("; " + pipe_desc + ", as a "
                        "fork inside the graph") if sched else "; synchronous schedule")) if graph_coll else
                       (("eager launches, " + pipe_desc if sched else "eager launches") +
                        (", the step's kernels between pull and push replayed from %d small hipGraphs" % n_cg if n_cg else "")))
        if getattr(de, "local_only", False):
            return ("entity table range-sharded (world 1: ONE shard = this GPU's %d-row table), relation table replicated; every row of a "
                    "batch is local, so the all-to-all engine runs the in-place step on the shard - no routing, no row cache, no gradient "
                    "messages, no owner-side apply launch (dist.DistEngine.local_only; the N > 1 path with its exchanges kept at world 1 "
                    "is the `*_a2a_forced_exchange*` legs); %s; sampling + plan on the device inside the timed region"
                    % (spec.n_local, launch_desc))
        return ("entity table range-sharded, relation table replicated; per step: device-side routing into %d-row owner buckets, "
                "all-to-all pull of the unique rows, the single-GPU kernels against the row cache, all-to-all push of one packed "
                "single-trace gradient message per row (a second one, in the bucket's small extra region, only for a row that is in both "
                "traces of the batch), owner-side Adagrad in rank order (one merged launch), %s "
                "(parameter-server semantics, RCCL%s); %s; sampling + plan on the device inside the timed region"
                % (de.cap or 0, rel_desc, comm_desc, launch_desc))
    de.describe = describe
    desc = describe(pipelined)
    return eng, run, rows, desc, de


def second_schedule_wins(first_value, leg, K):
    """may the schedule measured SECOND (the leg dict of main()) become the line's value?  Only when it ran exactly the K steps the
    line reports, between the same barriers (its wall time is there), finished without an error and is more than 2 % faster than
    the schedule timed first - whose line has already been delivered, so a leg that hangs or fails can only cost itself.
    KGE_DIST_PROMOTE=0: never."""
    if not leg or leg.get("error") or os.environ.get("KGE_DIST_PROMOTE", "1") == "0":
        return False
    if not leg.get("wall_s") or leg.get("steps") != K or not leg.get("value"):
        return False
    return leg["value"] > 1.02 * first_value


def _progress(mark):
    """worker -> orchestrator: one line per finished phase (the orchestrator's per-phase watchdog resets on every new line)"""
    path = os.environ.get("KGE_DIST_PROGRESS")
    if path:
        with open(path, "a") as f:
            f.write("%s %.3f\n" % (mark, time.time()))


def _deliver(line):
    """rank 0's JSON line: to the orchestrator's result file when there is one (it prints it), else to stdout"""
    path = os.environ.get("KGE_DIST_RESULT")
    if path:
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            f.write(line + "\n")
        os.replace(tmp, path)
        return False
    return True


# what the orchestrator tries, in order, until one attempt delivers a line on every rank: the north_star mode
# on the direct librccl communicator, the same on the c10d wrappers, the peer-to-peer shared tables, and - so that a node whose
# links do not come up still yields a measured line that says so - N independent replicas of the per-GPU step without any exchange
ATTEMPTS = (("a2a", "rccl-graph"), ("a2a", "rccl"), ("a2a", "rccl-sync"), ("a2a", "torch"), ("p2p", ""), ("replicas", ""))
# ("rccl-graph", round 5: kernels AND collectives of a group of steps replay from one hipGraph - DistEngine.run_group: the synchronous
#  schedule is timed and delivered, then the overlapped one, and the faster is the line's value; "rccl": the same calls as eager
#  launches, every exchange on a side stream - DistEngine._steps_overlapped)
# ("rccl-sync": the same direct communicator with the synchronous schedule - no pull on a side stream, one stream issues every
#  collective: if two streams sharing one communicator are what hangs, this attempt still measures the north_star mode)
# seconds allowed until the named progress mark appears (a hang shows up as a mark that does not come;
# KGE_DIST_PHASE_TIMEOUTS="start,tables,setup,warmup,timed,headline" overrides): start = interpreter + torch import + build check;
# tables = local allocations; setup = communicator + the first eager steps (first contact with the links); headline .. end =
# the secondary legs
PHASE_ORDER = ("start", "tables", "setup", "warmup", "timed", "headline", "end")
PHASE_BUDGET = {"start": 300.0, "tables": 120.0, "setup": 90.0, "warmup": 90.0, "timed": 120.0, "headline": 60.0, "end": 200.0}


def orchestrate(args, world, rank, local_rank):
    """N > 1: every torchrun rank becomes a supervisor that never touches the GPU.  It runs the measurement in a CHILD process
    (`KGE_DIST_WORKER=1`, its own rendezvous port per attempt), follows the child's progress marks with a per-phase watchdog, kills
    the child's process group when a phase does not end, agrees with the other supervisors (gloo, CPU) on the outcome and moves on to
    the next entry of ATTEMPTS - so that the first contact with xGMI can hang or crash without costing the JSON line.  Rank 0 prints
    ONE line: the first attempt's that completed on all ranks, with `config.mode`, `config.fallback_reason` and the attempts' history."""
    import signal
    import subprocess
    import tempfile
    if "MASTER_ADDR" not in os.environ:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
    base_port = int(os.environ["MASTER_PORT"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    budgets = dict(PHASE_BUDGET)
    if os.environ.get("KGE_DIST_PHASE_TIMEOUTS"):
        for k, v in zip(PHASE_ORDER, os.environ["KGE_DIST_PHASE_TIMEOUTS"].split(",")):
            budgets[k] = float(v)
    order = {a: b for a, b in zip(PHASE_ORDER[:-1], PHASE_ORDER[1:])}
    first = os.environ.get("KGE_DIST_MODE", "a2a")
    attempts = [a for a in ATTEMPTS if a[0] == first] + [a for a in ATTEMPTS if a[0] != first]
    if os.environ.get("KGE_DIST_COMM") == "torch":
        attempts = [a for a in attempts if a[1] not in ("rccl-graph", "rccl", "rccl-sync")]
    history, line = [], None
    tmpdir = tempfile.mkdtemp(prefix="kge_dist_%d_" % rank)
    for ai, (mode, comm) in enumerate(attempts):
        res_path = os.path.join(tmpdir, "result_%d.json" % ai)
        prog_path = os.path.join(tmpdir, "progress_%d.txt" % ai)
        # (torchrun's agent-store variables would make the child look for a store server on ITS port: the child hosts its own)
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
        env.update({"KGE_DIST_WORKER": "1", "KGE_DIST_MODE": mode, "KGE_DIST_RESULT": res_path, "KGE_DIST_PROGRESS": prog_path,
                    "MASTER_PORT": str(base_port + 1 + ai), "RANK": str(rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(world)})
        env["KGE_DIST_ATTEMPT"] = comm
        if comm == "rccl-graph":         # (the synchronous schedule is timed and DELIVERED first; the overlapped schedule - every exchange
            env["KGE_DIST_COMM"], env["KGE_DIST_GRAPH"] = "rccl", "1"       # off the compute stream - runs behind it and becomes the line's value when it is faster)
            env.setdefault("KGE_DIST_PIPELINE", "0")
        elif comm == "rccl-sync":
            env["KGE_DIST_COMM"], env["KGE_DIST_PIPELINE"], env["KGE_DIST_GRAPH"] = "rccl", "0", "0"
        elif comm:
            env["KGE_DIST_COMM"], env["KGE_DIST_GRAPH"] = comm, "0"
            if comm == "rccl":           # eager launches: every exchange on the side stream (117.7 vs 136.7 us synchronous on the world-1 proxy)
                env.setdefault("KGE_DIST_PIPELINE", "overlap")
        # (KGE_DIST_WORKER_SCRIPT: the CPU test of this supervisor substitutes a scripted worker, tests/test_bench_supervisor.py)
        cmd = [sys.executable, os.environ.get("KGE_DIST_WORKER_SCRIPT") or
               os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench.py")] + sys.argv[1:]
        errf = open(os.path.join(tmpdir, "stderr_%d.txt" % ai), "w")
        t_start = time.time()
        child = subprocess.Popen(cmd, env=env, stdout=errf, stderr=errf, start_new_session=True)
        phase, deadline, why, seen = "start", time.time() + budgets["start"], "", 0
        mark_times = []
        while True:
            rc = child.poll()
            marks = open(prog_path).read().split("\n") if os.path.exists(prog_path) else []
            mark_times = [(m.split()[0], float(m.split()[1])) for m in marks if len(m.split()) >= 2]
            marks = [m.split()[0] for m in marks if m.strip()]
            if len(marks) > seen:                       # a phase ended: the next one gets its own budget
                seen = len(marks)
                phase = marks[-1]
                # (after the headline the secondary legs set up engines of their own and repeat earlier marks: one budget for all that)
                deadline = time.time() + (budgets["end"] if "headline" in marks else budgets.get(order.get(phase, "end"), budgets["end"]))
            if rc is not None:
                if rc != 0:
                    why = "worker exited with status %d in the phase after '%s'" % (rc, phase)
                break
            if time.time() > deadline:
                why = "no progress for %.0f s in the phase after '%s' (watchdog)" % (
                    budgets.get(order.get(phase, "end"), 0.0) if seen else budgets["start"], phase if seen else "launch")
                try:
                    os.killpg(child.pid, signal.SIGKILL)
                except OSError:
                    pass
                child.wait()
                break
            time.sleep(0.25)
        errf.close()
        have = os.path.exists(res_path)                  # (replicas: every rank delivers; else rank 0 does)
        needs = rank == 0 or mode == "replicas"
        ok = 1 if ((have or not needs) and (not why or "headline" in marks)) else 0
        # a worker that hung or died AFTER its headline was delivered (a secondary leg) still counts: the line is there
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        reasons = [None] * world
        dist.all_gather_object(reasons, why)
        # seconds every rank spent until each progress mark of the worker (start = interpreter + import + build check, tables = local
        # allocations, setup = communicator + first eager steps, warmup, timed, headline): where a slow or failed attempt's time went
        spans, prev = {}, t_start
        for name, t in mark_times:
            if name not in spans:
                spans[name] = round(t - prev, 2)
            prev = t
        all_spans = [None] * world
        dist.all_gather_object(all_spans, spans)
        rec = {"mode": mode, "comm": comm or None, "ok": bool(int(flag.item())), "seconds": round(time.time() - t_start, 1),
               "why": next((r for r in reasons if r), None), "phase_seconds_per_rank": all_spans}
        if rec["why"] and not rec["ok"]:
            try:
                tail = open(os.path.join(tmpdir, "stderr_%d.txt" % ai)).read()[-400:]
                rec["stderr_tail_rank%d" % rank] = tail
            except OSError:
                pass
        history.append(rec)
        if int(flag.item()) == 1:
            if mode == "replicas":
                mine = json.loads(open(res_path).read())
                alls = [None] * world
                dist.all_gather_object(alls, mine)
                if rank == 0:
                    line = _replicas_line(args, world, alls, history)
            elif rank == 0:
                d = json.loads(open(res_path).read())
                if "diagnostics" in d.get("config", {}):       # the worker's per-rank diagnostics belong to the attempt that produced them
                    history[-1]["diagnostics"] = d["config"].pop("diagnostics")
                d["config"]["attempts"] = history
                if len(history) > 1:
                    d["config"]["fallback_reason"] = "; ".join("%s%s: %s" % (h["mode"], "/" + h["comm"] if h["comm"] else "", h["why"])
                                                                for h in history[:-1])
                line = json.dumps(d)
            break
    if rank == 0:
        if line is None:
            line = json.dumps({"metric": "positive edges/sec (whole node)", "value": 0.0, "unit": "edges/s", "n_gpus": world,
                               "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                               "config": {"workload": args.workload, "mode": None, "attempts": history,
                                          "fallback_reason": "every attempt failed"}})
        print(line, flush=True)
    dist.barrier()
    dist.destroy_process_group()
    import shutil
    shutil.rmtree(tmpdir, ignore_errors=True)


def _replicas_line(args, world, alls, history):
    """last resort of the fallback chain: N independent replicas of the per-GPU step (no exchange at all): value = N x K x B over the
    slowest replica's wall time.  Says what it is; not a scaling measurement of the sharded step."""
    w = DIST_WORKLOADS[dist_workload_name(args)]
    wall = max(a["wall"] for a in alls)
    K = args.steps
    return json.dumps({
        "metric": "positive edges/sec (whole node)", "value": round(K * w["B"] * world / wall, 1), "unit": "edges/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(1e3 * wall / K, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s per-GPU step on %d INDEPENDENT replicas (one shard-sized table each, NO exchange between the GPUs): "
                               "the fallback of the fallback chain - every sharded mode failed on this node" % (w["model"], world),
                   "global_batch": w["B"] * world, "parallelism": "replicas only", "mode": "replicas", "attempts": history,
                   "fallback_reason": "; ".join("%s%s: %s" % (h["mode"], "/" + h["comm"] if h["comm"] else "", h["why"])
                                                for h in history[:-1])},
        "per_rank_us_per_step": [round(1e6 * a["wall"] / K, 2) for a in alls]})


def _replica_worker(args, rank, local_rank):
    """KGE_DIST_MODE=replicas: this rank's per-GPU step on a one-rank engine over a shard-sized table; no process group."""
    import __graft_entry__
    __graft_entry__.build()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    w = dict(DIST_WORKLOADS[dist_workload_name(args)])
    d_e = 2 * w["hidden"] if w["de"] else w["hidden"]
    _progress("start")
    world = max(1, int(os.environ.get("WORLD_SIZE", "1")))
    n_shard = dist_entities(w, world) if w.get("fixed_graph") else dist_entities(w, world) // world
    eng, run, rows, desc, _ = _a2a_setup(args, 1, 0, dev, w, n_shard, d_e, (w["gamma"] + 2.0) / w["hidden"],
                                         allow_force_coll=False)
    _progress("setup")
    run(max(args.warmup, 20))
    torch.cuda.synchronize()
    _progress("warmup")
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    _progress("timed")
    _deliver(json.dumps({"wall": wall, "rank": rank}))
    _progress("headline")


def _fb15k_leg(args, world, rank, dev):
    """secondary leg of a Freebase-scale line: the FB15k-shaped graph (BASELINE configs[1]: TransE_l2, batch 1000, neg 200, dim 400,
    14 951 entities range-sharded over the ranks, 483 142 / N triples per rank) through the same all-to-all engine with the schedule
    of the headline, exactly --steps steps between barriers, and the same step on a one-rank engine over the whole table (the
    N = 1 point of THIS curve, = bench.py's own N = 1 headline workload)."""
    w2 = dict(DIST_WORKLOADS["transe_l2_fb15k"])
    emb2 = (w2["gamma"] + 2.0) / w2["hidden"]
    out = {"workload": "TransE_l2 synthetic FB15k-shaped (BASELINE configs[1]): n_ent=%d n_rel=%d, per-GPU batch=%d neg=%d dim=%d, "
                       "entity table range-sharded over %d GPU%s (graph NOT weak-scaled), %d triples per rank"
                       % (w2["n_ent"], w2["n_rel"], w2["B"], w2["N"], w2["hidden"], world, "" if world == 1 else "s",
                          dist_triples(w2, world))}
    eng2, run2, rows2, desc2, de2 = _a2a_setup(args, world, rank, dev, w2, w2["n_ent"], w2["hidden"], emb2)
    try:
        run2(args.warmup)
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        run2(args.steps)                 # (the group sizes _a2a_setup recorded: warm-up and --steps)
        torch.cuda.synchronize(); dist.barrier()
        tw = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        n = args.steps
        _pl = os.environ.get("KGE_DIST_PIPELINE", "1")
        sched = "overlapped" if _pl == "overlap" else "synchronous" if (_pl == "0" or not de2.coll) else "pipelined_pull"
        out.update({"value": round(n * w2["B"] * world / float(tw.item()), 1), "unit": "edges/s", "steps": n,
                    "us_per_step": round(1e6 * float(tw.item()) / n, 2), "schedule": sched, "launch": rows2.get("launch"),
                    "bucket_rows": de2.cap, "message_extra_rows": getattr(de2, "cap2", 0) or None,
                    "bucket_overflows": de2.check_overflow(), "desc": desc2})
        # the same engine's groups under the OTHER schedule (at this step size the exchanges are what a step costs: 64.5 synchronous
        # against 47.8 us overlapped on the world-1 proxy), replayed from their own hipGraphs: record, warm, time
        if de2.coll and rows2.get("launch") == "graph" and sched in ("synchronous", "overlapped"):
            other = "overlap" if sched == "synchronous" else False
            smp2, G2 = de2.bench_sampler, de2.bench_sampler.n_slots
            for _ in range(3):
                torch.cuda.synchronize(); dist.barrier()
                t0 = time.perf_counter()
                for k in [G2] * (n // G2) + ([n % G2] if n % G2 else []):
                    de2.run_group(smp2.sample(k), graph=True, pipelined=other)
                torch.cuda.synchronize(); dist.barrier()
                t2 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            out["other_schedule"] = {"schedule": "overlapped" if other else "synchronous", "steps": n,
                                     "us_per_step": round(1e6 * float(t2.item()) / n, 2),
                                     "value": round(n * w2["B"] * world / float(t2.item()), 1), "unit": "edges/s"}
    finally:
        de2.close()
    # the same graph on the peer-to-peer shared tables (no collective in the step: at this size the exchange is latency, and a
    # remote row costs one xGMI round instead of a share of three RCCL launches)
    try:
        peng, prun, prows, pdesc, ptabs = _p2p_setup(args, world, rank, dev, w2, w2["n_ent"], w2["hidden"], w2["hidden"], emb2)
        try:
            prun(args.warmup)
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            prun(args.steps)
            torch.cuda.synchronize(); dist.barrier()
            tp = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            out["p2p"] = {"value": round(args.steps * w2["B"] * world / float(tp.item()), 1), "unit": "edges/s", "steps": args.steps,
                          "us_per_step": round(1e6 * float(tp.item()) / args.steps, 2)}
        finally:
            ptabs.close()
    except Exception as e:              # noqa: BLE001
        out["p2p"] = {"error": repr(e)}
    seng, srun, _, _, sde = _a2a_setup(args, 1, 0, dev, w2, w2["n_ent"], w2["hidden"], emb2, allow_force_coll=False)
    s_steps = max(20, min(args.steps, 240))
    srun(min(20, s_steps))
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    srun(s_steps)
    torch.cuda.synchronize()
    sw = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(sw, op=dist.ReduceOp.MAX)
    out["per_gpu_step_without_exchange"] = {"us_per_step": round(1e6 * float(sw.item()) / s_steps, 2), "steps": s_steps,
                                            "edges_per_s_per_gpu": round(s_steps * w2["B"] / float(sw.item()), 1)}
    return out


def main(args, world, rank, local_rank):
    if world > 1 and os.environ.get("KGE_DIST_WORKER") != "1" and os.environ.get("KGE_DIST_SUPERVISE", "1") != "0":
        return orchestrate(args, world, rank, local_rank)
    if os.environ.get("KGE_DIST_SHARE_GPU") == "1":      # test aid: every rank on GPU 0 (the fallback chain on a one-GPU box)
        local_rank = 0
    if os.environ.get("KGE_DIST_MODE") == "replicas":
        return _replica_worker(args, rank, local_rank)
    import __graft_entry__
    __graft_entry__.build()      # serialised by a file lock; a no-op when the .so is current
    if os.environ.get("KGE_FAULTHANDLER"):      # developer aid: every thread's Python stack to stderr after N seconds (a hang's address)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["KGE_FAULTHANDLER"]), repeat=True, file=sys.stderr)

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    _progress("start")
    if not dist.is_initialized():
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    name = dist_workload_name(args)
    w = dict(DIST_WORKLOADS[name])
    # weak scaling of the table as well: every GPU holds 1/8 of the Freebase entity table, N = 8 is the full graph (dist_entities)
    n_ent = dist_entities(w, world)
    d_e = 2 * w["hidden"] if w["de"] else w["hidden"]
    d_r = 2 * w["hidden"] if w["dr"] else w["hidden"]
    emb_init = (w["gamma"] + 2.0) / w["hidden"]

    # headline mode = the partitioning BASELINE.json's north_star names (entity range-shard + RCCL all-to-all, relations
    # replicated); KGE_DIST_MODE=p2p makes the peer-to-peer shared-table mode the headline instead.  The other mode is timed as
    # a secondary leg (bounded, under a watchdog) and reported next to the headline, never instead of it.
    mode = os.environ.get("KGE_DIST_MODE", "a2a")
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
        eng, run, rows, desc, _de = _a2a_setup(args, world, rank, dev, w, n_ent, d_e, emb_init)

    # which schedule the headline run uses (a2a): KGE_DIST_PIPELINE 0 / 1 / overlap (_a2a_setup)
    _pl = os.environ.get("KGE_DIST_PIPELINE", "1")
    sched_name = ("overlapped" if _pl == "overlap" else
                  "synchronous" if (_pl == "0" or not (world > 1 or os.environ.get("KGE_DIST_FORCE_COLL") == "1")) else "pipelined_pull")
    _progress("setup")
    run(args.warmup)
    torch.cuda.synchronize()
    dist.barrier()
    _progress("warmup")
    eng.loss_accum.zero_()
    torch.cuda.synchronize()
    if os.environ.get("KGE_DIST_PROFILE") == "1" and rank == 0:      # developer aid: where the HOST time of the eager step goes
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        run(args.steps)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(22)
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    dist.barrier()
    wall = time.perf_counter() - t0
    _progress("timed")
    tw = torch.tensor([wall], dtype=torch.float64, device=dev)
    dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    wall = float(tw.item())
    sums = eng.read_loss_sums()
    K = args.steps
    overflow = _de.check_overflow() if mode != "p2p" else 0
    if mode != "p2p":
        rows["cap"], rows["grown"] = _de.cap, [list(g) for g in getattr(_de, "grown", [])] or None
        rows["cap2"], rows["grown_extra"] = getattr(_de, "cap2", 0), [list(g) for g in getattr(_de, "grown_extra", [])] or None
    eager = None
    other = "p2p" if mode == "a2a" else "a2a"
    # (KGE_DIST_OTHER_LEG=force: run the secondary legs at N = 1 too - a smoke test of this code path on one GPU)
    want_other = (world > 1 and os.environ.get("KGE_DIST_OTHER_LEG", "1") != "0") or os.environ.get("KGE_DIST_OTHER_LEG") == "force"
    import threading
    lock = threading.Lock()
    state = {"emitted": False}

    def emit(leg, now=True):
        with lock:                       # the watchdog and the main thread may both get here: ONE line
            if state["emitted"]:
                return None
            state["emitted"] = True
        if rank != 0:
            return None
        res = _result_line(args, w, n_ent, world, wall, K, rows, d_e, eng.d_r, desc, mode, why, sums, other, leg, overflow)
        res["config"]["schedule"] = sched_name
        if second_schedule_wins(res["value"], pipe_leg, K):
            # the OTHER schedule of the same engine ran exactly K steps between the same barriers and is faster: it is the line's
            # value (both are valid training under the flags of the reference's recipe for this config, which passes
            # --async_update); the schedule measured first stays beside it
            first = {"schedule": sched_name, "value": res["value"], "ms_per_step": res["ms_per_step"], "steps": K,
                     "what": "the schedule this worker timed first (delivered before the other one ran)"}
            res = _result_line(args, w, n_ent, world, pipe_leg["wall_s"], K, rows, d_e, eng.d_r, _de.describe(other_sched),
                               mode, why, sums, other, leg, overflow)
            res["config"]["schedule"] = pipe_leg["schedule"]
            res["mean_loss_of"] = "the %s run (timed first)" % sched_name
            res["a2a_graph_" + sched_name] = first
        elif pipe_leg is not None:
            res["a2a_graph_" + pipe_leg.get("schedule", "other")] = {k: v for k, v in pipe_leg.items() if k != "wall_s"}
        if eager is not None:
            res["a2a_eager"] = eager
        if local_leg is not None:
            res["per_gpu_step_without_exchange"] = local_leg
        if fb_leg is not None:
            res["fb15k_shaped"] = fb_leg
        if diag is not None:
            res["config"]["diagnostics"] = diag
        line = json.dumps(res)
        if now:
            if _deliver(line):
                print(line, flush=True)
        return line

    leg = None
    local_leg = None
    fb_leg = None
    # the headline is delivered NOW (result file of the orchestrator): a secondary leg that hangs or dies can only cost itself
    if rank == 0 and os.environ.get("KGE_DIST_RESULT"):
        _deliver(json.dumps(_result_line(args, w, n_ent, world, wall, K, dict(rows), d_e, eng.d_r, desc, mode, why, sums, other,
                                         None, overflow)))
    _progress("headline")
    # the same engine's groups once more under the OTHER schedule, replayed from their own hipGraphs: synchronous steps next to a
    # headline that overlaps its exchanges, or - the supervisor's first attempt times the synchronous schedule first - the
    # overlapped schedule (DistEngine._steps_overlapped: push, owner-side apply and pull on a side stream, bit-identical to the
    # pull pipeline; --async_update licence).  Exactly K steps between barriers, like the headline: when it is faster, emit()
    # makes it the line's value and keeps the first measurement beside it.  Which one wins depends on what the exchanges cost on
    # the links; at world 1 with forced collectives: 136.7 (synchronous) / 117.7 us (overlapped), profiles/r05_overlap_schedule.txt
    diag = None
    pipe_leg = None
    other_sched = False if sched_name != "synchronous" else "overlap"
    if (mode == "a2a" and _de.coll and rows.get("launch") == "graph" and
            (world > 1 or os.environ.get("KGE_DIST_PIPE_LEG") == "1") and os.environ.get("KGE_DIST_PIPE_LEG") != "0"):
        done_p = threading.Event()

        def watchdog_p():
            if not done_p.wait(float(os.environ.get("KGE_DIST_LEG_TIMEOUT", "120"))):
                emit({"error": "the second-schedule leg did not finish in time (watchdog)"})
                os._exit(0)
        threading.Thread(target=watchdog_p, daemon=True).start()
        try:
            l_steps = K if 20 <= K <= 2400 else max(20, min(K, 240))      # exactly K steps whenever that is affordable: only then may it become the line's value
            smp_ = _de.bench_sampler
            G_ = smp_.n_slots
            for _ in range(3):               # pass 1 records the graphs (eager), pass 2 warms, pass 3 is timed
                torch.cuda.synchronize(); dist.barrier()
                t0 = time.perf_counter()
                for n in [G_] * (l_steps // G_) + ([l_steps % G_] if l_steps % G_ else []):
                    _de.run_group(smp_.sample(n), graph=True, pipelined=other_sched)
                torch.cuda.synchronize(); dist.barrier()
                tp_ = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(tp_, op=dist.ReduceOp.MAX)
            pipe_leg = {"schedule": "overlapped" if other_sched else "synchronous",
                        "us_per_step": round(1e6 * float(tp_.item()) / l_steps, 2), "steps": l_steps,
                        "value": round(l_steps * w["B"] * world / float(tp_.item()), 1), "unit": "edges/s", "wall_s": float(tp_.item()),
                        "launch": ("every exchange off the compute stream: push + owner-side apply of step s and the pull of step "
                                   "s+2 run next to step s+1 (one-step-stale entity rows, --async_update licence)" if other_sched
                                   else "synchronous schedule")}
        except Exception as e:          # noqa: BLE001
            pipe_leg = {"error": repr(e), "schedule": "overlapped" if other_sched else "synchronous"}
        done_p.set()
    # (KGE_DIST_EAGER_LEG=1: the same engine's steps once more as eager launches - what the graph replay is measured against)
    if mode == "a2a" and os.environ.get("KGE_DIST_EAGER_LEG") == "1" and _de.coll:
        try:
            l_steps = max(20, min(K, 240))
            smp_ = _de.bench_sampler
            for _ in range(2):
                t0 = time.perf_counter()
                left = l_steps
                while left > 0:
                    n = min(smp_.n_slots, left)
                    _de.run_group(smp_.sample(n), graph=False, pipelined=os.environ.get("KGE_DIST_PIPELINE", "1") != "0")
                    left -= n
                torch.cuda.synchronize()
                te = time.perf_counter() - t0
            eager = {"us_per_step": round(1e6 * te / l_steps, 2), "steps": l_steps, "launch": "eager launches (run_group(graph=False))"}
        except Exception as e:          # noqa: BLE001
            eager = {"error": repr(e)}
    if want_other:
        done = threading.Event()

        def watchdog():
            if not done.wait(float(os.environ.get("KGE_DIST_LEG_TIMEOUT", "120"))):
                emit({"error": "the %s leg did not finish in time (watchdog)" % other})
                os._exit(0)
        th = threading.Thread(target=watchdog, daemon=True)
        th.start()
        try:
            l_steps = max(20, min(K, 240))
            if other == "p2p":
                leng, lrun, lrows, ldesc, ltabs = _p2p_setup(args, world, rank, dev, w, n_ent, d_e, d_r, emb_init)
            else:
                leng, lrun, lrows, ldesc, _ = _a2a_setup(args, world, rank, dev, w, n_ent, d_e, emb_init)
                ltabs = None
            lrun(min(20, l_steps))
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            lrun(l_steps)
            torch.cuda.synchronize(); dist.barrier()
            aw = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(aw, op=dist.ReduceOp.MAX)
            leg = {"value": round(l_steps * w["B"] * world / float(aw.item()), 1), "unit": "edges/s", "steps": l_steps,
                   "us_per_step": round(1e6 * float(aw.item()) / l_steps, 2), "desc": ldesc}
            if ltabs is not None:
                tabs = ltabs
        except Exception as e:          # noqa: BLE001
            leg = {"error": repr(e)}
        # the SAME per-GPU workload without its exchanges: every rank steps through a one-rank engine over a shard-sized table
        # (the N = 1 point of this curve, `python bench.py --gpus 1 --workload ...`, measured inside this job: bench.py's own
        # N = 1 default is configs[1], a different workload)
        try:
            if os.environ.get("KGE_DIST_LOCAL_LEG", "1") != "0":
                seng, srun, _, _, _ = _a2a_setup(args, 1, 0, dev, w, n_ent if w.get("fixed_graph") else (n_ent + world - 1) // world, d_e,
                                                 emb_init, allow_force_coll=False)
                s_steps = max(20, min(K, 240))
                srun(min(20, s_steps))
                torch.cuda.synchronize(); dist.barrier()
                t0 = time.perf_counter()
                srun(s_steps)
                torch.cuda.synchronize()
                sw = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
                dist.all_reduce(sw, op=dist.ReduceOp.MAX)
                local_leg = {"us_per_step": round(1e6 * float(sw.item()) / s_steps, 2), "steps": s_steps,
                             "edges_per_s_per_gpu": round(s_steps * w["B"] / float(sw.item()), 1),
                             "what": "the same per-GPU step on a one-rank engine over %s (no exchange), "
                                     "hipGraph of [1 sampler launch + G steps], max over ranks"
                                     % ("the whole graph's table" if w.get("fixed_graph") else "a shard-sized table")}
                del seng, srun
        except Exception as e:          # noqa: BLE001
            local_leg = {"error": repr(e)}
        done.set()
    # north_star: "edges/sec on synthetic FB15k-shaped triples reported at 1/2/4/8 GPUs" - BASELINE configs[1]'s graph through the
    # same engine at this world size (DIST_WORKLOADS["transe_l2_fb15k"]), next to its own N = 1 point; a leg of its own, under
    # its own watchdog
    if want_other and mode == "a2a" and not w.get("fixed_graph") and os.environ.get("KGE_DIST_FB15K_LEG", "1") != "0":
        done_f = threading.Event()

        def watchdog_f():
            nonlocal fb_leg
            if not done_f.wait(float(os.environ.get("KGE_DIST_LEG_TIMEOUT", "120"))):
                fb_leg = {"error": "the FB15k-shaped leg did not finish in time (watchdog)"}
                emit(leg)
                os._exit(0)
        threading.Thread(target=watchdog_f, daemon=True).start()
        try:
            fb_leg = _fb15k_leg(args, world, rank, dev)
        except Exception as e:          # noqa: BLE001
            fb_leg = {"error": repr(e)}
        done_f.set()
    # ---- diagnostics (round 6): per rank, the communicator's creation time, the owner buckets' capacity and growth
    # events, and ONE group of eager synchronous steps with a HIP event behind every phase (route / ids a2a / gather / rows a2a /
    # compute / push / apply, DistEngine.profile_phases) - so that the first run on a multi-GPU node explains itself in one shot.
    # The LAST leg (a hang here costs nothing else), behind the delivered headline and under its own watchdog; the supervisor files it under config.attempts[-1].diagnostics.
    if mode == "a2a" and os.environ.get("KGE_DIST_DIAG", "1") != "0":
        done_d = threading.Event()

        def watchdog_d():
            if not done_d.wait(float(os.environ.get("KGE_DIST_LEG_TIMEOUT", "120"))):
                emit({"error": "the diagnostics leg did not finish in time (watchdog)"})
                os._exit(0)
        threading.Thread(target=watchdog_d, daemon=True).start()
        try:
            smp_ = _de.bench_sampler
            torch.cuda.synchronize(); dist.barrier()
            ph = _de.profile_phases(smp_.sample(min(smp_.n_slots, 20)))
            mine = {"rank": rank, "communicator": type(_de.comm).__name__ if _de.coll else None,
                    "communicator_create_s": getattr(_de, "comm_create_s", None), "bucket_rows": _de.cap,
                    "bucket_growth": [list(g) for g in getattr(_de, "grown", [])] or None,
                    "message_extra_rows": getattr(_de, "cap2", 0) or None,
                    "message_extra_growth": [list(g) for g in getattr(_de, "grown_extra", [])] or None,
                    "phase_us_per_step": ph}
            alls = [None] * world
            dist.all_gather_object(alls, mine)
            diag = alls
        except Exception as e:          # noqa: BLE001 - a diagnostic must never cost the line
            diag = [{"rank": rank, "error": repr(e)}]
        done_d.set()
    line = emit(leg, now=False)
    try:
        dist.barrier()
        if tabs is not None:
            tabs.close()
        if mode != "p2p":
            _de.close()                  # the group graphs first, THEN the engine's own RCCL communicator (ncclCommDestroy waits for
            #                              every hipGraph that recorded one of its collectives - dist.RcclComm.close)
    finally:
        dist.destroy_process_group()
    if line is not None:
        # RCCL prints its version banner through C stdio, which is flushed at exit when stdout is a file: push it out first so
        # that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:       # noqa: BLE001
            pass
        if _deliver(line):
            print(line, flush=True)


def _result_line(args, w, n_ent, world, wall, K, rows, d_e, d_r, desc, mode, why, sums, other, leg, overflow):
    if True:
        # algorithmic bytes per rank-step (SURVEY 8d formula on the sampled batches)
        bytes_step = 12.0 * (rows["R_e"] * d_e + rows["B"] * d_r) + 16.0 * (rows["R_e"] + rows["B"])
        # bytes crossing xGMI per rank-step.  p2p: every traced row is read once and read-modify-written once.  a2a: fixed-size
        # buckets - ids (8 B), rows (4 D_e) and packed gradient messages (4 (2 D_e + 4)) per bucket row, + the relation all-gather
        if mode == "p2p":
            xgmi_step = 3.0 * (rows["R_e"] * d_e + rows["B"] * d_r) * 4 * (world - 1) / world
        else:
            cap = rows.get("cap") or rows["UE"]
            rel_bytes = 0 if rows.get("rel_part") else (world - 1) * rows["B"] * 4 * (d_r + 4)      # (no relation exchange under --rel_part)
            # gradient messages per bucket: packed single-trace (round 6): (cap + cap2) rows of d_e + 4 floats; else cap rows of 2 d_e + 4
            msg_bytes = ((cap + rows["cap2"]) * 4 * (d_e + 4)) if rows.get("cap2") else (cap * 4 * (2 * d_e + 4))
            xgmi_step = ((world - 1) * (cap * (8 + 4 * d_e) + msg_bytes) + rel_bytes) * 1.0
        out = {
            "metric": "positive edges/sec (whole node)",
            "value": round(K * w["B"] * world / wall, 1), "unit": "edges/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(1e3 * wall / K, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("%s synthetic FB15k-shaped (BASELINE configs[1]'s graph, NOT weak-scaled: the 14 951-entity table "
                                    "range-sharded over the ranks, %d triples per rank): n_ent=%d n_rel=%d, per-GPU batch=%d neg=%d "
                                    "hidden=%d (D_e=%d) over %d GPU%s; %s"
                                    % (w["model"], dist_triples(w, world), n_ent, w["n_rel"], w["B"], w["N"], w["hidden"], d_e, world,
                                       "" if world == 1 else "s", desc)) if w.get("fixed_graph") else
                                   ("%s synthetic Freebase-scale (BASELINE configs[4], weak-scaled: %d/8 of the 86 054 151-entity "
                                    "graph): n_ent=%d n_rel=%d, per-GPU batch=%d neg=%d hidden=%d (D_e=%d) over %d GPU%s "
                                    "(%.1f GB of entity rows per GPU); %s"
                                    % (w["model"], world, n_ent, w["n_rel"], w["B"], w["N"], w["hidden"], d_e, world,
                                       "" if world == 1 else "s", (n_ent + world - 1) // world * d_e * 4 / 1e9, desc)),
                       "global_batch": w["B"] * world,
                       "parallelism": ("shared tables over %d GPUs' HBM, peer-to-peer xGMI (Hogwild)" % world)
                       if mode == "p2p" else ("entity-shard x%d, relations replicated (RCCL all-to-all)" % world),
                       "mode": mode, "fallback_reason": why or None},
            "roofline": {"bound": "hbm", "achieved": round(bytes_step * world / (wall / K) / 1e9, 2),
                         "peak": 8000.0 * world, "unit": "GB/s",
                         "frac": round(bytes_step / (wall / K) / 1e9 / 8000.0, 5), "traffic": None,
                         "algorithmic_bytes_per_rank_step": round(bytes_step, 1),
                         "xgmi_bytes_per_rank_step": round(xgmi_step, 1),
                         "xgmi_GBps_per_gpu": round(xgmi_step / (wall / K) / 1e9, 2),
                         "xgmi_peak_GBps_per_gpu": 1071.0},
            "mean_loss": round(sums[2] / K, 6),
            # the N = 1 point of this line's curve (bench.py's own N = 1 default is configs[1]; measured inside an N > 1 job as
            # `per_gpu_step_without_exchange`)
            "n1_same_workload": "python bench.py --gpus 1 --workload %s" % dist_workload_name(args),
        }
        if mode == "a2a":
            out["config"]["relation_partition"] = bool(rows.get("rel_part"))
            out["config"]["launch"] = rows.get("launch")            # "graph": kernels + collectives of a group from one hipGraph
            out["config"]["bucket_rows"] = rows.get("cap")
            out["config"]["bucket_growth"] = rows.get("grown")
            # packed single-trace gradient messages (round 6): rows of a bucket's extra region (second messages of the rows that are in
            # both traces of a batch); null / 0 = two-trace messages
            out["config"]["message_extra_rows"] = rows.get("cap2") or None
            out["config"]["message_extra_growth"] = rows.get("grown_extra")
            out["config"]["message_floats"] = (d_e + 4) if rows.get("cap2") else (2 * d_e + 4)
            out["config"]["bucket_overflows"] = overflow
        if leg is not None:
            out[other] = leg
        return out
