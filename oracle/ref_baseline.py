"""bench.py's CPU baseline of kind "reference": the reference's OWN training loop timed on the host cores.

*** MEASUREMENT INFRASTRUCTURE ONLY - never imported by the product (dgl-ke_amd/). ***

What runs is `dglke.train_pytorch.train()` (train_pytorch.py:95-197: next(sampler) -> KEModel.forward -> loss.backward() ->
KEModel.update, with its own four timers) on the UNMODIFIED files oracle/make_ref.py staged from /root/reference into
oracle/_ref, with DGL replaced by oracle/ref_stub.py.  Like every CPU baseline of this repository the sampler is excluded on
both sides: `train_sampler` is a generator over id batches built beforehand (PosG / NegG duck types of DGL's subgraphs).
Two configurations, as the reference's README recommends for a many-core host:
  * one process, torch intra-op threads (`--num_proc 1 --num_thread T`);
  * `--num_proc P`: P single-thread trainer processes, lock-free on one shared-memory model (train.py:298-317
    `model.share_memory()` + one process per trainer).
"""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    from oracle import make_ref
    return make_ref.available()


def _import_reference():
    from oracle import ref_stub
    ref_stub.install_stubs()
    p = os.path.join(HERE, "_ref")
    if p not in sys.path:
        sys.path.insert(0, p)
    import dglke.train_pytorch as tp            # the staged reference module
    from dglke.models import KEModel
    return tp, KEModel


def make_args(w, max_step, num_proc=1):
    """the reference CLI's arguments for this workload (train.py ArgParser defaults where the workload does not say)"""
    from oracle.ref_stub import Args
    a = Args()
    a.model_name = w["model"]
    a.gpu = [-1]
    a.mix_cpu_gpu = False
    a.has_edge_importance = False
    a.strict_rel_part = a.soft_rel_part = False
    a.async_update = False
    a.lr = w["lr"]
    a.neg_deg_sample = a.neg_deg_sample_eval = a.eval_filter = False
    a.regularization_coef, a.regularization_norm = w["reg_coef"], w["reg_norm"]
    a.loss_genre = "Logsigmoid"
    a.neg_adversarial_sampling, a.adversarial_temperature = w["adv"], w["adv_temp"]
    a.pairwise, a.margin = False, 1.0
    a.num_thread, a.num_proc = 1, num_proc
    a.max_step = max_step
    a.log_interval = 1 << 30                    # (no prints inside the timed loop)
    a.force_sync_interval = -1
    a.valid, a.eval_interval = False, 1 << 30
    return a


def make_model(w, args, seed=0):
    _, KEModel = _import_reference()
    th.manual_seed(seed)
    return KEModel(args, w["model"], w["n_ent"], w["n_rel"], w["hidden"], w["gamma"],
                   double_entity_emb=w["de"], double_relation_emb=w["dr"])


def graphs_of(plans, w):
    """(PosG, NegG) pairs of host-built id plans (dglke_amd.plan.build_plan dicts or oracle.synth_batch-style dicts)"""
    from oracle.ref_stub import NegG, PosG
    out = []
    for p in plans:
        r = p["rel_ids"] if "rel_ids" in p else p["r"]
        neg = p["neg_ids"] if "neg_ids" in p else p["neg"]
        C = w["B"] // w["N"]
        out.append((lambda p=p, r=r, neg=neg: (
            PosG(th.from_numpy(np.asarray(p["nid"], np.int64)), th.from_numpy(np.asarray(p["h_local"], np.int64)),
                 th.from_numpy(np.asarray(p["t_local"], np.int64)), th.from_numpy(np.asarray(r, np.int64))),
            NegG(th.from_numpy(np.asarray(neg, np.int64)), C, w["N"], w["N"], bool(p["neg_head"])))))
    return out


def _sampler(makers):
    k = 0
    while True:                                 # fresh graph objects per step: forward() writes into them
        yield makers[k % len(makers)]()
        k += 1


def run_train(model, args, makers, rank=0):
    """the reference's train() for args.max_step steps; returns the wall time of the call"""
    tp, _ = _import_reference()
    sink = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sink):      # ('proc 0 takes ... seconds')
        tp.train(args, model, _sampler(makers), rank=rank)
    return time.perf_counter() - t0


def single(w, plans, budget_s=12.0, probe_steps=8):
    """one process: the intra-op thread count that is fastest on this host, then as many steps as fit the budget"""
    nthreads = th.get_num_threads()
    makers = graphs_of(plans, w)
    a1 = make_args(w, 2)
    model = make_model(w, a1)
    run_train(model, a1, makers)                # warm-up
    tried = {}
    for nt in sorted({1, 4, 8, 16, 32, 64, nthreads}):
        if nt > nthreads:
            continue
        th.set_num_threads(nt)
        ap = make_args(w, probe_steps)
        tried[nt] = round(probe_steps * w["B"] / run_train(model, ap, makers), 1)
    best = max(tried, key=tried.get)
    th.set_num_threads(best)
    steps = int(max(8, min(400, budget_s * tried[best] / w["B"])))
    dt = run_train(model, make_args(w, steps), makers)
    th.set_num_threads(nthreads)
    return {"value": round(steps * w["B"] / dt, 1), "threads": best, "steps": steps, "seconds": round(dt, 2),
            "edges_per_s_by_threads": tried}


def _hog_worker(rank, tables, w, steps, procs, ready, go, out):
    th.set_num_threads(1)
    from oracle import kge_oracle as O
    # (spawned, not forked: the parent has run OpenMP regions.  The trainer builds its KEModel and then points its two embedding
    #  tables at the parent's shared-memory tensors - what fork + KEModel.share_memory() gives the reference's trainer processes)
    model = make_model(w, make_args(w, 1, procs))
    model.entity_emb.emb, model.entity_emb.state_sum, model.relation_emb.emb, model.relation_emb.state_sum = tables
    rng = np.random.RandomState(1000 + rank)
    plans = [O.synth_batch(rng, w["n_ent"], w["n_rel"], w["B"], w["N"], w["N"], s) for s in range(1, 13)]
    makers = graphs_of(plans, w)
    run_train(model, make_args(w, 1), makers, rank)
    ready.put(rank)
    go.wait()
    t0 = time.perf_counter()
    run_train(model, make_args(w, steps), makers, rank)
    out.put((rank, steps, time.perf_counter() - t0))


def hogwild(w, procs, seconds=5.0, rate_1=None, timeout=180.0):
    """`--num_proc procs`: single-thread trainer processes on ONE shared-memory model (KEModel.share_memory(), train.py:298-317).
    train() runs a fixed number of steps, so the per-process step count is sized from the single-thread rate for ~`seconds` of
    work; returns (aggregate edges/s = all steps x B / the slowest process's wall, total steps)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    args = make_args(w, 1, procs)
    model = make_model(w, args)
    model.share_memory()
    tables = (model.entity_emb.emb, model.entity_emb.state_sum, model.relation_emb.emb, model.relation_emb.state_sum)
    # under contention a process runs slower than alone: half the stand-alone rate is a fair first guess for the step count
    steps = int(max(4, min(2000, seconds * (rate_1 or 60000.0) / w["B"] * 0.5)))
    ready, out, go = ctx.Queue(), ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=_hog_worker, args=(r, tables, w, steps, procs, ready, go, out), daemon=True) for r in range(procs)]
    for p in ps:
        p.start()
    def take(q, limit):                          # a queue item - or an error as soon as a trainer has died (not after the timeout)
        import queue
        t_end = time.time() + limit
        while True:
            try:
                return q.get(timeout=1.0)
            except queue.Empty:
                dead = [p_.exitcode for p_ in ps if p_.exitcode not in (None, 0)]
                if dead or time.time() > t_end:
                    raise RuntimeError("reference --num_proc leg: trainer exit codes %r / timeout" % (dead,))
    try:
        for _ in range(procs):
            take(ready, timeout)
        go.set()
        res = [take(out, 8 * seconds + timeout) for _ in range(procs)]
    finally:
        for p in ps:
            p.join(timeout=5)
            if p.is_alive():
                p.kill()                        # our own children, by handle
    wall = max(dt for _, _, dt in res)
    total = sum(n for _, n, _ in res)
    return total * w["B"] / wall, total, wall
