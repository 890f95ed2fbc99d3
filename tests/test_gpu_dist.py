"""The range-sharded multi-GPU step (dglke_amd/dist.py) with its DEVICE arithmetic running for real:
  * unit tests of the routing kernels (kge_route_build, kge_gather_rows_req, kge_adagrad_apply_merged) against numpy;
  * DistEngine + HipOps at world = 2: two processes on one GPU, the fixed-size messages carried by gloo
    (tests/dist_worker.py) - against the fp64 oracle's statement of the synchronous sharded step, against the fused
    single-table step (batches on disjoint supports), and the pipelined step against the one-step-stale statement."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import kge_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def test_route_build_matches_numpy():
    """owner buckets of fixed capacity, cache rows, re-addressed batch - host plan and device-built plan."""
    import ctypes as C
    from dglke_amd import _lib, plan
    from dglke_amd import dist as kd
    rng = np.random.RandomState(3)
    for world, n_ent, B, N, cap in ((1, 500, 64, 16, None), (2, 501, 64, 16, 96), (3, 1000, 96, 32, 80), (8, 5000, 128, 32, 48),
                                    (4, 400, 128, 32, 20)):            # the last one overflows its buckets on purpose
        bt = O.synth_batch(rng, n_ent, 7, B, N, N, 1)
        b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"], DEV)
        per = (n_ent + world - 1) // world
        cap = cap or b.UE
        eng_like = type("S", (), {})()
        z = lambda n, dt: torch.zeros(n, dtype=dt, device=DEV)
        eng_like.req_ids, eng_like.h_loc, eng_like.t_loc, eng_like.neg_loc = z(world * cap, torch.int64), z(B, torch.int64), z(B, torch.int64), z(b.C * N, torch.int64)
        eng_like.ue_loc, eng_like.ue_rec_loc, eng_like.overflow = z(b.UE, torch.int64), z(b.UE * 8, torch.int32), z(1, torch.int32)
        eng_like.req_ids.fill_(12345)                              # pads must be WRITTEN
        lb = kd.HipOps().route(b, world, per, cap, eng_like)
        torch.cuda.synchronize()
        ue = b.p["ue_id"]
        owner = ue // per
        start = np.searchsorted(owner, np.arange(world + 1))
        pos = np.arange(len(ue)) - start[owner]
        fits = pos < cap
        cr = np.where(fits, owner * cap + pos, world * cap)
        req = np.full(world * cap, -1, np.int64)
        req[cr[fits]] = ue[fits]
        assert int(eng_like.overflow.item()) == int((~fits).sum())
        assert np.array_equal(eng_like.req_ids.cpu().numpy(), req)
        assert np.array_equal(eng_like.ue_loc.cpu().numpy()[:len(ue)], cr)
        row_of = dict(zip(ue.tolist(), cr.tolist()))
        assert np.array_equal(eng_like.h_loc.cpu().numpy(), [row_of[x] for x in b.p["h_gid"].tolist()])
        assert np.array_equal(eng_like.t_loc.cpu().numpy(), [row_of[x] for x in b.p["t_gid"].tolist()])
        assert np.array_equal(eng_like.neg_loc.cpu().numpy(), [row_of[x] for x in b.p["neg_ids"].tolist()])
        rec = eng_like.ue_rec_loc.cpu().numpy().reshape(-1, 8)[:len(ue)]
        want = b.p["ue_rec"].reshape(-1, 8).copy()
        want[:, 0], want[:, 1] = (cr & 0xffffffff).astype(np.uint32).view(np.int32), (cr >> 32).astype(np.int32)
        assert np.array_equal(rec, want)
        assert lb.c.B == B and lb.c.UE == b.UE


def test_route_fill_and_capacity_growth():
    """kge_route_fill (largest owner-bucket fill of a GROUP of batches: host-built plans one by one, consecutive sampler slots in
    one launch) against numpy, and DistEngine.ensure_capacity on top of it: heavy-tailed ids whose hubs sit in the first shard,
    a deliberately small capacity - the buckets grow BEFORE the group runs and the overflow counter of the routing kernel stays 0
    (the exchange itself is covered by the world-2 tests below, the decision across ranks by tests/test_dist_gloo.py at world 8)."""
    from dglke_amd import plan
    from dglke_amd import dist as kd
    from dglke_amd.dataloader import DeviceSampler
    rng = np.random.RandomState(9)
    ops = kd.HipOps()
    world, n_ent, B, N = 8, 4000, 128, 32
    per = (n_ent + world - 1) // world
    host = []
    for k in range(5):
        bt = O.synth_batch(rng, n_ent, 7, B, N, N, k + 1)
        skew = lambda x: np.where(rng.rand(len(x)) < 0.7, x % per, x)          # 70 % of the ids fall into shard 0
        host.append(plan.make_batch(skew(bt["h"]), skew(bt["t"]), bt["r"], skew(bt["neg"]), N, N, bt["neg_head"], DEV))
    out = torch.zeros(2, dtype=torch.int32, device=DEV)      # ABI 8: {largest bucket fill, most both-trace entries of one bucket}
    ops.route_fill(host, world, per, out)
    want = max(int(np.bincount(np.minimum(b.p["ue_id"] // per, world - 1), minlength=world).max()) for b in host)

    def both_max(ue, rec):
        both = (rec[:, 3] > rec[:, 2]) & (rec[:, 5] > rec[:, 4])
        return int(np.bincount(np.minimum(ue // per, world - 1)[both], minlength=world).max())
    want2 = max(both_max(b.p["ue_id"], b.p["ue_rec"].reshape(-1, 8)) for b in host)
    assert out.tolist() == [want, want2] and want > 150 and want2 > 0
    # consecutive sampler slots: ONE launch
    h = rng.randint(0, n_ent, 20000); t = rng.randint(0, n_ent, 20000); r = rng.randint(0, 7, 20000)
    h = np.where(rng.rand(20000) < 0.6, h % per, h)
    smp = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=6, seed=3)
    dbs = smp.sample()
    out.zero_()
    ops.route_fill(dbs, world, per, out)
    torch.cuda.synchronize()
    fills = []
    for k in range(6):
        a = smp.slot_arrays(k)
        ue = a["ue_id"][:a["counts"][0]]
        fills.append(int(np.bincount(np.minimum(ue // per, world - 1), minlength=world).max()))
    assert int(out[0].item()) == max(fills)
    # ensure_capacity at world 1 is a no-op (one owner: the bound); the growth rule itself on a world-8 spec without collectives
    class _NoComm(object):
        world, rank = 8, 0
    eng = type("E", (), {})()
    eng.lr, eng.rel = 0.1, torch.zeros(7, 16, device=DEV)
    spec = kd.ShardSpec(n_ent, 8, 0)
    de = kd.DistEngine(eng, spec, torch.zeros(spec.n_local, 16, device=DEV), torch.zeros(spec.n_local, device=DEV), ops=ops,
                       comm=_NoComm(), cap=64)
    de.coll = False                                   # (no peers in this process: only the local measurement)
    logs = []
    cap = de.ensure_capacity(dbs, log=logs.append)
    assert cap >= max(fills) > 64 and cap % 64 == 0 and logs and de.grown[0][0] == 64
    assert de.ensure_capacity(dbs) == cap and len(de.grown) == 1         # nothing to grow the second time
    for b in dbs:                                     # every entry now fits its bucket
        de.ops.route(b, 8, spec.shard, de.cap, de.slots[0])
    torch.cuda.synchronize()
    assert de.check_overflow() == 0
    # kge_route_build_group (round 4): the whole group routed in ONE launch into a pool row per sampler slot - the same arrays as
    # routing the batches one by one
    de.prepare_group(dbs)
    torch.cuda.synchronize()
    off, stride = ops.route_layout(dbs[0], 8, de.cap)
    pool = de._route_pool.cpu().numpy()
    for b in dbs:
        lb = de._routed_ahead(b)
        s0 = de.slots[0]
        de.ops.route(b, 8, spec.shard, de.cap, s0)
        torch.cuda.synchronize()
        row = pool[b.slot * stride:(b.slot + 1) * stride]
        cnt = int(smp.slot_arrays(b.slot)["counts"][0])
        for name, ref, n, dt in (("req_ids", s0.req_ids, 8 * de.cap, np.int64), ("h_loc", s0.h_loc, B, np.int64),
                                 ("t_loc", s0.t_loc, B, np.int64), ("neg_loc", s0.neg_loc, dbs[0].C * N, np.int64),
                                 ("ue_loc", s0.ue_loc, cnt, np.int64), ("ue_rec_loc", s0.ue_rec_loc, 8 * cnt, np.int32)):
            got = row[off[name]:off[name] + n * np.dtype(dt).itemsize].view(dt)
            assert np.array_equal(got, ref.cpu().numpy()[:n]), "%s of slot %d differs between group and single routing" % (name, b.slot)
        assert np.array_equal(lb.req_ids.cpu().numpy(), s0.req_ids.cpu().numpy())
    assert de.check_overflow() == 0


def test_precaptured_compute_graphs_equal_eager_launches():
    """the eager sharded step with its kernels between pull and push replayed from small hipGraphs (DistEngine.precapture:
    one per sampler slot, corruption mode and cache slot; the collectives stay eager) against the same steps launched kernel by
    kernel: bit-identical shard, state and relation table; world 1 through the collective code path (RCCL calls with one rank,
    pipelined pull on the side stream) on device-sampled batches routed by prepare_group."""
    import torch.distributed as dist
    from dglke_amd import dist as kd
    from dglke_amd.dataloader import DeviceSampler
    from dglke_amd.engine import StepEngine
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        n_ent, n_rel, hidden, B, N = 6000, 40, 64, 128, 32
        rng = np.random.RandomState(31)
        h, r, t = rng.randint(0, n_ent, 30000), rng.randint(0, n_rel, 30000), rng.randint(0, n_ent, 30000)
        res = []
        for graphs in (False, True):
            torch.manual_seed(3)
            eng = StepEngine("RotatE", 1, n_rel, hidden, 12.0, 0.05, DEV, True, False, True, 1.0, 1e-6, 3)
            ent = torch.empty(n_ent, 2 * hidden, device=DEV).uniform_(-0.2, 0.2)
            state = torch.zeros(n_ent, device=DEV)
            smp = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=6, seed=5)
            comm = kd.RcclComm()
            de = kd.DistEngine(eng, kd.ShardSpec(n_ent, 1, 0), ent, state, always_collective=True, comm=comm)
            de._cg_on = graphs
            for grp in range(5):                     # 5 groups: sizes 6, 5 (odd: the corruption mode of every slot flips), 6, 6, 3
                dbs = smp.sample((6, 5, 6, 6, 3)[grp])
                de.prepare_group(dbs)
                if grp == 0:
                    n = de.precapture(smp)
                    assert n == (6 * 2 * 2 if graphs else 0)
                for k, b in enumerate(dbs):
                    de.step_pipelined(b, dbs[k + 1] if k + 1 < len(dbs) else None)
            torch.cuda.synchronize()
            assert de.check_overflow() == 0
            res.append((ent.cpu(), state.cpu(), eng.rel.cpu().clone(), eng.rel_state.cpu().clone()))
            comm.close()
        for x, y in zip(res[0], res[1]):
            assert torch.equal(x, y)
        assert float((res[0][1] > 0).sum()) > 100
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pipelined", [False, True], ids=["synchronous", "pipelined_pull"])
def test_group_graph_with_recorded_collectives_equals_eager_launches(pipelined):
    """DistEngine.run_group replays [routing of the group + ONE id exchange + the group's steps WITH their RCCL
    collectives] from one hipGraph (RcclComm at world 1 through the collective code path: ncclAllToAll / grouped push recorded
    into the graph).  Against the same groups as eager launches: shard, state and relation table bit-identical; group sizes
    6, 6, 5 (odd: the corruption mode of every slot flips -> another graph), 5, 6, 6; and the teardown order that rounds 2-4
    mistook for a replay hang - the graphs die before ncclCommDestroy (DistEngine.close)."""
    from dglke_amd import dist as kd
    from dglke_amd.dataloader import DeviceSampler
    from dglke_amd.engine import StepEngine
    n_ent, n_rel, hidden, B, N = 6000, 40, 64, 128, 32
    rng = np.random.RandomState(31)
    h, r, t = rng.randint(0, n_ent, 30000), rng.randint(0, n_rel, 30000), rng.randint(0, n_ent, 30000)
    res = []
    for graph in (False, True):
        torch.manual_seed(3)
        eng = StepEngine("RotatE", 1, n_rel, hidden, 12.0, 0.05, DEV, True, False, True, 1.0, 1e-6, 3)
        ent = torch.empty(n_ent, 2 * hidden, device=DEV).uniform_(-0.2, 0.2)
        state = torch.zeros(n_ent, device=DEV)
        smp = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=6, seed=5)
        de = kd.DistEngine(eng, kd.ShardSpec(n_ent, 1, 0), ent, state, always_collective=True, comm=kd.RcclComm())
        replayed = [de.run_group(smp.sample(n), graph=graph, pipelined=pipelined) for n in (6, 6, 5, 5, 6, 6, 6)]
        torch.cuda.synchronize()
        assert de.check_overflow() == 0
        # first group of a (size, parity) geometry: eager + recorded; 6/even, 6/even -> replay, 5/even, 5/odd, 6/even(after 2 x 5) ...
        assert replayed == ([False, True, False, False, True, True, True] if graph else [False] * 7), replayed
        res.append((ent.cpu(), state.cpu(), eng.rel.cpu().clone(), eng.rel_state.cpu().clone()))
        de.close()                                   # returns: the graphs are destroyed before the communicator
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x, y)
    assert float((res[0][1] > 0).sum()) > 100


def test_overlapped_schedule_equals_pipelined_pull():
    """DistEngine._steps_overlapped (push, owner-side apply and pull on the side stream; the compute stream runs compute + the
    relation half back to back) is the SAME dataflow as step_pipelined - entity rows of step s+1 gathered after update s-1 and
    before update s, relations never stale - so shard, state and relation table must come out bit-identical: as eager launches and
    replayed from the group graphs, with the relation all-gather and under relation-local updates; world 1 through the collective
    code path (RcclComm), group sizes 6, 6, 5, 5, 1, 6, 2."""
    from dglke_amd import dist as kd
    from dglke_amd.dataloader import DeviceSampler
    from dglke_amd.engine import StepEngine
    n_ent, n_rel, hidden, B, N = 6000, 40, 64, 128, 32
    rng = np.random.RandomState(31)
    h, r, t = rng.randint(0, n_ent, 30000), rng.randint(0, n_rel, 30000), rng.randint(0, n_ent, 30000)
    for rel_local in (False, True):
        res = []
        for sched, graph in ((True, False), ("overlap", False), ("overlap", True)):
            torch.manual_seed(3)
            eng = StepEngine("RotatE", 1, n_rel, hidden, 12.0, 0.05, DEV, True, False, True, 1.0, 1e-6, 3)
            ent = torch.empty(n_ent, 2 * hidden, device=DEV).uniform_(-0.2, 0.2)
            state = torch.zeros(n_ent, device=DEV)
            smp = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=6, seed=5)
            de = kd.DistEngine(eng, kd.ShardSpec(n_ent, 1, 0), ent, state, always_collective=True, comm=kd.RcclComm(),
                               rel_local=rel_local)
            replayed = [de.run_group(smp.sample(n), graph=graph, pipelined=sched) for n in (6, 6, 5, 5, 1, 6, 2)]
            torch.cuda.synchronize()
            assert de.check_overflow() == 0
            assert any(replayed) == graph
            res.append((ent.cpu(), state.cpu(), eng.rel.cpu().clone(), eng.rel_state.cpu().clone()))
            de.close()
        for other in res[1:]:
            for x, y in zip(res[0], other):
                assert torch.equal(x, y)
        assert float((res[0][1] > 0).sum()) > 100


def test_gather_rows_req_skips_pads_and_foreign_ids():
    from dglke_amd import dist as kd
    t = torch.arange(0, 80, dtype=torch.float32, device=DEV).reshape(10, 8)
    ids = torch.tensor([105, -1, 100, 109, 99, 110], device=DEV)          # shard holds ids 100..109
    out = torch.full((6, 8), -7.0, device=DEV)
    kd.HipOps().gather_req(t, ids, 100, out)
    torch.cuda.synchronize()
    want = torch.full((6, 8), -7.0)
    want[0], want[2], want[3] = t[5].cpu(), t[0].cpu(), t[9].cpu()
    assert torch.equal(out.cpu(), want)


@pytest.mark.parametrize("nsrc,cap,dim,n_rows", [(1, 40, 8, 64), (2, 33, 8, 40), (3, 100, 64, 120), (8, 257, 400, 600),
                                                 (5, 7, 1024, 12), (16, 64, 16, 200), (64, 9, 8, 30)])
def test_apply_merged_equals_sequential_apply(nsrc, cap, dim, n_rows):
    """rows that arrive from several sources are applied once, in source order, by the first source's wavefront: compare with a
    sequential numpy statement (kvserver.py:41-51 per pushed row, tensor_models.py:316 trace order)."""
    from dglke_amd import dist as kd
    rng = np.random.RandomState(nsrc * 1000 + cap)
    lo, T, lr = 1000, 2, 0.1
    ld = T * dim + 4
    table = rng.randn(n_rows, dim).astype(np.float32)
    state = rng.rand(n_rows).astype(np.float32)
    ids = np.full((nsrc, cap), -1, np.int64)
    msg = rng.randn(nsrc * cap, ld).astype(np.float32) * 0.05
    msg[:, T * dim:T * dim + T] = rng.rand(nsrc * cap, T).astype(np.float32) * 0.01
    for s in range(nsrc):
        k = rng.randint(0, min(cap, n_rows) + 1)
        ids[s, :k] = np.sort(rng.choice(n_rows, k, replace=False)) + lo
    # a few messages with one silent trace (increment 0: that trace is skipped)
    sil = rng.rand(nsrc * cap) < 0.2
    msg[sil, T * dim] = 0.0
    t_d, s_d = torch.from_numpy(table).to(DEV), torch.from_numpy(state).to(DEV)
    kd.HipOps().apply_merged(t_d, s_d, nsrc, cap, torch.from_numpy(ids.reshape(-1)).to(DEV), lo, torch.from_numpy(msg).to(DEV), T, lr)
    torch.cuda.synchronize()
    t64, s64 = table.astype(np.float64), state.astype(np.float64)
    for s in range(nsrc):
        for p in range(cap):
            i = ids[s, p]
            if i < 0:
                continue
            m = msg[s * cap + p].astype(np.float64)
            for t in range(T):
                inc = m[T * dim + t]
                if inc == 0.0:
                    continue
                s64[i - lo] += inc
                t64[i - lo] += -lr * m[t * dim:(t + 1) * dim] / (np.sqrt(s64[i - lo]) + 1e-10)
    np.testing.assert_allclose(s_d.cpu().numpy(), s64, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(t_d.cpu().numpy(), t64, rtol=1e-5, atol=2e-6)
    # ids inside the messages (the relation all-gather's format)
    t2, s2 = torch.from_numpy(table).to(DEV), torch.from_numpy(state).to(DEV)
    m2 = msg.copy()
    m2.view(np.int32)[:, T * dim + T] = ((ids.reshape(-1)) & 0xffffffff).astype(np.uint32).view(np.int32)
    m2.view(np.int32)[:, T * dim + T + 1] = (ids.reshape(-1) >> 32).astype(np.int32)
    kd.HipOps().apply_merged(t2, s2, nsrc, cap, None, lo, torch.from_numpy(m2).to(DEV), T, lr)
    torch.cuda.synchronize()
    assert torch.equal(t2, t_d) and torch.equal(s2, s_d)
    # round 4: the step's two applies as ONE launch (kge_adagrad_apply_merged_pair: this job next to a second, narrower,
    # single-trace job with the ids inside its messages - the relation replica's) - bit-identical to the two separate calls
    if dim % 4 == 0:
        d2 = max(4, dim // 2 // 4 * 4)
        tab_b = rng.randn(37, d2).astype(np.float32)
        st_b = rng.rand(37).astype(np.float32)
        ids_b = np.full((nsrc, 8), -1, np.int64)
        for s in range(nsrc):
            k = rng.randint(1, 9)
            ids_b[s, :k] = np.sort(rng.choice(37, k, replace=False))
        msg_b = (rng.randn(nsrc * 8, d2 + 4) * 0.05).astype(np.float32)
        msg_b[:, d2] = rng.rand(nsrc * 8).astype(np.float32) * 0.01
        msg_b.view(np.int32)[:, d2 + 1] = (ids_b.reshape(-1) & 0xffffffff).astype(np.uint32).view(np.int32)
        msg_b.view(np.int32)[:, d2 + 2] = (ids_b.reshape(-1) >> 32).astype(np.int32)
        ops = kd.HipOps()
        ta, sa = torch.from_numpy(table).to(DEV), torch.from_numpy(state).to(DEV)
        tb1, sb1 = torch.from_numpy(tab_b).to(DEV), torch.from_numpy(st_b).to(DEV)
        tb2, sb2 = tb1.clone(), sb1.clone()
        ids_d, msg_d, msgb_d = torch.from_numpy(ids.reshape(-1)).to(DEV), torch.from_numpy(msg).to(DEV), torch.from_numpy(msg_b).to(DEV)
        ops.apply_merged(tb1, sb1, nsrc, 8, None, 0, msgb_d, 1, lr)
        ops.apply_merged_pair((ta, sa, nsrc, cap, ids_d, lo, msg_d, T), (tb2, sb2, nsrc, 8, None, 0, msgb_d, 1), lr)
        torch.cuda.synchronize()
        assert torch.equal(ta, t_d) and torch.equal(sa, s_d) and torch.equal(tb2, tb1) and torch.equal(sb2, sb1)
        assert not torch.equal(tb1.cpu(), torch.from_numpy(tab_b))


@pytest.mark.parametrize("mode", ["sampled", "sampled_overlap"])
def test_world8_on_one_device_matches_the_oracle_statement(tmp_path, mode):
    """round 6: BASELINE's world size - eight ranks (eight processes sharing this GPU, messages through gloo): owner buckets for eight
    shards, packed single-trace messages with their extra regions, the merged apply with eight sources, device-sampled groups routed
    ahead with one id exchange per group; synchronous and with every exchange on the side stream - against the fp64 statement."""
    import dist_worker as W
    world = 8
    z = _run_workers(tmp_path, mode, world=world)
    for model, de_, dr_ in W.MODELS:
        ent, es, rel, rs = _oracle_statement(model, de_, dr_, z, world, mode)
        lr = W.LR
        np.testing.assert_allclose(z[model + "_state"], es, rtol=2e-3, atol=1e-9, err_msg=model + " entity state")
        np.testing.assert_allclose(z[model + "_ent"], ent, rtol=1e-4, atol=5e-3 * lr, err_msg=model + " entity rows")
        np.testing.assert_allclose(z[model + "_relstate0"], rs, rtol=2e-3, atol=1e-9, err_msg=model + " relation state")
        np.testing.assert_allclose(z[model + "_rel0"], rel, rtol=1e-4, atol=5e-3 * lr, err_msg=model + " relation rows")
        for r in range(1, world):
            assert np.array_equal(z[model + "_rel0"], z[model + "_rel%d" % r]), "relation replicas differ"


def _run_workers(tmp_path, mode, world=2, transport="host"):
    port = str(29700 + 13 * (world // 8) + os.getpid() % 250 + {"random": 0, "disjoint": 1, "pipelined": 2, "nd": 6, "relpart": 7, "sampled": 8, "sampled_pipelined": 9, "overlap": 10, "sampled_overlap": 11}[mode] + (3 if transport == "rccl" else 0))
    env = dict(os.environ)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world), port, str(tmp_path), mode,
                               transport], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "worker failed:\n" + "\n----\n".join(outs)
    return np.load(os.path.join(str(tmp_path), "result.npz"))


def _oracle_statement(model, de_, dr_, z, world, mode):
    """fp64 statement of the sharded step: every rank's gradients from the SAME tables, applied owner-side in rank order (trace 0
    then trace 1 per rank), relations in rank order.  mode 'pipelined': the ENTITY rows of step s were pulled before update s-1
    landed (exact one-step staleness); the replicated relation table is always current."""
    import dist_worker as W
    cfg = O.Config(model, 12.0, W.HID, W.LR, adv=True, adv_temp=1.0, reg_coef=1e-6, reg_norm=3, double_ent=de_, double_rel=dr_,
                   neg_deg=(mode == "nd"))
    ent, rel = z[model + "_init_ent"].astype(np.float64), z[model + "_init_rel"].astype(np.float64)
    es, rs = np.zeros(W.N_ENT), np.zeros(W.N_REL)
    if mode.startswith("sampled"):         # the ids the device sampler drew (read back by the workers), groups of two steps
        d = {k: z["%s_drawn_%s" % (model, k)] for k in ("h", "t", "r", "neg", "neg_head")}
        bts = []
        for s in range(d["h"].shape[0]):
            row = []
            for k in range(world):
                nid, inv = np.unique(np.concatenate([d["h"][s, k], d["t"][s, k]]), return_inverse=True)
                row.append(dict(nid=nid, h_local=inv[:W.B], t_local=inv[W.B:], r=d["r"][s, k], neg=d["neg"][s, k],
                                neg_head=bool(d["neg_head"][s, k])))
            bts.append(row)
    else:
        bts = W.batches(world, W.STEPS, "relpart" if mode == "relpart" else "random")
    pulled = ent.copy()                    # what the pull of the current step saw
    for s, row in enumerate(bts):
        # (sampled_pipelined: the first step of every group of two pulls for itself, behind its predecessor's update)
        src = pulled if mode in ("pipelined", "overlap") or (mode in ("sampled_pipelined", "sampled_overlap") and s % 2 == 1) else ent
        outs = [O.forward_backward(cfg, src, rel, bt["nid"], bt["h_local"], bt["t_local"], bt["r"], bt["neg"], bt["neg_head"],
                                   W.N, W.N) for bt in row]
        pulled = ent.copy()                # the pull of step s+1 runs now: after update s-1, before update s
        for bt, out in zip(row, outs):
            O.adagrad_update(ent, es, bt["nid"], out["g_pos_ent"], W.LR)
            O.adagrad_update(ent, es, bt["neg"], out["g_neg"], W.LR)
        for bt, out in zip(row, outs):
            O.adagrad_update(rel, rs, bt["r"], out["g_rel"], W.LR)
    return ent, es, rel, rs


@pytest.mark.parametrize("mode,transport", [("random", "host"), ("pipelined", "host"), ("nd", "host"), ("relpart", "host"),
                                            ("sampled", "host"), ("sampled_pipelined", "host"), ("overlap", "host"),
                                            ("sampled_overlap", "host"), ("overlap", "rccl"),
                                            ("random", "rccl"), ("pipelined", "rccl"), ("relpart", "rccl"), ("sampled", "rccl")])
def test_world2_hip_ops_match_the_oracle_statement(tmp_path, mode, transport):
    """`host`: two processes on ONE device, messages staged through gloo.  `rccl`: two processes on TWO devices, the product's
    transport - dist.RcclComm (ncclAllToAll / ncclAllGather on the step's streams, grouped push, side-stream pull) - against
    the same fp64 statement: what a multi-GPU box checks before it benches (skipped when there is one GPU)."""
    import dist_worker as W
    if transport == "rccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    z = _run_workers(tmp_path, mode, transport=transport)
    for model, de_, dr_ in W.MODELS:
        ent, es, rel, rs = _oracle_statement(model, de_, dr_, z, 2, mode)
        lr = W.LR
        assert np.abs(z[model + "_ent"] - z[model + "_init_ent"]).max() > 1e-3, "training did not move the table"
        np.testing.assert_allclose(z[model + "_state"], es, rtol=2e-3, atol=1e-9, err_msg=model + " entity state")
        np.testing.assert_allclose(z[model + "_ent"], ent, rtol=1e-4, atol=5e-3 * lr, err_msg=model + " entity rows")
        for r in range(1 if mode == "relpart" else 2):
            np.testing.assert_allclose(z[model + "_relstate%d" % r], rs, rtol=2e-3, atol=1e-9, err_msg=model + " relation state")
            np.testing.assert_allclose(z[model + "_rel%d" % r], rel, rtol=1e-4, atol=5e-3 * lr, err_msg=model + " relation rows")
        if mode == "relpart":
            # relation partitioning: no relation exchange - rank 1's replica holds current rows for ITS relations only; rank 0's was
            # completed from the owners (checked against the statement above)
            own1 = np.arange(W.N_REL) % 2 == 1
            assert np.array_equal(z[model + "_rel0"][own1], z[model + "_rel1"][own1])
            continue
        assert np.array_equal(z[model + "_rel0"], z[model + "_rel1"]), "relation replicas differ"
        assert np.array_equal(z[model + "_relstate0"], z[model + "_relstate1"])


@pytest.mark.parametrize("mode", ["pipelined", "relpart", "sampled", "sampled_pipelined", "overlap", "sampled_overlap"])
def test_world4_on_one_device_matches_the_oracle_statement(tmp_path, mode):
    """four ranks (four processes sharing this GPU, messages through gloo): owner buckets for four shards, the merged apply with
    four sources and rows that arrive from several ranks at once, the one-step-stale pipeline / relation partitioning at a
    world size beyond two - against the fp64 statement of the same schedule."""
    import dist_worker as W
    world = 4
    z = _run_workers(tmp_path, mode, world=world)
    for model, de_, dr_ in W.MODELS:
        ent, es, rel, rs = _oracle_statement(model, de_, dr_, z, world, mode)
        lr = W.LR
        np.testing.assert_allclose(z[model + "_state"], es, rtol=2e-3, atol=1e-9, err_msg=model + " entity state")
        np.testing.assert_allclose(z[model + "_ent"], ent, rtol=1e-4, atol=5e-3 * lr, err_msg=model + " entity rows")
        np.testing.assert_allclose(z[model + "_relstate0"], rs, rtol=2e-3, atol=1e-9, err_msg=model + " relation state")
        np.testing.assert_allclose(z[model + "_rel0"], rel, rtol=1e-4, atol=5e-3 * lr, err_msg=model + " relation rows")
        if mode != "relpart":
            for r in range(1, world):
                assert np.array_equal(z[model + "_rel0"], z[model + "_rel%d" % r]), "relation replicas differ"


def test_world2_equals_single_table_step_on_disjoint_batches(tmp_path):
    """rank k's batches only touch rank-k entities / relations: the synchronous sharded step is then the fused single-GPU step
    applied rank after rank on one table (same kernels; the gradient-emitting update + owner-side apply is a different
    instantiation of the update code, hence a few ulp, see test_sharded_engine_world1_equals_fused_step)."""
    import dist_worker as W
    z = _run_workers(tmp_path, "disjoint")
    for model, _, _ in W.MODELS:
        np.testing.assert_allclose(z[model + "_ent"], z[model + "_ref_ent"], rtol=1e-5, atol=5e-6, err_msg=model)
        np.testing.assert_allclose(z[model + "_state"], z[model + "_ref_state"], rtol=1e-5, atol=1e-8, err_msg=model)
        np.testing.assert_allclose(z[model + "_rel0"], z[model + "_ref_rel"], rtol=1e-5, atol=5e-6, err_msg=model)
        np.testing.assert_allclose(z[model + "_relstate0"], z[model + "_ref_rel_state"], rtol=1e-5, atol=1e-8, err_msg=model)


@pytest.mark.parametrize("model", ["RotatE", "TransE_l2"])
def test_world1_local_shortcut_is_the_single_table_step(model):
    """round 6: the all-to-all engine at world 1 without collectives runs the in-place step on its shard (DistEngine.local_only - no
    routing, no row cache, no gradient messages, no apply launch).  Bit-identical to StepEngine on the same table, eagerly and
    through run_group's group graphs; the route -> gather -> messages -> apply path (KGE_DIST_LOCAL_SHORTCUT=0: what every rank runs at
    world > 1) agrees with it within the row tolerance of the sharded tests."""
    from dglke_amd import dist as kd
    from dglke_amd.dataloader import DeviceSampler
    from dglke_amd.engine import StepEngine
    n_ent, n_rel, hidden, B, N = 6000, 40, 64, 128, 32
    de_flag = model == "RotatE"
    rng = np.random.RandomState(41)
    h, r, t = rng.randint(0, n_ent, 30000), rng.randint(0, n_rel, 30000), rng.randint(0, n_ent, 30000)
    d_e = hidden * (2 if de_flag else 1)
    torch.manual_seed(11)
    ent0 = torch.empty(n_ent, d_e, device=DEV).uniform_(-0.2, 0.2)
    rel0 = torch.empty(n_rel, hidden, device=DEV).uniform_(-0.2, 0.2)
    lr = 0.05

    def fresh(n_rows):
        eng = StepEngine(model, n_rows, n_rel, hidden, 12.0, lr, DEV, de_flag, False, True, 1.0, 1e-6, 3)
        eng.rel.copy_(rel0); eng.rel_state.zero_()
        return eng
    # reference: the single-table step
    eng = fresh(n_ent)
    eng.ent.copy_(ent0); eng.ent_state.zero_()
    smp = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=6, seed=5)
    for n in (6, 5, 6, 6):
        for b in smp.sample(n):
            eng.step(b)
    torch.cuda.synchronize()
    want = (eng.ent.clone(), eng.ent_state.clone(), eng.rel.clone(), eng.rel_state.clone())
    got = {}
    for name, env, graph in (("shortcut", "1", False), ("shortcut_graph", "1", True), ("routed", "0", False)):
        os.environ["KGE_DIST_LOCAL_SHORTCUT"] = env
        try:
            e2 = fresh(1)
            ent, state = ent0.clone(), torch.zeros(n_ent, device=DEV)
            de = kd.DistEngine(e2, kd.ShardSpec(n_ent, 1, 0), ent, state)
            assert de.local_only == (env == "1")
            s2 = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=6, seed=5)
            for n in (6, 5, 6, 6):
                de.run_group(s2.sample(n), graph=graph)
            torch.cuda.synchronize()
            assert de.check_overflow() == 0
            got[name] = (ent, state, e2.rel.clone(), e2.rel_state.clone())
            de.close()
        finally:
            os.environ.pop("KGE_DIST_LOCAL_SHORTCUT", None)
    for name in ("shortcut", "shortcut_graph"):
        for x, y in zip(want, got[name]):
            assert torch.equal(x, y), name
    for x, y in zip(want, got["routed"]):
        assert float((x - y).abs().max()) <= 1e-3 * lr, "routed path vs in-place step"
    assert float((want[1] > 0).sum()) > 100



def test_route_build_packed_message_rows_match_numpy():
    """round 6 (ABI 8): kge_route_build with ue_msg - per union entry the row of its single-trace message, owner * (cap + cap2) +
    position, and for an entry that is in BOTH traces the position of its second message in the bucket's extra region (its rank
    among the both-trace entries of the bucket); entries beyond cap2 are counted in the overflow word."""
    from dglke_amd import plan
    from dglke_amd import dist as kd
    rng = np.random.RandomState(13)
    for world, n_ent, B, N, cap, cap2 in ((1, 300, 64, 16, None, 64), (2, 301, 64, 16, 120, 40), (4, 600, 96, 32, 90, 24),
                                          (8, 900, 128, 32, 64, 3), (3, 5000, 512, 128, 700, 16)):     # (8, .., 3): the extra region overflows
        bt = O.synth_batch(rng, n_ent, 7, B, N, N, 1)
        b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"], DEV)
        per = (n_ent + world - 1) // world
        cap = cap or b.UE
        bf = type("S", (), {})()
        z = lambda n, dt: torch.zeros(n, dtype=dt, device=DEV)
        bf.req_ids, bf.h_loc, bf.t_loc, bf.neg_loc = z(world * cap, torch.int64), z(B, torch.int64), z(B, torch.int64), z(b.C * N, torch.int64)
        bf.ue_loc, bf.ue_rec_loc, bf.overflow, bf.ue_msg = z(b.UE, torch.int64), z(b.UE * 8, torch.int32), z(1, torch.int32), z(b.UE * 2, torch.int32)
        lb = kd.HipOps().route(b, world, per, cap, bf, cap2=cap2)
        torch.cuda.synchronize()
        assert lb.msg_rows == bf.ue_msg.data_ptr()
        ue, rec = b.p["ue_id"], b.p["ue_rec"].reshape(-1, 8)
        owner = np.minimum(ue // per, world - 1)
        start = np.searchsorted(owner, np.arange(world + 1))
        pos = np.arange(len(ue)) - start[owner]
        both = (rec[:, 3] > rec[:, 2]) & (rec[:, 5] > rec[:, 4])
        rank = np.zeros(len(ue), np.int64)
        for o in range(world):
            m = owner == o
            rank[m] = np.cumsum(both[m]) - both[m]
        capT = cap + cap2
        fits = pos < cap
        want_main = np.where(fits, owner * capT + pos, world * capT)
        want_extra = np.where(both & fits & (rank < cap2), rank, -1)
        got = bf.ue_msg.cpu().numpy().reshape(-1, 2)[:len(ue)]
        assert np.array_equal(got[:, 0], want_main) and np.array_equal(got[:, 1], want_extra), (world, cap2)
        assert int(bf.overflow.item()) == int((~fits).sum()) + int((both & fits & (rank >= cap2)).sum())
        assert both.sum() > 0


@pytest.mark.parametrize("nsrc,cap,cap2,dim,n_rows", [(1, 40, 8, 8, 64), (2, 33, 5, 8, 40), (3, 100, 17, 64, 120), (8, 257, 64, 400, 600),
                                                      (5, 7, 7, 1024, 12), (64, 9, 2, 8, 30)])
def test_apply_merged_packed_messages_equal_sequential_apply(nsrc, cap, cap2, dim, n_rows):
    """round 6 (ABI 8): packed single-trace messages [g | gs | link] - a row's first message, and for link >= 0 a second one in the
    extra region of the SAME source's bucket, applied behind the first; rows from several sources in source order.  Against the
    sequential numpy statement (kvserver.py:41-51 per pushed trace, tensor_models.py:316 trace order)."""
    from dglke_amd import dist as kd
    rng = np.random.RandomState(nsrc * 1000 + cap + cap2)
    lo, lr = 1000, 0.1
    ld, capT = dim + 4, cap + cap2
    table = rng.randn(n_rows, dim).astype(np.float32)
    state = rng.rand(n_rows).astype(np.float32)
    ids = np.full((nsrc, cap), -1, np.int64)
    msg = (rng.randn(nsrc * capT, ld) * 0.05).astype(np.float32)
    msg[:, dim] = rng.rand(nsrc * capT).astype(np.float32) * 0.01
    link = np.full((nsrc, cap), -1, np.int64)
    for s in range(nsrc):
        k = rng.randint(0, min(cap, n_rows) + 1)
        ids[s, :k] = np.sort(rng.choice(n_rows, k, replace=False)) + lo
        two = np.nonzero(rng.rand(k) < 0.3)[0][:cap2]            # rows that also carry a second (negative-trace) message
        link[s, two] = rng.permutation(cap2)[:len(two)]          # any injective placement inside the extra region
    sil = rng.rand(nsrc * capT) < 0.15
    msg[sil, dim] = 0.0                                          # silent traces (increment 0) are skipped
    mv = msg.view(np.int32)
    for s in range(nsrc):
        mv[s * capT:s * capT + cap, dim + 1] = link[s]
    t_d, s_d = torch.from_numpy(table).to(DEV), torch.from_numpy(state).to(DEV)
    kd.HipOps().apply_merged(t_d, s_d, nsrc, cap, torch.from_numpy(ids.reshape(-1)).to(DEV), lo, torch.from_numpy(msg).to(DEV), 1, lr,
                             cap_extra=cap2)
    torch.cuda.synchronize()
    t64, s64 = table.astype(np.float64), state.astype(np.float64)
    n_two = 0
    for s in range(nsrc):
        for p in range(cap):
            i = ids[s, p]
            if i < 0:
                continue
            rows = [s * capT + p] + ([s * capT + cap + link[s, p]] if link[s, p] >= 0 else [])
            n_two += len(rows) == 2
            for r in rows:
                m = msg[r].astype(np.float64)
                if m[dim] == 0.0:
                    continue
                s64[i - lo] += m[dim]
                t64[i - lo] += -lr * m[:dim] / (np.sqrt(s64[i - lo]) + 1e-10)
    np.testing.assert_allclose(s_d.cpu().numpy(), s64, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(t_d.cpu().numpy(), t64, rtol=1e-5, atol=2e-6)
    assert n_two > 0 or cap2 < 3


def test_packed_messages_extra_region_grows_on_a_small_graph():
    """round 6: on a small graph most rows of a batch are in BOTH traces - the extra region of the packed gradient messages (64 rows to start
    with) must grow before the first group runs (DistEngine.ensure_capacity: kge_route_fill's second word), every exchange buffer and the
    route pool are rebuilt, the group graphs are recorded for the new geometry, and the tables equal the single-table step (world 1 through
    RcclComm, synchronous and overlapped, eager first group + replayed groups)."""
    from dglke_amd import dist as kd
    from dglke_amd.dataloader import DeviceSampler
    from dglke_amd.engine import StepEngine
    n_ent, n_rel, hidden, B, N = 260, 9, 32, 128, 64
    rng = np.random.RandomState(77)
    h, r, t = rng.randint(0, n_ent, 20000), rng.randint(0, n_rel, 20000), rng.randint(0, n_ent, 20000)
    torch.manual_seed(21)
    ent0 = torch.empty(n_ent, hidden, device=DEV).uniform_(-0.3, 0.3)
    rel0 = torch.empty(n_rel, hidden, device=DEV).uniform_(-0.3, 0.3)
    lr = 0.05
    ref = StepEngine("DistMult", n_ent, n_rel, hidden, 6.0, lr, DEV, False, False, True, 1.0, 1e-6, 3)
    ref.ent.copy_(ent0); ref.rel.copy_(rel0); ref.ent_state.zero_(); ref.rel_state.zero_()
    smp = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=4, seed=9)
    for _ in range(4):
        for b in smp.sample(4):
            ref.step(b)
    torch.cuda.synchronize()
    for sched in (False,):          # (the synchronous schedule: the overlapped one is one step stale inside a group by design)
        e2 = StepEngine("DistMult", 1, n_rel, hidden, 6.0, lr, DEV, False, False, True, 1.0, 1e-6, 3)
        e2.rel.copy_(rel0); e2.rel_state.zero_()
        ent, state = ent0.clone(), torch.zeros(n_ent, device=DEV)
        comm = kd.RcclComm()
        de = kd.DistEngine(e2, kd.ShardSpec(n_ent, 1, 0), ent, state, always_collective=True, comm=comm)
        s2 = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=4, seed=9)
        logs = []
        replayed = 0
        for _ in range(4):
            replayed += bool(de.run_group(s2.sample(4), log=logs.append, graph=True, pipelined=sched))
        torch.cuda.synchronize()
        assert de.packed and de.grown_extra and de.cap2 > 64, (de.cap2, de.grown_extra)
        assert any("extra region of the packed gradient messages grows" in m for m in logs)
        assert de.check_overflow() == 0 and replayed >= 2
        assert float((ref.ent.cpu() - ent.cpu()).abs().max()) <= 1e-3 * lr and float((ref.rel.cpu() - e2.rel.cpu()).abs().max()) <= 1e-3 * lr
        assert float((ref.ent_state.cpu() - state.cpu()).abs().max()) <= 1e-5 * float(ref.ent_state.max())
        de.close()
