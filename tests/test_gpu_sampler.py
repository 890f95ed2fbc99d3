"""On-device sampler + plan builder (kge_sample_batches) vs the host plan (dglke_amd/plan.py) rebuilt
from the ids the kernel sampled; sampler semantics of dataloader/sampler.py:376-419, 853-859."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n_ent,n_rel,B,N,chunk", [(14951, 1345, 1000, 200, 200), (9, 2, 16, 4, 4), (86054151, 14824, 1024, 256, 256),
                                                    (500, 7, 120, 24, 40),
                                                    # the sampler launch's wide instance (round 6: 8 keys per thread, 13 code bits): the
                                                    # reference's batch-2048 recipes (6144 elements), the largest shape (8192), 64-bit keys
                                                    # (more than 2^19 entities), a tiny id range with long duplicate runs
                                                    (14951, 1345, 2048, 256, 256), (40943, 18, 2048, 128, 128), (3000, 11, 3072, 512, 1536),
                                                    (86054151, 14824, 2048, 512, 512), (50, 3, 2048, 64, 64), (20000, 30, 2048, 1024, 512)])
def test_device_plan_equals_host_plan(n_ent, n_rel, B, N, chunk):
    from dglke_amd import plan
    from dglke_amd.dataloader import DeviceSampler
    rng = np.random.RandomState(0)
    n_train = 5 * B + 17
    h, r, t = rng.randint(0, n_ent, n_train), rng.randint(0, n_rel, n_train), rng.randint(0, n_ent, n_train)
    s = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=5, neg_chunk_size=chunk, seed=3)
    batches = s.sample()                                       # the 5 whole batches of epoch 0
    torch.cuda.synchronize()
    perm = s.perm.cpu().numpy()
    C = B // chunk
    allneg = []
    for k, b in enumerate(batches):
        a = s.slot_arrays(k)
        e = perm[(k * B + np.arange(B)) % n_train]            # consecutive whole batches of the permuted epoch
        assert np.array_equal(a["h_gid"], h[e]) and np.array_equal(a["t_gid"], t[e]) and np.array_equal(a["rel_ids"], r[e])
        assert a["neg_ids"].min() >= 0 and a["neg_ids"].max() < n_ent
        assert b.neg_head == ((k + 1) % 2 == 0)               # step 1 corrupts tails, step 2 heads, ...
        assert a["counts"][2] == int(b.neg_head)
        allneg.append(a["neg_ids"])
        p = plan.build_plan(a["h_gid"], a["t_gid"], a["rel_ids"], a["neg_ids"], chunk, N, b.neg_head)
        UE, UR = int(a["counts"][0]), int(a["counts"][1])
        assert UE == p["UE"] and UR == p["UR"]
        assert np.array_equal(a["ue_id"][:UE], p["ue_id"])
        assert np.array_equal(a["ue_pos_ptr"][:UE + 1], p["ue_pos_ptr"])
        assert np.array_equal(a["ue_pos_adj"], p["ue_pos_adj"])
        assert np.array_equal(a["ue_neg_ptr"][:UE + 1], p["ue_neg_ptr"])
        assert np.array_equal(a["ue_neg_slot"], p["ue_neg_slot"])
        assert np.array_equal(a["ur_id"][:UR], p["ur_id"])
        assert np.array_equal(a["ur_ptr"][:UR + 1], p["ur_ptr"])
        assert np.array_equal(a["ur_edge"], p["ur_edge"])
        assert np.array_equal(a["ue_rec"][:8 * UE], p["ue_rec"])
        assert np.array_equal(a["ur_rec"][:8 * UR].reshape(-1, 8)[:, :5], p["ur_rec"].reshape(-1, 8)[:, :5])
    # the counter-based RNG: different steps differ, values spread over the id range
    allneg = np.concatenate(allneg)
    assert not np.array_equal(allneg[:C * N], allneg[C * N:2 * C * N])
    if n_ent > 1000:
        assert 0.35 * n_ent < allneg.mean() < 0.65 * n_ent
    # a second launch continues the step counter: epoch 1 (the 17 trailing triples of epoch 0 are dropped), new order
    more = s.sample(2)
    torch.cuda.synchronize()
    a = s.slot_arrays(0)
    assert more[0].neg_head == (6 % 2 == 0)
    assert not np.array_equal(a["h_gid"], h[perm[np.arange(B)]]), "epoch 1 repeats epoch 0's first batch"


def test_epochs_are_whole_batches_in_a_new_order():
    """reference: EdgeSampler(shuffle=True) reshuffles every epoch and the trailing partial batch is dropped
    (dataloader/sampler.py:408-419, 503-504).  Heads are the edge index here, so the sampled ids ARE the edges."""
    from dglke_amd.dataloader import DeviceSampler
    B, N, n_train, n_ent = 64, 16, 64 * 7 + 23, 10000
    h = np.arange(n_train)
    z = np.zeros(n_train, np.int64)
    s = DeviceSampler(h, z, z, n_ent, B, N, DEV, n_slots=21, seed=5)      # 3 epochs of 7 whole batches
    s.sample()
    torch.cuda.synchronize()
    epochs = [np.concatenate([s.slot_arrays(7 * e + k)["h_gid"] for k in range(7)]) for e in range(3)]
    for e, ids in enumerate(epochs):
        assert len(np.unique(ids)) == 7 * B, "epoch %d repeats an edge" % e     # a bijection: no edge twice in an epoch
        assert ids.min() >= 0 and ids.max() < n_train
    assert not np.array_equal(epochs[0], epochs[1]) and not np.array_equal(epochs[1], epochs[2])
    # batches are not just the same sets in another order: epoch 1's first batch mixes edges of many epoch-0 batches
    where0 = {int(x): k // B for k, x in enumerate(epochs[0])}
    assert len({where0.get(int(x), -1) for x in epochs[1][:B]}) > 3


def test_epoch_orders_do_not_repeat_with_period_eight():
    """the per-epoch order is perm o (i * mul + add mod n) with BOTH constants hashed from (seed, epoch) - a table of eight
    multipliers made epochs e and e + 8 the same cyclic sequence (same successor of every edge), i.e. nearly identical
    batches.  Twelve epochs: no two of them share the successor relation."""
    from dglke_amd.dataloader import DeviceSampler
    B, N, nb, n_ent = 32, 8, 9, 10000
    n_train = B * nb
    h = np.arange(n_train)
    z = np.zeros(n_train, np.int64)
    s = DeviceSampler(h, z, z, n_ent, B, N, DEV, n_slots=12 * nb, seed=11)
    s.sample()
    torch.cuda.synchronize()
    succ = []
    for e in range(12):
        ids = np.concatenate([s.slot_arrays(nb * e + k)["h_gid"] for k in range(nb)])
        assert len(np.unique(ids)) == n_train
        nxt = np.empty(n_train, np.int64)
        nxt[ids] = np.roll(ids, -1)            # cyclic successor of every edge in this epoch's order
        succ.append(nxt)
    for a in range(12):
        for b in range(a + 1, 12):
            same = (succ[a] == succ[b]).mean()
            assert same < 0.2, "epochs %d and %d walk (almost) the same sequence: %.0f %% equal successors" % (a, b, 100 * same)


def test_sampler_refuses_fewer_triples_than_a_batch_with_a_message():
    from dglke_amd import _lib
    from dglke_amd.dataloader import DeviceSampler
    z = np.zeros(10, np.int64)
    s = DeviceSampler(z, z, z, 100, 16, 4, DEV, n_slots=2, seed=1)
    with pytest.raises(_lib.KgeError, match="fewer training triples than one batch"):
        s.sample()


@pytest.mark.parametrize("rel_hub", [0.0, 0.4], ids=["uniform", "dominant_relation"])
def test_step_from_device_batch_equals_step_from_host_plan(rel_hub):
    """the fused step fed by a device-built batch gives bit-identical tables to the step fed by the
    host plan of the same ids.  (Device-built batches carry the length of their longest relation list and run the plain
    relation instance of the update kernel when no list is long; host plans always run the list-sharing instance: both
    must give the same bits - with a dominant relation both share the list.)"""
    from dglke_amd import plan
    from dglke_amd.dataloader import DeviceSampler
    from dglke_amd.engine import StepEngine
    rng = np.random.RandomState(1)
    n_ent, n_rel, B, N, D = 3000, 40, 256, 64, 64
    n_train = 4000
    h, r, t = rng.randint(0, n_ent, n_train), rng.randint(0, n_rel, n_train), rng.randint(0, n_ent, n_train)
    r[rng.rand(n_train) < rel_hub] = 3
    s = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=4, seed=9)
    dbs = s.sample()
    torch.cuda.synchronize()
    torch.manual_seed(0)
    e1 = StepEngine("TransE_l2", n_ent, n_rel, D, 12.0, 0.1, DEV, False, False, True, 1.0, 1e-6, 3)
    e2 = StepEngine("TransE_l2", n_ent, n_rel, D, 12.0, 0.1, DEV, False, False, True, 1.0, 1e-6, 3)
    e2.load_tables(e1.ent.clone(), e1.rel.clone())
    for k, db in enumerate(dbs):
        a = s.slot_arrays(k)
        hb = plan.make_batch(a["h_gid"], a["t_gid"], a["rel_ids"], a["neg_ids"], N, N, db.neg_head, DEV)
        e1.step(db)
        e2.step(hb)
    torch.cuda.synchronize()
    assert torch.equal(e1.ent, e2.ent) and torch.equal(e1.rel, e2.rel)
    assert torch.equal(e1.ent_state, e2.ent_state) and torch.equal(e1.rel_state, e2.rel_state)
    assert np.allclose(e1.read_loss_sums(), e2.read_loss_sums(), rtol=1e-6)


@pytest.mark.parametrize("n_ent,n_rel,B,N,chunk,model,skewed", [
    (14951, 1345, 1000, 200, 200, "TransE_l2", False),       # cfg-T: 32-bit keys, ~750 keys per bucket
    (500, 7, 120, 24, 40, "TransE_l2", False),               # chunk != N, tiny id range: long duplicate runs
    (60000, 40, 1024, 256, 256, "ComplEx", True),            # ids sorted by popularity: one bucket takes most keys (big instance)
    (9, 2, 16, 4, 4, "DistMult", False),
])
def test_sampler_tail_on_the_step_launches_builds_the_same_batches(n_ent, n_rel, B, N, chunk, model, skewed):
    """VERDICT r04 next-3 (round 5): kge_step_fused_sampling - tail workgroups on the step's first / backward / update launches
    build, phase by phase, one batch of the NEXT group per step.  Every array of every slot equals, bit for bit, what the
    stand-alone sampler launch (kge_sample_batches) builds from the same state - ids from the same counter RNG and epoch
    permutation, the same plan - over three groups (the third crosses an epoch boundary), the device state advances alike, and
    the step that carries the tail trains exactly as without it."""
    from dglke_amd.dataloader import DeviceSampler
    from dglke_amd.engine import StepEngine
    rng = np.random.RandomState(11)
    n_train = 7 * B + 5
    if skewed:                                               # 85 % of the edge ends in the lowest 2 % of the id range
        hot = lambda n: np.where(rng.rand(n) < 0.85, rng.randint(0, max(2, n_ent // 50), n), rng.randint(0, n_ent, n))
        h, t = hot(n_train), hot(n_train)
    else:
        h, t = rng.randint(0, n_ent, n_train), rng.randint(0, n_ent, n_train)
    r = rng.randint(0, n_rel, n_train)
    hidden = 32
    G = 3
    de = model == "ComplEx"

    def make():
        torch.manual_seed(1)
        e_ = StepEngine(model, n_ent, n_rel, hidden, 12.0, 0.1, DEV, de, de, True, 1.0, 1e-6, 3)
        s_ = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=2 * G, neg_chunk_size=chunk, seed=3)
        return e_, s_
    eng_a, smp_a = make()          # reference: stand-alone sampler launches
    eng_b, smp_b = make()          # tail jobs
    cur_a, cur_b = smp_a.sample(G, slot0=0), smp_b.sample(G, slot0=0)
    half = 0
    for grp in range(3):
        nxt_a = smp_a.sample(G, slot0=(half ^ 1) * G)
        jobs, nxt_b = smp_b.tail_jobs(G, slot0=(half ^ 1) * G)
        for k in range(G):
            eng_a.step(cur_a[k])
            eng_b.step(cur_b[k], sample_job=jobs[k])
        torch.cuda.synchronize()
        for k in range(G):
            a, b = smp_a.slot_arrays((half ^ 1) * G + k), smp_b.slot_arrays((half ^ 1) * G + k)
            UE, UR = int(a["counts"][0]), int(a["counts"][1])
            assert np.array_equal(a["counts"], b["counts"]), (grp, k, a["counts"], b["counts"])
            for name, n in (("h_gid", B), ("t_gid", B), ("rel_ids", B), ("neg_ids", (B // chunk) * N), ("ue_id", UE), ("ur_id", UR),
                            ("ue_pos_ptr", UE + 1), ("ue_pos_adj", 2 * B), ("ue_neg_ptr", UE + 1), ("ue_neg_slot", (B // chunk) * N),
                            ("ur_ptr", UR + 1), ("ur_edge", B), ("ue_rec", 8 * UE), ("ur_rec", 8 * UR)):
                assert np.array_equal(a[name][:n], b[name][:n]), "group %d batch %d: %s differs" % (grp, k, name)
            assert nxt_a[k].neg_head == nxt_b[k].neg_head
        assert torch.equal(smp_a.state[:2], smp_b.state[:2]), "device state"
        cur_a, cur_b, half = nxt_a, nxt_b, half ^ 1
    assert torch.equal(eng_a.ent, eng_b.ent) and torch.equal(eng_a.rel, eng_b.rel) and torch.equal(eng_a.ent_state, eng_b.ent_state)
    assert smp_a.host_step == smp_b.host_step


def test_sampler_tail_64bit_keys():
    """entity ids beyond 2^20 (Freebase: 86 M) take the 64-bit key instances of the three phases: the triples' ids span the whole
    range, the table of the carrying steps is a small stand-in (a step trains on device batches of ANOTHER, small-id sampler with
    the same geometry while its launches build the big-id batches)."""
    from dglke_amd.dataloader import DeviceSampler
    from dglke_amd.engine import StepEngine
    rng = np.random.RandomState(5)
    n_ent, n_rel, B, N = 86054151, 14824, 1024, 256
    n_train = 4 * B + 9
    h, r, t = rng.randint(0, n_ent, n_train), rng.randint(0, n_rel, n_train), rng.randint(0, n_ent, n_train)
    eng = StepEngine("DistMult", 5000, n_rel, 32, 12.0, 0.1, DEV, False, False, True, 1.0, 1e-6, 3)
    carrier = DeviceSampler(h % 5000, r, t % 5000, 5000, B, N, DEV, n_slots=3, seed=1)
    cur = carrier.sample(3)
    smp_a = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=3, seed=3)
    smp_b = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=3, seed=3)
    for rnd in range(2):                                     # the second round crosses the epoch boundary
        smp_a.sample(3)
        jobs, _ = smp_b.tail_jobs(3)
        for k in range(3):
            eng.step(cur[k], sample_job=jobs[k])
        torch.cuda.synchronize()
        for k in range(3):
            a, b = smp_a.slot_arrays(k), smp_b.slot_arrays(k)
            UE, UR = int(a["counts"][0]), int(a["counts"][1])
            assert np.array_equal(a["counts"], b["counts"])
            for name, n in (("h_gid", B), ("t_gid", B), ("rel_ids", B), ("neg_ids", N * 4), ("ue_id", UE), ("ur_id", UR), ("ue_pos_ptr", UE + 1),
                            ("ue_pos_adj", 2 * B), ("ue_neg_ptr", UE + 1), ("ue_neg_slot", N * 4), ("ur_ptr", UR + 1), ("ur_edge", B),
                            ("ue_rec", 8 * UE), ("ur_rec", 8 * UR)):
                assert np.array_equal(a[name][:n], b[name][:n]), "round %d batch %d: %s differs" % (rnd, k, name)
        assert torch.equal(smp_a.state[:2], smp_b.state[:2])


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hipGraph"])
def test_prefetched_groups_fused_trains_exactly_like_the_sampler_launch(graph):
    """dataloader.PrefetchedGroups(mode='fused'): step k of a group builds batch k of the next group on its own launches - the
    tables after four groups (sizes 6, 6, 4, 6: a smaller next group, then a LARGER one, which falls back to the launch) equal
    those of mode='serial' bit for bit, eagerly and replayed from hipGraphs (captured once per group geometry, replayed)."""
    from dglke_amd.dataloader import DeviceSampler, PrefetchedGroups
    from dglke_amd.engine import StepEngine
    rng = np.random.RandomState(2)
    n_ent, n_rel, B, N = 3000, 30, 256, 64
    n_train = 9 * B + 3
    h, r, t = rng.randint(0, n_ent, n_train), rng.randint(0, n_rel, n_train), rng.randint(0, n_ent, n_train)
    res = []
    for mode in ("serial", "fused"):
        torch.manual_seed(4)
        eng = StepEngine("TransE_l2", n_ent, n_rel, 64, 12.0, 0.1, DEV, False, False, True, 1.0, 1e-6, 3)
        smp = DeviceSampler(h, r, t, n_ent, B, N, DEV, n_slots=12, seed=9)
        pg = PrefetchedGroups(smp, eng.step, group_max=6, mode=mode)
        eng.workspace_for(smp.sample(1)[0])
        smp.state[:2] = torch.tensor([0, 1], device=DEV)     # (the probe batch above is not part of the run)
        smp.host_step = 1
        sizes = [6, 6, 4, 6, 6, 6, 4, 6]
        for rep in range(2):                                 # the second round replays the graphs captured in the first
            pg.buf, pg.ready = 0, None
            pg.prefill(sizes[0])
            for i in range(len(sizes) - 1):
                pg.run(sizes[i + 1], graph=graph)
        torch.cuda.synchronize()
        res.append((eng.ent.clone(), eng.rel.clone(), eng.ent_state.clone(), smp.state[:2].clone(), smp.host_step))
    for x, y in zip(res[0][:4], res[1][:4]):
        assert torch.equal(x, y)
    assert res[0][4] == res[1][4]


def test_prepare_tail_refuses_to_run_inside_a_graph_capture():
    """the permuted triple copies the tail jobs read are made ONCE, outside any capture: made lazily inside the capture of a group graph
    (round 5's first version) the three gather kernels were replayed with every group - 24 us per replay"""
    from dglke_amd import _lib
    from dglke_amd.dataloader import DeviceSampler
    rng = np.random.RandomState(0)
    h, r, t = rng.randint(0, 100, 1000), rng.randint(0, 5, 1000), rng.randint(0, 100, 1000)
    s = DeviceSampler(h, r, t, 100, 64, 16, DEV, n_slots=4, seed=1)
    g = torch.cuda.CUDAGraph()
    with pytest.raises(_lib.KgeError):
        with torch.cuda.graph(g):
            s.tail_jobs(2)
    s.prepare_tail()
    assert torch.equal(s._Hp, s.H[s.perm]) and s._tail_scratch.numel() > 0
    jobs, batches = s.tail_jobs(2)                       # (fine now, also while capturing)
    assert len(jobs) == 2 and jobs[0].pre_permuted == 1 and jobs[1].prev_slot and not jobs[0].prev_slot and jobs[1].advance == 2
