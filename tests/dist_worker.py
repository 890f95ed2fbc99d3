"""worker of tests/test_gpu_dist.py: one of `world` processes sharing ONE GPU; the device arithmetic of the range-sharded step
(dglke_amd.dist.HipOps: kge_route_build, kge_gather_rows_req, kge_step_grads, kge_adagrad_apply_merged) runs for real, the
fixed-size messages travel through gloo (staged through the host: RCCL refuses two ranks on one device).
With a 6th argument "rccl" every rank takes its OWN device (cuda:rank) and the messages travel through the real transport of the
product - dist.make_comm(): librccl called directly on the step's streams (RcclComm), the push grouped, the pull on the side
stream - which needs >= world GPUs (tests/test_gpu_dist.py skips it on a one-GPU box).
argv: rank world port out_dir mode [transport]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "dgl-ke_amd"))
sys.path.insert(0, os.path.join(HERE, ".."))

N_ENT, N_REL, HID, B, N, LR, STEPS = 2003, 31, 64, 128, 32, 0.1, 4
MODELS = (("TransE_l2", False, False), ("DistMult", False, False), ("RotatE", True, False))


def batches(world, steps, mode, seed=5):
    """per step and rank one batch of GLOBAL ids.  mode 'disjoint': rank k draws entities / relations from its own slice, so the
    synchronous step equals processing the ranks' batches one after the other on ONE table."""
    from oracle import kge_oracle as O
    rng = np.random.RandomState(seed)
    out = []
    for s in range(steps):
        row = []
        for k in range(world):
            if mode == "disjoint":
                ne, nr = N_ENT // world, N_REL // world
                bt = O.synth_batch(rng, ne, nr, B, N, N, s + 1)
                for key in ("h", "t", "neg", "nid"):
                    bt[key] = bt[key] + k * ne
                bt["r"] = bt["r"] + k * nr
            else:
                bt = O.synth_batch(rng, N_ENT, N_REL, B, N, N, s + 1)
                if mode == "relpart":        # rank k's triples use the relations r = k mod world only (the reference's --rel_part)
                    top = (N_REL - 1 - k) // world * world + k
                    bt["r"] = np.minimum((bt["r"] // world) * world + k, top)
            row.append(bt)
        out.append(row)
    return out


def main():
    rank, world, port, out_dir, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    transport = sys.argv[6] if len(sys.argv) > 6 else "host"
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dglke_amd import dist as kd, plan
    from dglke_amd.engine import StepEngine
    dev = "cuda:%d" % (rank if transport == "rccl" else 0)
    torch.cuda.set_device(torch.device(dev))
    comm = None
    if transport == "rccl":
        comm = kd.make_comm(kind="rccl")
        assert type(comm).__name__ == "RcclComm", "the direct librccl communicator could not be set up: %r" % (comm,)
    res = {}
    for model, de_, dr_ in MODELS:
        d_e = 2 * HID if de_ else HID
        g = torch.Generator().manual_seed(5)
        ent0 = ((torch.rand(N_ENT, d_e, generator=g) - 0.5) * 0.4)
        eng = StepEngine(model, 1, N_REL, HID, 12.0, LR, dev, de_, dr_, True, 1.0, 1e-6, 3,
                         flags=32 if mode == "nd" else 0)          # nd: --neg_deg_sample in the gradient-emitting step
        rel0 = ((torch.rand(N_REL, eng.d_r, generator=g) - 0.5) * 0.4)
        eng.rel.copy_(rel0)
        spec = kd.ShardSpec(N_ENT, world, rank)
        ent = ent0[spec.lo:spec.hi].to(dev).contiguous()
        state = torch.zeros(spec.n_local, device=dev)
        de = kd.DistEngine(eng, spec, ent, state, comm=comm or kd.HostStagedComm(), cap=None, slack=1.6,
                           rel_local=(mode == "relpart"))
        if mode.startswith("sampled"):
            # device-sampled batches, the trainer's order (sample a group, prepare_group = capacity + routing of the whole group +
            # ONE id exchange for the group, then its steps): two groups of two steps over the same two sampler slots, so the
            # second group runs on REFILLED slots.  The ids the kernel drew are read back for the test's fp64 statement.
            from dglke_amd.dataloader import DeviceSampler
            G = 2
            rk = np.random.RandomState(100 + rank)
            n_train = 3 * STEPS * B
            smp = DeviceSampler(rk.randint(0, N_ENT, n_train), rk.randint(0, N_REL, n_train), rk.randint(0, N_ENT, n_train),
                                N_ENT, B, N, dev, n_slots=G, seed=7 + rank)
            drawn = []
            for _ in range(STEPS // G):
                dbs = smp.sample(G)
                de.prepare_group(dbs)
                torch.cuda.synchronize()
                for b in dbs:
                    a = smp.slot_arrays(b.slot)
                    drawn.append(dict(h=a["h_gid"], t=a["t_gid"], r=a["rel_ids"], neg=a["neg_ids"], neg_head=np.int64(b.neg_head)))
                if mode == "sampled_overlap":     # push, apply and pull on the side stream: the same statement as sampled_pipelined
                    de._steps(dbs, "overlap")
                    continue
                for k, b in enumerate(dbs):
                    if mode == "sampled_pipelined":
                        de.step_pipelined(b, dbs[k + 1] if k + 1 < G else None)
                    else:
                        de.step(b)
            everybody = [None] * world
            dist.all_gather_object(everybody, drawn)
            bts, devb = [], []
            if rank == 0:
                res.setdefault("_drawn", {})[model] = everybody
        else:
            bts = batches(world, STEPS, mode if mode in ("disjoint", "relpart") else "random")
            devb = []
        ue_bound = 2 * B + (B // N) * N
        for row in bts:
            bt = row[rank]
            b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"], dev)
            b.UE = ue_bound                  # buffers are sized once for the bound (like the device sampler's slots)
            devb.append(b)
        if mode == "overlap" and devb:            # all steps as ONE group of the overlapped schedule: the 'pipelined' statement
            de._steps(devb, "overlap")
            devb = []
        for s, b in enumerate(devb):
            if mode == "pipelined":
                de.step_pipelined(b, devb[s + 1] if s + 1 < len(devb) else None)
            else:
                de.step(b)
        torch.cuda.synchronize()
        assert de.check_overflow() == 0
        if mode == "relpart":                # the owners' relation rows are collected on rank 0, like A2ATrainer.sync_tables does
            kd.relation_rows_from_owners(eng.rel, eng.rel_state, np.arange(N_REL) % world)
        shards = [None] * world
        dist.all_gather_object(shards, (ent.cpu().numpy(), state.cpu().numpy(), eng.rel.cpu().numpy(), eng.rel_state.cpu().numpy()))
        if rank == 0:
            res[model] = dict(ent=np.concatenate([s_[0] for s_ in shards]), state=np.concatenate([s_[1] for s_ in shards]),
                              rels=[s_[2] for s_ in shards], rel_states=[s_[3] for s_ in shards],
                              init_ent=ent0.numpy(), init_rel=rel0.numpy())
            if mode == "disjoint":           # the same batches, rank after rank, on ONE table through the fused single-GPU step
                ref = StepEngine(model, N_ENT, N_REL, HID, 12.0, LR, dev, de_, dr_, True, 1.0, 1e-6, 3)
                ref.load_tables(ent0.to(dev), rel0.to(dev))
                for row in bts:
                    for bt in row:
                        ref.step(plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"], dev))
                torch.cuda.synchronize()
                res[model].update(ref_ent=ref.ent.cpu().numpy(), ref_state=ref.ent_state.cpu().numpy(),
                                  ref_rel=ref.rel.cpu().numpy(), ref_rel_state=ref.rel_state.cpu().numpy())
        dist.barrier()
    drawn_all = res.pop("_drawn", {})
    extra = {}
    for m, everybody in drawn_all.items():          # [rank][step] -> arrays [step, rank, ...]
        for key in ("h", "t", "r", "neg", "neg_head"):
            extra["%s_drawn_%s" % (m, key)] = np.stack([np.stack([everybody[k][s_][key] for k in range(world)])
                                                        for s_ in range(len(everybody[0]))])
    if rank == 0:
        np.savez(os.path.join(out_dir, "result.npz"), **extra, **{m + "_" + k: v for m, d in res.items() for k, v in d.items()
                                                         if not isinstance(v, list)},
                 **{m + "_rel%d" % i: r for m, d in res.items() for i, r in enumerate(d["rels"])},
                 **{m + "_relstate%d" % i: r for m, d in res.items() for i, r in enumerate(d["rel_states"])})
    if comm is not None and hasattr(comm, "close"):
        torch.cuda.synchronize()
        comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
