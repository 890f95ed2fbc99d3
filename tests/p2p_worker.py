"""worker of tests/test_gpu_p2p.py::test_two_processes_share_tables_over_ipc - one of two processes
on the SAME GPU; they map each other's shard through hipIpc handles and train alternately (barrier
between the turns) so that the outcome is deterministic and must equal one engine processing the
same batches in the same order on un-sharded tables.  argv: rank world port out_dir [device ids, comma separated:
rank k runs on device ids[k] - two DIFFERENT GPUs exercise the real xGMI path]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "dgl-ke_amd"))
sys.path.insert(0, os.path.join(HERE, ".."))


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dglke_amd import p2p, plan
    from dglke_amd.engine import StepEngine
    from oracle import kge_oracle as O      # batch generator only (test infrastructure)
    devs = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0] * world
    dev = "cuda:%d" % devs[rank]
    torch.cuda.set_device(devs[rank])
    n_ent, n_rel, hidden, B, N = 3001, 37, 64, 128, 32
    res = {}
    for model, de_, dr_ in (("TransE_l2", False, False), ("ComplEx", True, True), ("RotatE", True, False),
                            ("TransR", False, False), ("RESCAL", False, False)):
        # TransR / RESCAL (round 6): entity table sharded, relation-side tables local to the trainers, every trainer's batches use ITS
        # relations only (r = rank mod world: the relation partition of the CLI), the owners' rows collected at the end
        rel_side = model in ("TransR", "RESCAL")
        hid = 32 if rel_side else hidden
        d_e = 2 * hid if de_ else hid
        d_r = 2 * hid if dr_ else hid
        if model == "RESCAL":
            d_r = d_e * d_e
        tabs = p2p.ShardedTables(n_ent, n_rel, d_e, d_r, dev, world, rank, rel_local=rel_side,
                                 proj_dim=d_e * d_r if model == "TransR" else 0)
        assert tabs.probe(), "peer mappings do not reach the other process' memory"
        g = torch.Generator().manual_seed(5)
        ent0 = (torch.rand(n_ent, d_e, generator=g) - 0.5) * 0.4
        rel0 = (torch.rand(n_rel, d_r, generator=g) - 0.5) * 0.4
        proj0 = (torch.rand(n_rel, d_e * d_r, generator=g) - 0.5) * 2.0 if model == "TransR" else None
        tabs.load_full(ent0.to(dev), rel0.to(dev), proj=proj0)
        eng = StepEngine(model, n_ent, n_rel, hid, 12.0, 0.1, dev, de_, dr_, True, 1.0, 1e-6, 3, shards=tabs)
        ref = None
        if rank == 0:
            # (flag 2: DistMult / ComplEx on local tables would otherwise take their per-edge gradients from the backward GEMM's
            #  epilogue, the sharded step from the edge-gradient kernel - same formulas, not the same bits)
            ref = StepEngine(model, n_ent, n_rel, hid, 12.0, 0.1, dev, de_, dr_, True, 1.0, 1e-6, 3,
                             flags=2 if model in ("DistMult", "ComplEx", "SimplE") else 0)
            ref.load_tables(ent0.to(dev), rel0.to(dev))
            if proj0 is not None:
                ref.proj.copy_(proj0.to(dev)); ref.proj_state.zero_()
        rng = np.random.RandomState(3)
        torch.cuda.synchronize()
        dist.barrier()
        for step in range(1, 9):
            bt = O.synth_batch(rng, n_ent, n_rel, B, N, N, step)       # same stream on both ranks
            if rel_side:                                               # the turn's trainer owns the relations r = turn mod world
                k = step % world
                bt["r"] = np.minimum((bt["r"] // world) * world + k, (n_rel - 1 - k) // world * world + k)
            b = plan.make_batch(bt["h"], bt["t"], bt["r"], bt["neg"], N, N, bt["neg_head"], dev)
            if step % world == rank:
                eng.step(b)
            if ref is not None:
                ref.step(b)
            torch.cuda.synchronize()
            dist.barrier()
        if rel_side:
            tabs.collect_relations(np.arange(n_rel) % world)
        if rank == 0:
            ids_e = torch.arange(n_ent, device=dev)
            ids_r = torch.arange(n_rel, device=dev)
            ok = (torch.equal(tabs.gather("ent", ids_e), ref.ent) and
                  torch.equal(tabs.gather("ent_state", ids_e), ref.ent_state) and
                  torch.equal(tabs.gather("rel", ids_r), ref.rel) and
                  torch.equal(tabs.gather("rel_state", ids_r), ref.rel_state))
            if model == "TransR":
                ok = ok and torch.equal(tabs.proj_tab, ref.proj) and torch.equal(tabs.proj_state_tab, ref.proj_state)
            moved = float((ref.ent.cpu() - ent0).abs().max())
            res[model] = (bool(ok), moved)
        torch.cuda.synchronize()
        dist.barrier()
        tabs.close()
    if rank == 0:
        with open(os.path.join(out_dir, "result.txt"), "w") as f:
            for k, (ok, moved) in res.items():
                f.write("%s %d %.6g\n" % (k, int(ok), moved))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
