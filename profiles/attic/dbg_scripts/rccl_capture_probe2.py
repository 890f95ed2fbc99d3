"""(round 5: what this probe ran into was NOT a replay hang - dist.destroy_process_group / ncclCommDestroy wait for every live hipGraph
that recorded the communicator's collectives; see tools/dbg/rccl_capture_matrix.py, profiles/r05_rccl_capture_diagnosis.txt)
RCCL collectives inside a hipGraph (world 1): does replay work when the host synchronises every k replays?
(round 2 found that a loop of 100 unsynchronised replays hangs; run under `timeout`)"""
import os, sys, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
a = torch.arange(1024, dtype=torch.float32, device=dev); b = torch.empty_like(a)
c = torch.empty(1024, dtype=torch.float32, device=dev)
dist.all_to_all_single(b, a); dist.all_gather_into_tensor(c, a); torch.cuda.synchronize()
NCOLL = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(NCOLL):
        dist.all_to_all_single(b, a)
        b.mul_(2.0)
        dist.all_gather_into_tensor(c, b)
torch.cuda.synchronize()
print("captured %d x (a2a + mul + allgather)" % NCOLL, flush=True)
for k in (1, 2, 4, 8, 16):
    t0 = time.time()
    n = 0
    for _ in range(64 // k):
        for _ in range(k):
            g.replay(); n += 1
        torch.cuda.synchronize()
    print("sync every %2d replays: %d replays ok, %.1f us per replay" % (k, n, (time.time() - t0) * 1e6 / n), flush=True)
dist.destroy_process_group()
