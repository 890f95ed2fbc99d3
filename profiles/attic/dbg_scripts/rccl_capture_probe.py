"""(round 5: what this probe ran into was NOT a replay hang - dist.destroy_process_group / ncclCommDestroy wait for every live hipGraph
that recorded the communicator's collectives; see tools/dbg/rccl_capture_matrix.py, profiles/r05_rccl_capture_diagnosis.txt)
does capturing RCCL collectives into a hipGraph work on this stack? (world = 1, run under `timeout`)"""
import os, sys, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
a = torch.arange(1024, dtype=torch.float32, device=dev); b = torch.empty_like(a)
c = torch.empty(1024, dtype=torch.float32, device=dev)
dist.all_to_all_single(b, a); dist.all_gather_into_tensor(c, a); torch.cuda.synchronize()
print("eager ok", flush=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        dist.all_to_all_single(b, a); dist.all_gather_into_tensor(c, a)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
t0 = time.time()
with torch.cuda.graph(g):
    for _ in range(4):
        dist.all_to_all_single(b, a)
        b.mul_(2.0)
        dist.all_gather_into_tensor(c, b)
print("captured in %.2fs" % (time.time() - t0), flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("replayed; c[1] =", c[1].item(), flush=True)
t0 = time.time()
for _ in range(100): g.replay()
torch.cuda.synchronize()
print("us per replay of 4x(a2a + mul + allgather): %.1f" % ((time.time() - t0) * 1e6 / 100))
dist.destroy_process_group()
