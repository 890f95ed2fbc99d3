"""where does the hipGraph replay of the sharded step with its (world-1, forced) RCCL collectives get stuck?
    python tools/dbg/dist_graph_probe.py VARIANT     (run under `timeout`; faulthandler dumps every thread's Python stack after 25 s)
variants (comma separated flags):
    pg        torch.distributed nccl process group initialised first (what bench_dist.py has)
    pipe      eager warm-up with the pipelined pull (side stream) before the capture, else synchronous warm-up steps
    insample  the sampler launch + prepare_group (routing + group id exchange) INSIDE the graph, else outside
    relaxed   capture_error_mode="thread_local" instead of torch's default "global"
    torchcomm the c10d wrappers (TorchComm) instead of direct librccl
"""
import faulthandler
import os
import sys
import time

faulthandler.dump_traceback_later(25, exit=False, file=sys.stderr)
faulthandler.dump_traceback_later(25, exit=False, file=sys.stderr)
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dgl-ke_amd"))
import torch                                                      # noqa: E402


def mark(s):
    print("[%7.2f] %s" % (time.time() - T0, s), flush=True)


T0 = time.time()
flags = set(sys.argv[1].split(",")) if len(sys.argv) > 1 and sys.argv[1] != "-" else set()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if "pg" in flags:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dist.barrier()
    mark("process group up")
import __graft_entry__                                            # noqa: E402
__graft_entry__.build()
from dglke_amd import _lib, dist as kd                            # noqa: E402
from dglke_amd.dataloader import DeviceSampler                    # noqa: E402
from dglke_amd.engine import StepEngine                           # noqa: E402

n_ent, n_rel, hidden, B, N = 1000003, 14824, 400, 1024, 256
eng = StepEngine("RotatE", 1, n_rel, hidden, 12.0, 0.01, dev, True, False, True, 1.0, 1e-7, 3)
spec = kd.ShardSpec(n_ent, 1, 0)
ent = torch.empty(n_ent, 800, device=dev).uniform_(-0.03, 0.03)
state = torch.zeros(n_ent, device=dev)
comm = kd.TorchComm() if "torchcomm" in flags else kd.RcclComm()
de = kd.DistEngine(eng, spec, ent, state, comm=comm, always_collective=True)
g = torch.Generator(device=dev)
g.manual_seed(1)
H, T, R = (torch.randint(0, hi, (200000,), device=dev, generator=g) for hi in (n_ent, n_ent, n_rel))
G = 20
smp = DeviceSampler(H, R, T, n_ent, B, N, dev, n_slots=G, seed=1)
dbs = smp.sample(G)
eng.workspace_for(dbs[0])
de.prepare_group(dbs)
for k, b in enumerate(dbs[:6]):
    if "pipe" in flags:
        de.step_pipelined(b, dbs[k + 1] if k + 1 < 6 else None)
    else:
        de.step(b)
torch.cuda.synchronize()
mark("eager warm-up done (%s)" % ("pipelined" if "pipe" in flags else "synchronous"))
gr = torch.cuda.CUDAGraph()
kw = dict(capture_error_mode="thread_local") if "relaxed" in flags else {}
if "insample" not in flags:
    dbs = smp.sample(G)
    de.prepare_group(dbs)
    torch.cuda.synchronize()
with torch.cuda.graph(gr, **kw):
    if "insample" in flags:
        dbs = smp.sample(G)
        de.prepare_group(dbs, check_capacity=False)
    for b in dbs:
        de.step(b)
torch.cuda.synchronize()
mark("captured %d steps" % G)
for r in range(12):
    gr.replay()
    torch.cuda.current_stream().synchronize()
    mark("replay %d done" % (r + 1))
t0 = time.time()
for r in range(20):
    gr.replay()
torch.cuda.synchronize()
mark("20 unsynchronised replays: %.1f us per step" % ((time.time() - t0) * 1e6 / (20 * G)))
print("DONE loss sums", eng.read_loss_sums()[:3], flush=True)
os._exit(0)
