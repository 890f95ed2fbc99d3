"""Which ingredient makes captured RCCL collectives hang on this stack?  (world 1, librccl through ctypes - no c10d, no watchdog)

    python tools/dbg/rccl_capture_matrix.py            # runs the matrix: one child process per case, each under its own timeout
    python tools/dbg/rccl_capture_matrix.py case ...   # one case (see `one`)

Every case captures `ncoll` x [ncclAllToAll(a -> b) ; b += 1] into ONE hipGraph on a side stream and replays it `replays` times,
synchronising the host every `sync` replays (0 = only at the end), optionally with an eager collective on the same communicator
between replays (`mix`).  The child prints a progress mark every 8 replays; the parent reports the last mark it saw - a case that
does not reach `done` within its timeout hung there.  Environment variants are passed to the child (`env`).
"""
import ctypes as C
import json
import os
import subprocess
import sys
import time


def one(argv):
    import torch
    nbytes, ncoll, replays, sync, mix, group = (int(x) for x in argv[:6])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class Uid(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    L.ncclAllToAll.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    uid = Uid()
    assert L.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert L.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    a = torch.arange(nbytes // 4, dtype=torch.float32, device=dev)
    b = torch.zeros_like(a)
    c = torch.zeros_like(a)
    s = torch.cuda.Stream()

    def a2a(src, dst, stream):
        rc = L.ncclAllToAll(src.data_ptr(), dst.data_ptr(), nbytes, 0, comm, stream)
        assert rc == 0, rc

    with torch.cuda.stream(s):
        for _ in range(3):
            a2a(a, b, s.cuda_stream)
    torch.cuda.synchronize()
    print("eager", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(ncoll):
            if group:
                L.ncclGroupStart()
            a2a(a, b, s.cuda_stream)
            if group:
                a2a(a, c, s.cuda_stream)
                L.ncclGroupEnd()
            b.add_(1.0)
    torch.cuda.synchronize()
    print("captured", flush=True)
    t0 = time.time()
    for r in range(replays):
        g.replay()
        if mix:
            a2a(a, c, torch.cuda.current_stream().cuda_stream)
        if sync and (r + 1) % sync == 0:
            torch.cuda.synchronize()
        if (r + 1) % 8 == 0:
            print("replay %d" % (r + 1), flush=True)
    torch.cuda.synchronize()
    ok = bool((b[:4].cpu() == (a[:4].cpu() + 1.0)).all())
    print("done %.1f us/replay ok=%s" % ((time.time() - t0) * 1e6 / replays, ok), flush=True)
    os._exit(0)                      # (no communicator teardown: it is not what is being probed)


CASES = [
    # name, bytes, ncoll, replays, sync, mix, group, env
    ("small_1coll_sync1", 4096, 1, 64, 1, 0, 0, {}),
    ("small_4coll_nosync", 4096, 4, 64, 0, 0, 0, {}),
    ("small_4coll_nosync_nomix", 4096, 4, 64, 0, 0, 0, {"NCCL_GRAPH_MIXING_SUPPORT": "0"}),
    ("big_3coll_sync1", 13 << 20, 3, 64, 1, 0, 0, {}),
    ("big_3coll_sync1_nomix", 13 << 20, 3, 64, 1, 0, 0, {"NCCL_GRAPH_MIXING_SUPPORT": "0"}),
    ("big_3coll_nosync_nomix", 13 << 20, 3, 64, 0, 0, 0, {"NCCL_GRAPH_MIXING_SUPPORT": "0"}),
    ("big_60coll_nosync_nomix", 13 << 20, 60, 32, 0, 0, 0, {"NCCL_GRAPH_MIXING_SUPPORT": "0"}),
    ("big_60coll_sync1", 13 << 20, 60, 32, 1, 0, 0, {}),
    ("big_3coll_nosync_noreg", 13 << 20, 3, 64, 0, 0, 0, {"NCCL_GRAPH_REGISTER": "0"}),
    ("big_3coll_mixed_eager_nomix", 13 << 20, 3, 64, 0, 1, 0, {"NCCL_GRAPH_MIXING_SUPPORT": "0"}),
    ("big_3coll_mixed_eager", 13 << 20, 3, 64, 1, 1, 0, {}),
    ("big_grouped_nosync_nomix", 13 << 20, 3, 64, 0, 0, 1, {"NCCL_GRAPH_MIXING_SUPPORT": "0"}),
]


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "case":
        return one(sys.argv[2:])
    out = []
    tmo = float(os.environ.get("PROBE_TIMEOUT", "25"))
    for name, nbytes, ncoll, replays, sync, mix, group, env in CASES:
        e = dict(os.environ)
        e.update(env)
        t0 = time.time()
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "case"] + [str(x) for x in (nbytes, ncoll, replays, sync, mix, group)],
                             env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
        try:
            txt, _ = p.communicate(timeout=tmo)
            hung = False
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, 9)
            txt, _ = p.communicate()
            hung = True
        marks = [l for l in txt.splitlines() if l.split(" ")[0] in ("eager", "captured", "replay", "done")]
        rec = dict(case=name, bytes=nbytes, ncoll=ncoll, replays=replays, sync=sync, mix=mix, group=group, env=env, hung=hung,
                   last=marks[-1] if marks else None, rc=p.returncode, seconds=round(time.time() - t0, 1),
                   tail=[l for l in txt.splitlines() if "amdgpu.ids" not in l][-3:] if (hung or p.returncode) else None)
        print(json.dumps(rec), flush=True)
        out.append(rec)


if __name__ == "__main__":
    main()
