"""bench.py --gpus N (N > 1): the supervisor of bench_dist.orchestrate on CPU, world 2 over gloo, with a scripted worker
(tests/fake_dist_worker.py) in place of the GPU worker: a set-up that hangs is killed by the per-phase watchdog, a crash is seen,
one stuck rank fails the attempt for everybody, the chain walks a2a/rccl -> a2a/torch -> p2p -> replicas, a secondary leg that
hangs after the headline was delivered costs only itself, and rank 0 ALWAYS prints exactly one JSON line that names the mode and
why the earlier attempts failed (the first contact with xGMI must be survivable)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(plan, world=2, timeout=120):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "KGE_DIST_WORKER_SCRIPT": os.path.join(HERE, "fake_dist_worker.py"),
                    "KGE_FAKE_PLAN": plan, "KGE_DIST_PHASE_TIMEOUTS": "5,3,3,3,3,3,3"})
        env.pop("KGE_DIST_MODE", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "20",
                                       "--warmup", "5"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    assert all(rc == 0 for rc, _, _ in outs), outs
    lines = [ln for ln in outs[0][1].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line: %r" % (outs[0],)
    assert not [ln for ln in outs[1][1].splitlines() if ln.startswith("{")], "only rank 0 prints"
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_first_attempt_succeeds():
    d = _run("a2a/rccl-graph=ok")
    assert d["value"] == 123.0 and d["config"]["mode"] == "a2a" and d["config"]["fallback_reason"] is None
    assert [h["ok"] for h in d["config"]["attempts"]] == [True]
    # round 6: the attempt explains itself - seconds per progress mark of every rank, and the worker's per-rank diagnostics
    # (communicator creation time, bucket capacity / growth, microseconds per phase of the synchronous step) filed under the attempt
    at = d["config"]["attempts"][0]
    assert len(at["phase_seconds_per_rank"]) == 2
    for spans in at["phase_seconds_per_rank"]:
        assert list(spans) == ["start", "tables", "setup", "warmup", "timed", "headline"] and all(v >= 0 for v in spans.values())
    assert "diagnostics" not in d["config"] and len(at["diagnostics"]) == 2
    for r, dg in enumerate(at["diagnostics"]):
        assert dg["rank"] == r and dg["communicator_create_s"] is not None and "bucket_growth" in dg
        assert set(dg["phase_us_per_step"]) >= {"route", "ids_a2a", "gather", "rows_a2a", "compute", "push", "apply"}


@pytest.mark.timeout(300)
def test_hang_then_crash_then_one_stuck_rank_then_p2p():
    d = _run("a2a/rccl-graph=hang_setup,a2a/rccl=hang_setup,a2a/rccl-sync=hang_setup,a2a/torch=crash,p2p=ok")
    at = d["config"]["attempts"]
    assert [(h["mode"], h["comm"], h["ok"]) for h in at] == [("a2a", "rccl-graph", False), ("a2a", "rccl", False),
                                                             ("a2a", "rccl-sync", False), ("a2a", "torch", False), ("p2p", None, True)]
    assert "watchdog" in at[0]["why"] and "tables" in at[0]["why"] and "status 3" in at[3]["why"]
    assert list(at[0]["phase_seconds_per_rank"][0]) == ["start", "tables"]          # a failed attempt still says how far every rank got
    assert d["config"]["mode"] == "p2p" and "a2a/rccl" in d["config"]["fallback_reason"] and d["value"] == 123.0


@pytest.mark.timeout(300)
def test_one_stuck_rank_fails_the_attempt_and_replicas_close_the_chain():
    d = _run("a2a/rccl-graph=hang_rank1,a2a/rccl=hang_rank1,a2a/rccl-sync=ok_rank0_only,a2a/torch=hang_rank1,p2p=crash,replicas=ok")
    at = d["config"]["attempts"]
    assert [h["ok"] for h in at] == [False, False, False, False, False, True] and at[-1]["mode"] == "replicas"
    assert d["config"]["mode"] == "replicas" and d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["parallelism"] == "replicas only" and len(d["per_rank_us_per_step"]) == 2


@pytest.mark.timeout(300)
def test_a_leg_that_hangs_after_the_headline_costs_only_itself():
    d = _run("a2a/rccl-graph=leg_hang")
    assert d["value"] == 123.0 and d["config"]["mode"] == "a2a"
    assert d["config"]["attempts"][0]["ok"] and "watchdog" in (d["config"]["attempts"][0]["why"] or "")


@pytest.mark.timeout(300)
def test_every_attempt_failing_still_prints_a_line():
    d = _run("a2a/rccl-graph=crash,a2a/rccl=crash,a2a/rccl-sync=crash,a2a/torch=crash,p2p=crash,replicas=crash")
    assert d["value"] == 0.0 and d["config"]["fallback_reason"] == "every attempt failed" and len(d["config"]["attempts"]) == 6


def test_second_schedule_becomes_the_value_only_on_equal_terms():
    """bench_dist.second_schedule_wins: the schedule measured behind the delivered headline replaces its value only when it timed
    exactly the K steps of the line, carries its wall time, did not fail and is more than 2 % faster."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "dgl-ke_amd"))
    import bench_dist as bd
    leg = {"schedule": "overlapped", "value": 120.0, "steps": 20, "wall_s": 0.01}
    assert bd.second_schedule_wins(100.0, leg, 20)
    assert not bd.second_schedule_wins(118.0, leg, 20)                       # within 2 %
    assert not bd.second_schedule_wins(100.0, dict(leg, steps=240), 20)      # another number of steps than the line reports
    assert not bd.second_schedule_wins(100.0, dict(leg, wall_s=None), 20)
    assert not bd.second_schedule_wins(100.0, {"error": "x", "schedule": "overlapped"}, 20)
    assert not bd.second_schedule_wins(100.0, None, 20)
    os.environ["KGE_DIST_PROMOTE"] = "0"
    try:
        assert not bd.second_schedule_wins(100.0, leg, 20)
    finally:
        del os.environ["KGE_DIST_PROMOTE"]
