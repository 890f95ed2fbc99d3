"""scripted stand-in for the GPU worker of bench_dist.orchestrate (tests/test_bench_supervisor.py): behaves per attempt as the
KGE_FAKE_PLAN environment variable says - "mode/comm=behaviour,..." with behaviour in
  hang_setup   marks 'start', then never returns           (a collective that deadlocks during set-up)
  crash        exits with status 3 after 'start'
  hang_rank1   rank 1 hangs after 'setup', rank 0 runs through   (one rank stuck: the other ranks' lines must not count)
  leg_hang     delivers the headline, then hangs             (a secondary leg that never ends)
  ok_rank0_only rank 1 crashes, rank 0 delivers a line       (the line of a rank whose peers failed must not count)
  ok           all marks, delivers a line"""
import json
import os
import sys
import time

mode, comm = os.environ["KGE_DIST_MODE"], os.environ.get("KGE_DIST_ATTEMPT", os.environ.get("KGE_DIST_COMM", ""))
rank = int(os.environ["RANK"])
plan = dict(item.split("=") for item in os.environ["KGE_FAKE_PLAN"].split(","))
what = plan.get("%s/%s" % (mode, comm), plan.get(mode, "ok"))


def mark(m):
    with open(os.environ["KGE_DIST_PROGRESS"], "a") as f:
        f.write("%s %.3f\n" % (m, time.time()))


def deliver(obj):
    with open(os.environ["KGE_DIST_RESULT"], "w") as f:
        f.write(json.dumps(obj) + "\n")


mark("start")
mark("tables")
if what == "hang_setup":
    time.sleep(3600)
if what == "crash" or (what == "ok_rank0_only" and rank == 1):
    sys.exit(3)
mark("setup")
if what == "hang_rank1" and rank == 1:
    time.sleep(3600)
mark("warmup")
mark("timed")
if mode == "replicas":
    deliver({"wall": 0.002 * (rank + 1), "rank": rank})
elif rank == 0:
    deliver({"metric": "positive edges/sec (whole node)", "value": 123.0, "unit": "edges/s",
             "config": {"mode": mode, "fallback_reason": None, "comm": comm,
                        # (what the real worker's diagnostics leg adds: per rank communicator creation time, bucket growth, phase us)
                        "diagnostics": [{"rank": r, "communicator": "RcclComm", "communicator_create_s": 0.5, "bucket_rows": 512,
                                         "bucket_growth": None, "phase_us_per_step": {"route": 1.0, "ids_a2a": 2.0, "gather": 3.0,
                                                                                      "rows_a2a": 4.0, "compute": 90.0, "push": 5.0,
                                                                                      "apply": 6.0, "steps": 20}} for r in range(2)]}})
mark("headline")
if what == "leg_hang":
    time.sleep(3600)
