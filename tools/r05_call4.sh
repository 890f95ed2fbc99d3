#!/bin/bash
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05c4; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -q --timeout=300 2>&1 | tail -15 ) > $O/pytest.log
tail -6 $O/pytest.log
for v in graph_sync graph_pipe eager_pipe; do
  case $v in
    graph_sync) EXTRA="KGE_DIST_GRAPH=1 KGE_DIST_PIPELINE=0";;
    graph_pipe) EXTRA="KGE_DIST_GRAPH=1 KGE_DIST_PIPELINE=1";;
    eager_pipe) EXTRA="KGE_DIST_GRAPH=0 KGE_DIST_PIPELINE=1";;
  esac
  env KGE_FAULTHANDLER=80 KGE_DIST_FORCE_COLL=1 KGE_DIST_MODE=a2a KGE_DIST_OTHER_LEG=0 $EXTRA timeout -s KILL 150 python bench.py --workload rotate_freebase --steps 600 --warmup 40 --no-cpu-baseline > $O/proxy_$v.json 2> $O/proxy_$v.err
  echo "== proxy $v rc=$?"; python - <<P
import json
try:
    d=json.loads(open("$O/proxy_$v.json").read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step")}, d["config"].get("launch"), d["config"]["workload"][-330:])
except Exception as e:
    print("no line:", e); print(open("$O/proxy_$v.err").read()[-1500:])
P
done
