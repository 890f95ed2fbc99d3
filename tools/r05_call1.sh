#!/bin/bash
# round 5, GPU call 1: changed tests, RCCL capture matrix, forced-collective proxy (eager / graph with NCCL_GRAPH_MIXING_SUPPORT=0)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05c1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -m gpu -q --timeout=300 -x \
    -k "world2 or world4 or config_shapes or real_table or route or capacity or precaptured" 2>&1 | tail -15 ) > $O/pytest.log
tail -4 $O/pytest.log
timeout 420 python tools/dbg/rccl_capture_matrix.py > $O/rccl_matrix.jsonl 2> $O/rccl_matrix.err
cat $O/rccl_matrix.jsonl | cut -c1-260
for v in eager graph graph_nomix; do
  case $v in
    eager) EXTRA="";;
    graph) EXTRA="KGE_DIST_GRAPH=1 KGE_DIST_GRAPH_TIMEOUT=40";;
    graph_nomix) EXTRA="KGE_DIST_GRAPH=1 KGE_DIST_GRAPH_TIMEOUT=40 NCCL_GRAPH_MIXING_SUPPORT=0";;
  esac
  env KGE_DIST_FORCE_COLL=1 KGE_DIST_MODE=a2a KGE_DIST_OTHER_LEG=0 $EXTRA timeout 200 python bench.py --workload rotate_freebase --steps 600 --warmup 40 --no-cpu-baseline > $O/proxy_$v.json 2> $O/proxy_$v.err
  echo "== proxy $v rc=$?"; python - <<P
import json
try:
    d=json.loads(open("$O/proxy_$v.json").read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step")}, d.get("a2a_eager"), d["config"].get("launch","")[:200] if isinstance(d.get("config"),dict) else None)
except Exception as e:
    print("no line:", e); print(open("$O/proxy_$v.err").read()[-800:])
P
done
