#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
KGE_DIST_FORCE_COLL=1 KGE_DIST_GRAPH=1 KGE_DIST_GRAPH_TIMEOUT=60 timeout 150 python bench.py $B --workload rotate_freebase --steps 240 --warmup 40 2> $O/c34_graph.err | grep "^{" | tail -1 > $O/c34_graph.json
python - <<'PY'
import json,os
p=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/c34_graph.json"
try:
    d=json.load(open(p)); print("wall %.3f us" % (1e3*d["ms_per_step"])); print(d.get("a2a_eager")); print(d["config"]["workload"][-300:])
except Exception as e:
    print("no line:", e); print(open(p.replace(".json",".err")).read()[-1500:])
PY
