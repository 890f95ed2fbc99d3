#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
KGE_DIST_FORCE_COLL=1 KGE_DIST_OTHER_LEG=force timeout 280 python bench.py $B --workload rotate_freebase --steps 240 --warmup 40 2> $O/c35.err | grep "^{" | tail -1 > $O/c35.json
python - <<'PY'
import json,os
p=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/c35.json"
try:
    d=json.load(open(p)); print("wall %.3f us" % (1e3*d["ms_per_step"])); print("p2p:", {k:v for k,v in d.get("p2p",{}).items() if k!="desc"}); print("local:", d.get("per_gpu_step_without_exchange"))
except Exception as e:
    print("no line:", e); print(open(p.replace(".json",".err")).read()[-2500:])
PY
