#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R; export GRAFT_REPO_ROOT=$R
timeout 900 python -m pytest tests -m gpu -q --timeout=300 2>&1 | grep -v "amdgpu.ids" | grep -v "^  File\|^Extension" | tail -8 > $O/r03_v2_pytest.log; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/r03_v2_pytest.log | tail -3
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/r03_v2_bench_driver_like.err | grep "^{" | tail -1 > $O/r03_v2_bench_driver_like.json
timeout 400 python bench.py 2> $O/r03_v2_bench_long.err | grep "^{" | tail -1 > $O/r03_v2_bench_transe_l2_fb15k.json
python - <<'PY'
import json,os
R=os.environ["GRAFT_REPO_ROOT"]
for n in ("r03_v2_bench_driver_like","r03_v2_bench_transe_l2_fb15k"):
    d=json.load(open(R+"/gpurun_out/%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("event_ms_per_step"), {k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.items() if k in ("hogwild","async_update","async_update_rel","heavy_tailed_ids","cpu_baseline")})
PY
bash tools/final_profiles.sh r03_v2 2>&1 | tail -24
