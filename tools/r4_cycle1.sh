#!/bin/bash
# round 4, GPU call 1: the 3-launch strict step (loss rows inside the first launch) - parity, A/B bench, timelines
[ -z "$GRAFT_REPO_ROOT" ] && export GRAFT_REPO_ROOT=$(pwd)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -q --timeout=200 -k "loss_rows or hand_off or merged_forward or four_phase or deterministic" 2>&1 | tail -25 > $O/r4c1_newtests.log
tail -8 $O/r4c1_newtests.log
B="--no-cpu-baseline --hogwild 0 --no-async-update"
for f in 0 1024; do
  timeout 120 python bench.py $B --steps 3000 --warmup 300 --flags $f > $O/r4c1_long_f$f.json 2> $O/r4c1_long_f$f.err
  python - <<PY
import json
d=json.loads(open("$O/r4c1_long_f$f.json").read().strip().splitlines()[-1])
print("long flags=$f", d["ms_per_step"]*1e3, "us/step", d["value"])
PY
  for k in 1 2; do
    timeout 120 python bench.py $B --steps 20 --warmup 5 --flags $f > $O/r4c1_drv_f${f}_$k.json 2>> $O/r4c1_long_f$f.err
    python - <<PY
import json
d=json.loads(open("$O/r4c1_drv_f${f}_$k.json").read().strip().splitlines()[-1])
print("driver-shape flags=$f", d["ms_per_step"]*1e3, "us/step", d["value"])
PY
  done
done
for f in 0 1024; do
  KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 120 python tools/timeline.py --flags $f > $O/r4c1_timeline_f$f.txt 2>&1
  head -40 $O/r4c1_timeline_f$f.txt
done
timeout 600 python -m pytest tests -m gpu -q --timeout=300 2>&1 | tail -25 > $O/r4c1_pytest.log; tail -12 $O/r4c1_pytest.log
