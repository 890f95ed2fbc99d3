#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1 wall %.3f events %.3f' % (1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step']))"; }
for W in 5 120 600 2400; do timeout 100 python bench.py $B --steps 20 --warmup $W 2>/dev/null | grep "^{" | p dev_K20_W$W; done
for W in 5 600; do timeout 100 python bench.py $B --steps 120 --warmup $W 2>/dev/null | grep "^{" | p dev_K120_W$W; done
for W in 5 600; do timeout 100 python bench.py $B --steps 20 --warmup $W --host-plan 2>/dev/null | grep "^{" | p host_K20_W$W; done
