#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1 wall %.3f events %.3f' % (1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step']))"; }
for i in 1 2; do timeout 100 python bench.py $B --steps 20 --warmup 5 --host-plan 2>/dev/null | grep "^{" | p hostplan_drv; done
timeout 100 python bench.py $B --host-plan 2>/dev/null | grep "^{" | p hostplan_long
for K in 40 80 120 240; do timeout 100 python bench.py $B --steps $K --warmup 5 2>/dev/null | grep "^{" | p dev_steps$K; done
for K in 40 120; do timeout 100 python bench.py $B --steps $K --warmup 5 --host-plan 2>/dev/null | grep "^{" | p host_steps$K; done
