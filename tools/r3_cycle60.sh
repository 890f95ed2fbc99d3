#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update --steps 20 --warmup 5"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('%-30s wall %.3f events %.3f' % ('$1', 1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step']))"; }
for P in 0 fill mm 0 fill mm; do KGE_BENCH_PREROLL=$P timeout 100 python bench.py $B 2>/dev/null | grep "^{" | p preroll_$P; done
