#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1 wall %.3f events %.3f' % (1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step']))"; }
for i in 1 2; do timeout 100 python bench.py $B --steps 20 --warmup 5 --no-graph 2>/dev/null | grep "^{" | p eager_K20; done
timeout 100 python bench.py $B --no-graph 2>/dev/null | grep "^{" | p eager_long
timeout 100 python bench.py $B --steps 20 --warmup 5 --graph-steps 4 2>/dev/null | grep "^{" | p graph4_K20
timeout 100 python bench.py $B --steps 20 --warmup 5 --graph-steps 10 2>/dev/null | grep "^{" | p graph10_K20
timeout 100 python bench.py $B --graph-steps 10 2>/dev/null | grep "^{" | p graph10_long
timeout 100 python bench.py $B --graph-steps 20 2>/dev/null | grep "^{" | p graph20_long
