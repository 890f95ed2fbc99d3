#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1 wall %.3f events %.3f' % (1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step']))"; }
for i in 1 2 3; do timeout 100 python bench.py $B --steps 20 --warmup 5 2>/dev/null | grep "^{" | p devsync_K20; done
for i in 1 2 3; do KGE_BENCH_STREAM_SYNC_FIRST=1 timeout 100 python bench.py $B --steps 20 --warmup 5 2>/dev/null | grep "^{" | p streamsync_K20; done
