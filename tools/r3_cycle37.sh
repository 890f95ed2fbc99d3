#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_w && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_w -- python $R/bench.py $B --steps 600 --warmup 120 > /tmp/prof_w.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_w/*/*_results.db | head -1) | head -8 | cut -c1-64,73-118
cd $R
for i in 1 2 3 4 5 6; do timeout 100 python bench.py $B --steps 20 --warmup 5 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print('drv wall %.3f events %.3f' % (1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step']))"; done
