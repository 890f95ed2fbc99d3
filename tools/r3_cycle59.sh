#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('%-30s wall %.3f events %.3f' % ('$1', 1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step']))"; }
for W in 5 5 120 600; do timeout 100 python bench.py $B --steps 20 --warmup $W 2>/dev/null | grep "^{" | p K20_W$W; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_w && timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_w -- python $R/bench.py $B --steps 20 --warmup 120 > /tmp/prof_w.log 2>&1
python $R/tools/first_steps.py $(ls /tmp/prof_w/*/*_results.db | head -1) 20 | tail -18
