#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1 wall %.3f events %.3f' % (1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step']))"; }
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^$" | head -20
timeout 100 python bench.py $B --steps 20 --warmup 5 2>/dev/null | grep "^{" | p auto_K20
rocm-smi --setperflevel high 2>&1 | tail -3
rocm-smi --showperflevel 2>&1 | grep -i perf
timeout 100 python bench.py $B --steps 20 --warmup 5 2>/dev/null | grep "^{" | p high_K20
timeout 100 python bench.py $B --steps 20 --warmup 5 2>/dev/null | grep "^{" | p high_K20
timeout 100 python bench.py $B 2>/dev/null | grep "^{" | p high_long
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_w && timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_w -- python $R/bench.py $B --steps 20 --warmup 5 > /tmp/prof_w.log 2>&1
python $R/tools/first_steps.py $(ls /tmp/prof_w/*/*_results.db | head -1) 20 | tail -16
rocm-smi --setperflevel auto 2>&1 | tail -2
