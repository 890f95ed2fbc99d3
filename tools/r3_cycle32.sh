#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 800 python -m pytest tests -m gpu -q --timeout=300 -x 2>&1 | grep -v "amdgpu.ids" | grep -v "^  File\|^Extension" | tail -8 > $O/c32_pytest.log; tail -3 $O/c32_pytest.log
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" 2> $O/c32_$n.err | grep "^{" | tail -1 > $O/c32_$n.json
  python -c "import json;d=json.load(open('$O/c32_$n.json'));print('%-34s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline'].get('event_ms_per_step', 0)))" || tail -3 $O/c32_$n.err
}
run rotate --workload rotate_fb15k
run l1 --workload transe_l1_fb15k
W=rotate_fb15k
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_w && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_w -- python $R/bench.py $B --steps 600 --warmup 120 --workload $W > /tmp/prof_w.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_w/*/*_results.db | head -1) | head -9 | cut -c1-64,73-118 | tee $O/c32_stats_$W.txt
