#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 800 python -m pytest tests -m gpu -q --timeout=300 -x 2>&1 | grep -v "amdgpu.ids" | grep -v "^  File\|^Extension" | tail -8
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" 2> $O/c28_$n.err | grep "^{" | tail -1 > $O/c28_$n.json
  python -c "import json;d=json.load(open('$O/c28_$n.json'));print('%-34s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline'].get('event_ms_per_step', 0)))" || tail -3 $O/c28_$n.err
}
run rotate --workload rotate_fb15k
run l1 --workload transe_l1_fb15k
run rotfb --workload rotate_freebase
KGE_DIST_MODE=p2p run rotfb_p2p --workload rotate_freebase
