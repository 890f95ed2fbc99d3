#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03t}
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -x 2>&1 | grep -v "amdgpu.ids" | tail -30 > $O/${TAG}_pytest.log; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/${TAG}_pytest.log | tail -4
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" > $O/${TAG}_$n.json 2> $O/${TAG}_$n.err
  python -c "import json;d=json.load(open('$O/${TAG}_$n.json'));print('%-34s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline']['event_ms_per_step']))" || tail -3 $O/${TAG}_$n.err
}
run l1 --workload transe_l1_fb15k
run l1_split --workload transe_l1_fb15k --flags 128
