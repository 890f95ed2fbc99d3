#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03f}
timeout 900 python -m pytest tests -m gpu -q --timeout=180 -x 2>&1 | grep -v "amdgpu.ids" | tail -12 > $O/${TAG}_pytest.log; tail -6 $O/${TAG}_pytest.log
B="--no-cpu-baseline --hogwild 0 --no-async-update --sampler-mode serial"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" > $O/${TAG}_$n.json 2> $O/${TAG}_$n.err
  python -c "import json;d=json.load(open('$O/${TAG}_$n.json'));print('%-28s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline']['event_ms_per_step']))"
}
run default
run bwd_direct --flags 1024
run split_bwd_direct --flags 1152
run default_2
run default_drv --steps 20 --warmup 5
run distmult --workload distmult_fb15k
run distmult_bwd_direct --workload distmult_fb15k --flags 1024
run complex --workload complex_wikikg2
run complex_bwd_direct --workload complex_wikikg2 --flags 1024
run simple --workload simple_fb15k
run simple_bwd_direct --workload simple_fb15k --flags 1024
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 100 python tools/timeline.py > $O/${TAG}_timeline.txt 2>&1; tail -8 $O/${TAG}_timeline.txt
