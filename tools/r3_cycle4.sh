#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03d}
timeout 900 python -m pytest tests -m gpu -q --timeout=180 -x 2>&1 | grep -v "amdgpu.ids" | tail -6 > $O/${TAG}_pytest.log; tail -3 $O/${TAG}_pytest.log
B="--no-cpu-baseline --hogwild 0 --no-async-update --sampler-mode serial"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" > $O/${TAG}_$n.json 2> $O/${TAG}_$n.err
  python -c "import json;d=json.load(open('$O/${TAG}_$n.json'));print('%-28s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline']['event_ms_per_step']))"
}
run nb2_dense
run nb2_gather --flags 256
run split --flags 128
KGE_LIB=$R/dgl-ke_amd/variants/libkge_nb1.so run nb1_dense
KGE_LIB=$R/dgl-ke_amd/variants/libkge_nb3.so run nb3_dense
run nb2_dense_drv --steps 20 --warmup 5
run distmult --workload distmult_fb15k
run distmult_split --workload distmult_fb15k --flags 128
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 100 python tools/timeline.py > $O/${TAG}_timeline.txt 2>&1; tail -8 $O/${TAG}_timeline.txt
