#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('%-44s wall %.3f events %.3f' % ('$1', 1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step']))"; }
run() { n="$1"; shift; timeout 100 env "$@" python bench.py $B 2>/dev/null | grep "^{" | p "$n"; }
B="$B --steps 20 --warmup 5"
run sampler_behind A=1
run sampler_behind A=1
run upload KGE_GRAPH_UPLOAD=1
run upload KGE_GRAPH_UPLOAD=1
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run long A=1
timeout 300 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_bench.py -m gpu -q --timeout=200 -x 2>&1 | grep -v "amdgpu.ids" | tail -3
