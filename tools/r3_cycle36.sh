#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_sampler.py -m gpu -q --timeout=200 -x 2>&1 | grep -v "amdgpu.ids" | grep -v "^  File\|^Extension" | tail -5
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tlm.so timeout 100 python tools/sampler_timeline.py 20 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tee $O/c36_sampler_timeline.txt
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" 2> $O/c36_$n.err | grep "^{" | tail -1 > $O/c36_$n.json
  python -c "import json;d=json.load(open('$O/c36_$n.json'));print('%-34s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline'].get('event_ms_per_step', 0)))" || tail -3 $O/c36_$n.err
}
run drv --steps 20 --warmup 5
run drv2 --steps 20 --warmup 5
run long
