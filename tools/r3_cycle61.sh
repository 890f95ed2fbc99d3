#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update --steps 20 --warmup 5"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('%-30s wall %.3f events %.3f loss %.6f' % ('$1', 1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step'], d.get('mean_loss',0)))"; }
for i in 1 2 3 4; do timeout 100 python bench.py $B 2>/dev/null | grep "^{" | p drv; done
timeout 100 python bench.py --no-cpu-baseline --hogwild 0 --no-async-update 2>/dev/null | grep "^{" | p long
timeout 300 python -m pytest tests/test_gpu_bench.py -m gpu -q --timeout=200 -x 2>&1 | grep -v "amdgpu.ids" | tail -3
