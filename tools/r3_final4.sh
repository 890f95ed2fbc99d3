#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R; export GRAFT_REPO_ROOT=$R
timeout 900 python -m pytest tests -m gpu -q --timeout=300 2>&1 | grep -v "amdgpu.ids" | grep -v "^  File\|^Extension" | tail -8 > $O/r03_v2_pytest.log; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/r03_v2_pytest.log | tail -3
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/r03_v2_bench_driver_like.err | grep "^{" | tail -1 > $O/r03_v2_bench_driver_like.json
python -c "
import json,os
d=json.load(open('$O/r03_v2_bench_driver_like.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('event_ms_per_step'))"
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -1
