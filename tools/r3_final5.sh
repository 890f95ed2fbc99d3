#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R; export GRAFT_REPO_ROOT=$R
bash tools/other_workloads.sh r03_v3 > /dev/null 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/r03_v3_workloads.txt | grep "^==\|edges/s" | head -20
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/r03_v3_bench_driver_like.err | grep "^{" | tail -1 > $O/r03_v3_bench_driver_like.json
python -c "
import json
d=json.load(open('$O/r03_v3_bench_driver_like.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
