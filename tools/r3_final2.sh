#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R; export GRAFT_REPO_ROOT=$R
bash tools/other_workloads.sh r03_v2 > /dev/null 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/r03_v2_workloads.txt | grep "^==\|edges/s" 
for M in a2a p2p; do
KGE_DIST_MODE=$M timeout 300 python bench.py --gpus 1 --workload rotate_freebase --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > $O/r03_v2_bench_rotate_freebase_n1_$M.json
python -c "import json;d=json.load(open('$O/r03_v2_bench_rotate_freebase_n1_$M.json'));print('$M', d['value'], d['ms_per_step'])"
done
