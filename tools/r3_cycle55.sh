#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('%-30s wall %.3f' % ('$1', 1e3*d['ms_per_step']))"; }
for V in "" rwr8 rwr5 rwr6; do
  if [ -n "$V" ]; then export KGE_LIB=$R/dgl-ke_amd/variants/libkge_$V.so; else unset KGE_LIB; fi
  timeout 100 python bench.py $B --workload transe_l1_fb15k 2>/dev/null | grep "^{" | p "l1_$V"
done
