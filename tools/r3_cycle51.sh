#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
for W in transe_l1_fb15k rotate_fb15k; do
echo "=== $W"
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tlm.so timeout 200 python tools/timeline.py --workload $W 2>&1 | grep -v "amdgpu.ids\|socket.cpp"
done | tee $O/c51_pairwise_timeline.txt
