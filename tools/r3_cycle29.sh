#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
for V in tl tlm; do
echo "=== variant $V"
KGE_LIB=$R/dgl-ke_amd/variants/libkge_$V.so timeout 200 python tools/timeline.py --workload rotate_fb15k 2>&1 | grep -v "amdgpu.ids\|socket.cpp"
done | tee $O/c29_rotate_timeline.txt
