#!/bin/bash
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; O=gpurun_out/r05c9; mkdir -p $O
bash tools/build_variant.sh tl -DKGE_TIMELINE > $O/build.log 2>&1; tail -1 $O/build.log
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 200 python tools/timeline.py --workload transe_l2_fb15k --graph-steps 10 --tail > $O/timeline_tail.txt 2>&1
grep -v amdgpu.ids $O/timeline_tail.txt
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 200 python tools/timeline.py --workload transe_l2_fb15k --graph-steps 10 > $O/timeline_notail.txt 2>&1
grep -v amdgpu.ids $O/timeline_notail.txt
