#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03g}
python tools/dbg/dbg_bwd_lds.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_dbg.txt
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tlm.so timeout 100 python tools/timeline.py --flags 1024 > $O/${TAG}_timeline_marks_direct.txt 2>&1; tail -12 $O/${TAG}_timeline_marks_direct.txt
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tlm.so timeout 100 python tools/timeline.py > $O/${TAG}_timeline_marks_lds.txt 2>&1; tail -12 $O/${TAG}_timeline_marks_lds.txt
