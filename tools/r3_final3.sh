#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R; export GRAFT_REPO_ROOT=$R
timeout 400 bash tools/time_to_mrr.sh 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -40 > $O/r03_v2_time_to_mrr.txt; grep -i "mrr\|takes\|reach\|target\|steps take" $O/r03_v2_time_to_mrr.txt | tail -12
timeout 900 python -m pytest tests -m gpu -q --timeout=300 2>&1 | grep -v "amdgpu.ids" | grep -v "^  File\|^Extension" | tail -8 > $O/r03_v2_pytest.log; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/r03_v2_pytest.log | tail -3
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -2
