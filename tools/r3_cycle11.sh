#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03k}
timeout 900 python -m pytest tests -m gpu -q --timeout=300 2>&1 | grep -v "amdgpu.ids" | tail -30 > $O/${TAG}_pytest.log; tail -12 $O/${TAG}_pytest.log
cat $O/row_error_trajectory.txt
