#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout=300 2>&1 | grep -v "amdgpu.ids" | grep -v "^  File\|^Extension" | tail -14 > $O/c65_pytest.log; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/c65_pytest.log | tail -12
