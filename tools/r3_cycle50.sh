#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 500 python -m pytest tests/test_gpu_cli.py tests/test_gpu_dist.py -m gpu -q --timeout=300 -x -k "multi_process or dist" 2>&1 | grep -v "amdgpu.ids" | tail -25 > $O/c50_pytest.log; tail -25 $O/c50_pytest.log
