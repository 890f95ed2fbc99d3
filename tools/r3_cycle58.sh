#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_cli.py -m gpu -q --timeout=200 -x -k "sharded or dist or multi_process" 2>&1 | grep -v "amdgpu.ids" | grep -v "^  File\|^Extension" | tail -12 > $O/c58_pytest.log; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/c58_pytest.log | tail -8
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 300 env "$@" python bench.py $B --workload rotate_freebase 2> $O/c58_$n.err | grep "^{" | tail -1 > $O/c58_$n.json
  python -c "import json;d=json.load(open('$O/c58_$n.json'));print('%-34s wall %.3f us' % ('$n', 1e3*d['ms_per_step']))" || tail -5 $O/c58_$n.err
}
run coll_rccl KGE_DIST_FORCE_COLL=1
run coll_rccl_sync KGE_DIST_FORCE_COLL=1 KGE_DIST_PIPELINE=0
run coll_torch KGE_DIST_FORCE_COLL=1 KGE_DIST_COMM=torch
