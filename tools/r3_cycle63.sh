#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_async.py tests/test_gpu_reference_style.py tests/test_gpu_dist.py -m gpu -q --timeout=300 -x 2>&1 | grep -v "amdgpu.ids" | grep -v "^  File\|^Extension" | tail -12 > $O/c63_pytest.log; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/c63_pytest.log | tail -10
B="--no-cpu-baseline --hogwild 0 --no-async-update"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('%-30s wall %.3f events %.3f' % ('$1', 1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step']))"; }
for W in distmult_fb15k transe_l2_fb15k complex_wikikg2; do timeout 100 python bench.py $B --workload $W 2>/dev/null | grep "^{" | p $W; done
