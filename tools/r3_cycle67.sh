#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 100 python tools/dbg/dbg_ew.py 2>&1 | grep -v amdgpu.ids | grep "bad\|rows\|sample\|Error\|error" | grep -v " bad 0" | head -20
timeout 900 python -m pytest tests -m gpu -q --timeout=300 2>&1 | grep -v "amdgpu.ids" | grep -v "^  File\|^Extension" | tail -14 > $O/c67_pytest.log; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/c67_pytest.log | tail -8
B="--no-cpu-baseline --hogwild 0 --no-async-update"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('%-30s wall %.3f events %.3f' % ('$1', 1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step']))"; }
for W in simple_fb15k complex_wikikg2; do timeout 100 python bench.py $B --workload $W 2>/dev/null | grep "^{" | p $W; done
