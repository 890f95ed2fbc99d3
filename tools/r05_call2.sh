#!/bin/bash
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05c2; mkdir -p $O
for v in - pg pg,pipe pg,pipe,insample pipe,insample pg,pipe,insample,relaxed pg,insample,torchcomm; do
  n=$(echo $v | tr ',' '_')
  timeout -s KILL 60 python tools/dbg/dist_graph_probe.py $v > $O/probe_$n.log 2>&1
  echo "== $v rc=$?"; grep -v "amdgpu.ids" $O/probe_$n.log | grep "^\[\|DONE\|Error\|error" | tail -4
done
( timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -m gpu -q --timeout=300 \
    -k "world2 or world4 or config_shapes or real_table or route or capacity or precaptured" 2>&1 | tail -15 ) > $O/pytest.log
tail -6 $O/pytest.log
