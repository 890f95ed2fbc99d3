#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
KGE_LIB=$R/dgl-ke_amd/variants/libkge_rs4.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=300 -x 2>&1 | grep -v "amdgpu.ids" | grep -v "^  File\|^Extension" | tail -5
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" 2> $O/c30_$n.err | grep "^{" | tail -1 > $O/c30_$n.json
  python -c "import json;d=json.load(open('$O/c30_$n.json'));print('%-34s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline'].get('event_ms_per_step', 0)))" || tail -3 $O/c30_$n.err
}
for V in "" rs4; do
  if [ -n "$V" ]; then export KGE_LIB=$R/dgl-ke_amd/variants/libkge_$V.so; else unset KGE_LIB; fi
  run rotate_$V --workload rotate_fb15k
  run l1_$V --workload transe_l1_fb15k
  run rotfb_$V --workload rotate_freebase
done
unset KGE_LIB
for V in tl rs4tl; do for W in rotate_fb15k transe_l1_fb15k; do
echo "=== variant $V $W"
KGE_LIB=$R/dgl-ke_amd/variants/libkge_$V.so timeout 200 python tools/timeline.py --workload $W 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | grep "kernel\|neg_\|span"
done; done | tee $O/c30_timeline.txt
