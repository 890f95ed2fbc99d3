#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" 2> $O/c31_$n.err | grep "^{" | tail -1 > $O/c31_$n.json
  python -c "import json;d=json.load(open('$O/c31_$n.json'));print('%-34s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline'].get('event_ms_per_step', 0)))" || tail -3 $O/c31_$n.err
}
for V in "" rs4 rs4c2 c2 c3; do
  if [ -n "$V" ]; then export KGE_LIB=$R/dgl-ke_amd/variants/libkge_$V.so; else unset KGE_LIB; fi
  run rotate_$V --workload rotate_fb15k
  run rotfb_$V --workload rotate_freebase
done
unset KGE_LIB
run l1 --workload transe_l1_fb15k
