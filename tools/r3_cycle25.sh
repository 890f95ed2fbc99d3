#!/bin/bash
# pairwise forward: rows per wavefront x runs per task (TA-bound per-lane row loads vs occupancy)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" 2> $O/c25_$n.err | grep "^{" | tail -1 > $O/c25_$n.json
  python -c "import json;d=json.load(open('$O/c25_$n.json'));print('%-34s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline'].get('event_ms_per_step', 0)))" || tail -3 $O/c25_$n.err
}
for V in "" rw8ks8 rw8ks4; do
  if [ -n "$V" ]; then export KGE_LIB=$R/dgl-ke_amd/variants/libkge_$V.so; else unset KGE_LIB; fi
  run rotate_$V --workload rotate_fb15k
  run l1_$V --workload transe_l1_fb15k
  run l1split_$V --workload transe_l1_fb15k --flags 128
  run rotfb_$V --workload rotate_freebase
done
unset KGE_LIB
for V in rw8ks8; do
export KGE_LIB=$R/dgl-ke_amd/variants/libkge_$V.so
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_w && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_w -- python $R/bench.py $B --steps 600 --warmup 120 --workload rotate_fb15k > /tmp/prof_w.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_w/*/*_results.db | head -1) | head -5 | cut -c1-64,73-118
cd $R
done
