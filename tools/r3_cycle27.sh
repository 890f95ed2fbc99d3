#!/bin/bash
# RotatE backward with the negatives split over workgroups (GA in parts) + TransE_l1 forward shapes
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_style.py tests/test_gpu_async.py -m gpu -q --timeout=300 -x 2>&1 | grep -v "amdgpu.ids" | tail -8
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" 2> $O/c27_$n.err | grep "^{" | tail -1 > $O/c27_$n.json
  python -c "import json;d=json.load(open('$O/c27_$n.json'));print('%-34s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline'].get('event_ms_per_step', 0)))" || tail -3 $O/c27_$n.err
}
run rotate --workload rotate_fb15k
run rotfb --workload rotate_freebase
for V in "" l1a l1b l1c; do
  if [ -n "$V" ]; then export KGE_LIB=$R/dgl-ke_amd/variants/libkge_$V.so; else unset KGE_LIB; fi
  run l1_$V --workload transe_l1_fb15k
  run l1split_$V --workload transe_l1_fb15k --flags 128
done
unset KGE_LIB
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_w && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_w -- python $R/bench.py $B --steps 600 --warmup 120 --workload rotate_fb15k > /tmp/prof_w.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_w/*/*_results.db | head -1) | head -8 | cut -c1-64,73-118 | tee $O/c27_stats_rotate.txt
