#!/bin/bash
# round-3 GPU cycle 2: sampler modes x forward variants
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03b}
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { # name args
  n=$1; shift
  timeout 200 python bench.py $B "$@" > $O/${TAG}_$n.json 2> $O/${TAG}_$n.err
  python -c "import json;d=json.load(open('$O/${TAG}_$n.json'));print('%-28s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline']['event_ms_per_step']))"
}
for m in serial streams fork; do
  run drv_${m}_merged --steps 20 --warmup 5 --sampler-mode $m
  run drv_${m}_split --steps 20 --warmup 5 --sampler-mode $m --flags 128
  run long_${m}_merged --sampler-mode $m
  run long_${m}_split --sampler-mode $m --flags 128
done
run drv_streams_merged_2 --steps 20 --warmup 5
run drv_streams_merged_3 --steps 20 --warmup 5
