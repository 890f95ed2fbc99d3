#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03m}
python -c "import torch;print('priority range', torch.cuda.Stream.priority_range())"
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" > $O/${TAG}_$n.json 2> $O/${TAG}_$n.err
  python -c "import json;d=json.load(open('$O/${TAG}_$n.json'));print('%-28s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline']['event_ms_per_step']))" || tail -3 $O/${TAG}_$n.err
}
run serial_drv --steps 20 --warmup 5
run serial_long
for p in 1 0 -1; do
KGE_SAMPLER_STREAM_PRIORITY=$p run streams_p${p}_drv --steps 20 --warmup 5 --sampler-mode streams
KGE_SAMPLER_STREAM_PRIORITY=$p run streams_p${p}_long --sampler-mode streams
done
