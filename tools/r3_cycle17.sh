#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03q}
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" > $O/${TAG}_$n.json 2> $O/${TAG}_$n.err
  python -c "import json;d=json.load(open('$O/${TAG}_$n.json'));print('%-34s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline']['event_ms_per_step']))" || tail -3 $O/${TAG}_$n.err
}
run base
for v in fu3 fu4 bu2; do KGE_LIB=$R/dgl-ke_amd/variants/libkge_$v.so run $v; done
run base2
