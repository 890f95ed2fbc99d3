#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03l}
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -x 2>&1 | grep -v "amdgpu.ids" | tail -30 > $O/${TAG}_pytest.log; tail -4 $O/${TAG}_pytest.log
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" > $O/${TAG}_$n.json 2> $O/${TAG}_$n.err
  python -c "import json;d=json.load(open('$O/${TAG}_$n.json'));print('%-28s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline']['event_ms_per_step']))"
}
run rotate --workload rotate_fb15k
run rotate_split --workload rotate_fb15k --flags 128
run l1 --workload transe_l1_fb15k
for m in a2a p2p; do
KGE_DIST_MODE=$m timeout 300 python bench.py --gpus 1 --workload rotate_freebase --steps 1200 --warmup 120 > $O/${TAG}_${m}_rf.json 2> $O/${TAG}_${m}_rf.err
tail -1 $O/${TAG}_${m}_rf.json | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$m rotate_freebase', 1e3*d['ms_per_step'], 'us/step', d['value'])" || tail -5 $O/${TAG}_${m}_rf.err
done
