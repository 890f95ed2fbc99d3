#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03c}
B="--no-cpu-baseline --hogwild 0 --no-async-update --steps 1200 --warmup 120"
for m in serial streams; do
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$m
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -- python $R/bench.py $B --sampler-mode $m > $O/${TAG}_prof_$m.log 2>&1
echo "== $m"; tail -1 $O/${TAG}_prof_$m.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('us/step', 1e3*d['ms_per_step'])"
python $R/tools/kernel_gaps.py $(ls /tmp/prof_$m/*/*_results.db | head -1) | tee $O/${TAG}_gaps_$m.txt
done
