#!/bin/bash
# pairwise forward with four wavefronts per SIMD: parity subset, step times, kernel stats
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-c24}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_style.py -m gpu -q --timeout=300 -x 2>&1 | grep -v "amdgpu.ids" | tail -8
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" > $O/${TAG}_$n.json 2> $O/${TAG}_$n.err
  python -c "import json;d=json.load(open('$O/${TAG}_$n.json'));print('%-34s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline']['event_ms_per_step']))" || tail -3 $O/${TAG}_$n.err
}
run rotate --workload rotate_fb15k
run l1 --workload transe_l1_fb15k
run rotate_fb --workload rotate_freebase
for W in rotate_fb15k transe_l1_fb15k; do
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_w && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_w -- python $R/bench.py $B --steps 600 --warmup 120 --workload $W > /tmp/prof_w.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_w/*/*_results.db | head -1) | head -9 | cut -c1-64,73-118 | tee $O/${TAG}_stats_$W.txt
cd $R
done
