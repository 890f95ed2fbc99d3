#!/bin/bash
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; O=gpurun_out/r05c12; mkdir -p $O
for m in serial fused; do
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$m -- python $R/bench.py --no-cpu-baseline --no-configs --no-async-update --hogwild 0 --steps 1200 --warmup 120 --sampler-mode $m > $R/$O/prof_$m.log 2>&1
cd $R && python tools/rocpd_stats.py $(ls $O/prof_$m/*/*_results.db | head -1) > $O/kernel_stats_$m.txt 2>&1; echo "== $m"; head -8 $O/kernel_stats_$m.txt | cut -c1-150
rm -rf $O/prof_$m
done
