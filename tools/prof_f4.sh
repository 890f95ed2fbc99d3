#!/bin/bash
# rocprofv3 kernel stats of the (f4) models' steps: TransR and RESCAL at their FB15k recipes
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/f4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for W in ${WORKLOADS:-transr_fb15k rescal_fb15k}; do
  rm -rf /tmp/prof_f; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -- python $R/bench.py --workload $W --no-cpu-baseline --no-configs --hogwild 0 --no-async-update --steps 200 --warmup 20 > $O/prof_$W.log 2>&1
  python $R/tools/rocpd_stats.py $(ls /tmp/prof_f/*/*_results.db | head -1) > $O/kernel_stats_$W.txt 2>&1; head -24 $O/kernel_stats_$W.txt | cut -c1-150
  grep '^{' $O/prof_$W.log | tail -1 | cut -c1-260
done
