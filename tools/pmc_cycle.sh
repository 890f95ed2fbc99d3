#!/bin/bash
# tools/pmc_cycle.sh TAG [bench args]: two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the bench
TAG=$1; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${TAG}_$c -- python $R/bench.py --no-cpu-baseline --no-configs --no-async-update --hogwild 0 --sampler-mode serial --steps 240 --warmup 120 --graph-steps 120 "$@" > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1
  db=$(ls /tmp/pmc_${TAG}_$c/*/*_results.db | head -1)
  cp $db $R/gpurun_out/${TAG}_pmc_$c.db
  python $R/tools/rocpd_stats.py $db --pmc > $R/gpurun_out/${TAG}_pmc_$c.txt 2>&1
  head -14 $R/gpurun_out/${TAG}_pmc_$c.txt
done
