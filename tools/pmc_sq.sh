#!/bin/bash
# tools/pmc_sq.sh "COUNTERS" [bench args]: one PMC pass with the given SQ counters, per-kernel sums
CTRS=$1; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_sq -- python $R/bench.py --no-cpu-baseline --no-configs --hogwild 0 --steps 240 --warmup 120 --graph-steps 120 "$@" > /tmp/pmc_sq.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/pmc_sq/*/*_results.db | head -1) --pmc | head -${PMC_HEAD:-40}
