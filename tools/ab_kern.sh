#!/bin/bash
# tools/ab_kern.sh "<bench args>" v1 v2 ...: per-kernel stats of library variants for one bench configuration
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
ARGS=$1; shift
for v in "$@"; do
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$v && KGE_LIB=$R/dgl-ke_amd/variants/libkge_$v.so timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -- python $R/bench.py --no-cpu-baseline --no-configs --hogwild 0 --steps 1200 --warmup 120 $ARGS > /tmp/prof_$v.log 2>&1
  echo "== $v"; python $R/tools/rocpd_stats.py $(ls /tmp/prof_$v/*/*_results.db | head -1) | head -7 | cut -c1-64,73-110
done
