#!/bin/bash
# like ab_kern.sh but prints min/avg per kernel INCLUDING repeated launches separately is not possible in stats;
# prints calls and avg so that a doubled launch count shows as the mean of both passes
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
ARGS=$1; shift
for v in "$@"; do
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$v && KGE_LIB=$R/dgl-ke_amd/variants/libkge_$v.so timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -- python $R/bench.py --no-cpu-baseline --hogwild 0 --steps 1200 --warmup 120 $ARGS > /tmp/prof_$v.log 2>&1
  echo "== $v"; python $R/tools/rocpd_stats.py $(ls /tmp/prof_$v/*/*_results.db | head -1) | head -7
done
