#!/bin/bash
# tools/ab_flags.sh "<bench args A>" "<bench args B>" ... : per-kernel stats for several bench configurations
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
i=0
for cfg in "$@"; do
  i=$((i+1))
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_f$i && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_f$i -- python $R/bench.py --no-cpu-baseline --no-configs --steps 1200 --warmup 120 $cfg > /tmp/prof_f$i.log 2>&1
  echo "== [$cfg]"; python $R/tools/rocpd_stats.py $(ls /tmp/prof_f$i/*/*_results.db | head -1) | head -8 | cut -c1-64,73-110
done
