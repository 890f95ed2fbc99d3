#!/bin/bash
# per-kernel stats of one filtered evaluation (tools/eval_timing.py): usage tools/eval_kstats.sh [model] [batch] ["" | variants/libkge_x.so ...]
R=$GRAFT_REPO_ROOT; M=${1:-TransE_l2}; B=${2:-1024}; shift; shift
cd /tmp; export TMPDIR=/tmp
for L in "${@:-}"; do
  rm -rf /tmp/prof_e
  KGE_LIB=${L:+$R/dgl-ke_amd/$L} timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_e -- python $R/tools/eval_timing.py $M $B > /tmp/prof_e.log 2>&1
  echo "== ${L:-main}  $(grep 'evaluate(cache) call 2' /tmp/prof_e.log)"
  python $R/tools/rocpd_stats.py $(ls /tmp/prof_e/*/*_results.db | head -1) 2>&1 | head -5 | cut -c1-64,73-130
done
