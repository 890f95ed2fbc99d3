#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of library variants on one workload:
#   tools/ab_kstats.sh WORKLOAD PATTERN "" variants/libkge_x.so ...      ("" = the in-tree library; PATTERN = egrep on kernel names)
R=$GRAFT_REPO_ROOT; W=$1; P=$2; shift; shift
cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  rm -rf /tmp/prof_v
  KGE_LIB=${L:+$R/dgl-ke_amd/$L} timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_v -- python $R/bench.py --workload $W --no-cpu-baseline --no-configs --hogwild 0 --no-async-update --steps 100 --warmup 10 > /tmp/prof_v.log 2>&1
  echo "== ${L:-main}  $(grep '^{' /tmp/prof_v.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")"
  python $R/tools/rocpd_stats.py $(ls /tmp/prof_v/*/*_results.db | head -1) 2>&1 | egrep "$P" | cut -c1-60,76-130
done
