#!/bin/bash
# tools/ab.sh v1 v2 ... : interleaved A/B of library variants (2 rounds), prints ms/step and per-kernel avg
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
for round in 1 2; do for v in "$@"; do
  KGE_LIB=$R/dgl-ke_amd/variants/libkge_$v.so timeout 200 python $R/bench.py --no-cpu-baseline --steps 2400 --warmup 240 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v round $round', d['ms_per_step'])"
done; done
for v in "$@"; do
  cd /tmp && export TMPDIR=/tmp && KGE_LIB=$R/dgl-ke_amd/variants/libkge_$v.so timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -- python $R/bench.py --no-cpu-baseline --steps 1200 --warmup 120 > /tmp/prof_$v.log 2>&1
  echo "== $v"; python $R/tools/rocpd_stats.py $(ls /tmp/prof_$v/*/*_results.db | head -1) | head -7 | cut -c1-60,73-120
done
