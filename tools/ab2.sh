#!/bin/bash
R=$GRAFT_REPO_ROOT
run() { # name lib args
  cd /tmp && export TMPDIR=/tmp && KGE_LIB=$2 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_$1 -- python $R/bench.py --no-cpu-baseline --steps 1200 --warmup 120 $3 > /tmp/p_$1.log 2>&1
  echo "== $1 [$3]"; python $R/tools/rocpd_stats.py $(ls /tmp/p_$1/*/*_results.db | head -1) | head -6 | cut -c1-64,73-110
}
run base $R/dgl-ke_amd/variants/libkge_base.so ""
run nosq $R/dgl-ke_amd/variants/libkge_nosq.so ""
run noadv $R/dgl-ke_amd/variants/libkge_base.so "--no-adv"
run nosq_noadv $R/dgl-ke_amd/variants/libkge_nosq.so "--no-adv"
