#!/bin/bash
# round 6: kernel stats + per-phase diagnostics of the FB15k-shaped graph through the N > 1 path on one GPU (cfg-T's step, RCCL
# collectives kept at world 1, relation partitioning, group graphs): synchronous schedule vs every exchange on the side stream
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
E="KGE_FORCE_DIST=1 KGE_DIST_MODE=a2a KGE_DIST_FORCE_COLL=1 KGE_DIST_OTHER_LEG=0 KGE_DIST_REL_PART=force WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29613"
env $E KGE_DIST_PIPELINE=0 timeout 300 python $R/bench.py --gpus 1 --workload transe_l2_fb15k --steps 600 --warmup 120 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'phase_us_per_step (eager, one HIP event per phase):', d['config']['diagnostics'][0]['phase_us_per_step'])"
for P in 0 overlap; do
  rm -rf /tmp/prof_p
  env $E KGE_DIST_DIAG=0 KGE_DIST_PIPELINE=$P timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -- python $R/bench.py --gpus 1 --workload transe_l2_fb15k --steps 600 --warmup 120 --no-cpu-baseline > /tmp/prof_p.log 2>&1
  echo "== pipeline=$P  $(grep '^{' /tmp/prof_p.log | tail -1 | python -c "import sys,json; print('ms_per_step', json.loads(sys.stdin.read())['ms_per_step'])")"
  python $R/tools/rocpd_stats.py $(ls /tmp/prof_p/*/*_results.db | head -1) 2>&1 | head -14 | cut -c1-72,73-130
done
