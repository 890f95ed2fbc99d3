#!/bin/bash
# tools/ab_schedules.sh [steps] : the schedules of the sharded step on ONE GPU - cfg-R's per-GPU step with its RCCL collectives kept at
# world 1 (KGE_DIST_FORCE_COLL=1), groups replayed from hipGraphs: synchronous / pull pipeline / every exchange on the side stream
# (DistEngine._steps_overlapped), with all-gathered relations and with relation partitioning (the N > 1 default of bench.py --gpus N).
# How profiles/r05_overlap_schedule.txt was measured.  Output: gpurun_out/ab_schedules.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
K=${1:-600}
run() {  # name, env...
  n=$1; shift
  env KGE_DIST_MODE=a2a KGE_DIST_FORCE_COLL=1 KGE_DIST_OTHER_LEG=0 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29650 + RANDOM % 200)) "$@" \
    timeout 150 python $R/bench.py --gpus 1 --steps $K --warmup 120 --no-cpu-baseline --hogwild 0 --no-async-update --no-configs --workload rotate_freebase 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('%-46s %8.2f us/step  launch=%s  mean_loss=%.6f' % ('$n', d['ms_per_step']*1000, d['config']['launch'], d.get('mean_loss',0)))"
}
for r in 1 2; do
run "synchronous, all-gathered relations" KGE_DIST_PIPELINE=0
run "pull pipeline, all-gathered relations" KGE_DIST_PIPELINE=1
run "overlapped, all-gathered relations" KGE_DIST_PIPELINE=overlap
run "synchronous, relation partitioning" KGE_DIST_PIPELINE=0 KGE_DIST_REL_PART=force
run "overlapped, relation partitioning" KGE_DIST_PIPELINE=overlap KGE_DIST_REL_PART=force
run "overlapped, relation partitioning, equal prio" KGE_DIST_PIPELINE=overlap KGE_DIST_REL_PART=force KGE_DIST_SIDE_PRIORITY=0
run "overlapped, relation partitioning, eager" KGE_DIST_PIPELINE=overlap KGE_DIST_REL_PART=force KGE_DIST_GRAPH=0
done | tee $O/ab_schedules.txt
