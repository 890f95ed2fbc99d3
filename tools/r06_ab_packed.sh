#!/bin/bash
# round 6: packed single-trace gradient messages (default) against the two-trace messages (KGE_DIST_PACKED=0) on the world-1 proxy of
# the N > 1 path (cfg-R, RCCL exchanges kept, group graphs): synchronous and overlapped schedules, with and without relation partitioning
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; cd $R
run() { # name, env...
  n=$1; shift
  env "$@" KGE_DIST_MODE=a2a KGE_DIST_FORCE_COLL=1 KGE_DIST_OTHER_LEG=0 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + RANDOM % 300)) \
    timeout 200 python bench.py --gpus 1 --workload rotate_freebase --steps 600 --warmup 120 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('$n', 'us/step', round(1e3*d['ms_per_step'],2), 'msg floats', c.get('message_floats'), 'bucket', c.get('bucket_rows'), '+', c.get('message_extra_rows'), 'growth', c.get('message_extra_growth'))
dg=c.get('diagnostics')
if dg: print('   phases', dg[0].get('phase_us_per_step'))"
}
for P in 1 0; do
  run "packed=$P sync          " KGE_DIST_PACKED=$P KGE_DIST_PIPELINE=0
  run "packed=$P sync relpart  " KGE_DIST_PACKED=$P KGE_DIST_PIPELINE=0 KGE_DIST_REL_PART=force
  run "packed=$P overlap relpart" KGE_DIST_PACKED=$P KGE_DIST_PIPELINE=overlap KGE_DIST_REL_PART=force
done
