#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for NCH in ${NCHS:-default 1 2 4 8}; do
  for W in rotate_freebase transe_l2_fb15k; do
    X=""; [ "$NCH" != "default" ] && X="NCCL_MAX_NCHANNELS=$NCH NCCL_MIN_NCHANNELS=$NCH"
    F=""; [ "$W" = "transe_l2_fb15k" ] && F="KGE_FORCE_DIST=1"
    r=$(env $X $F KGE_DIST_MODE=a2a KGE_DIST_FORCE_COLL=1 KGE_DIST_OTHER_LEG=0 KGE_DIST_DIAG=0 KGE_DIST_REL_PART=force KGE_DIST_PIPELINE=overlap WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29617 \
      timeout 200 python bench.py --gpus 1 --workload $W --steps 600 --warmup 120 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "nchannels=$NCH $W overlapped ms_per_step $r"
  done
done
