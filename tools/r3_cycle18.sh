#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03r}
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v amdgpu.ids | tail -3
for pl in 1 0; do
KGE_DIST_FORCE_COLL=1 KGE_DIST_PIPELINE=$pl timeout 300 python bench.py --gpus 1 --workload rotate_freebase --steps 400 --warmup 40 > $O/${TAG}_coll_$pl.json 2> $O/${TAG}_coll_$pl.err
tail -1 $O/${TAG}_coll_$pl.json | python -c "import json,sys;d=json.loads(sys.stdin.read());print('world-1 RCCL path pipeline=$pl', 1e3*d['ms_per_step'], 'us/step', d['value'], d['config'].get('bucket_overflows'), d['mean_loss'])" || tail -8 $O/${TAG}_coll_$pl.err
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
