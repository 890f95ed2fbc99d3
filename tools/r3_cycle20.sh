#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03s}
for wl in rotate_freebase transe_l2_freebase; do
KGE_DIST_FORCE_COLL=1 timeout 400 python bench.py --gpus 1 --workload $wl --steps 400 --warmup 40 > $O/${TAG}_$wl.json 2> $O/${TAG}_$wl.err
tail -1 $O/${TAG}_$wl.json | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$wl', 1e3*d['ms_per_step'], 'us/step', d['value'], d.get('a2a_eager'), d['config'].get('bucket_overflows'), d['mean_loss'], d['config']['workload'][-230:])" || tail -8 $O/${TAG}_$wl.err
done
KGE_DIST_FORCE_COLL=1 timeout 400 python bench.py --gpus 1 --workload rotate_freebase --steps 20 --warmup 5 > $O/${TAG}_drv.json 2> $O/${TAG}_drv.err
tail -1 $O/${TAG}_drv.json | python -c "import json,sys;d=json.loads(sys.stdin.read());print('driver-shaped', 1e3*d['ms_per_step'], 'us/step', d['value'], d.get('a2a_eager'))" || tail -8 $O/${TAG}_drv.err
