#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03h}
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -m gpu -q --timeout=300 -x -k "dist or route or gather_rows_req or apply_merged or world" 2>&1 | grep -v "amdgpu.ids" | tail -40 > $O/${TAG}_pytest.log; tail -30 $O/${TAG}_pytest.log
for m in a2a p2p; do for wl in transe_l2_freebase rotate_freebase; do
KGE_DIST_MODE=$m timeout 300 python bench.py --gpus 1 --workload $wl --steps 1200 --warmup 120 > $O/${TAG}_${m}_$wl.json 2> $O/${TAG}_${m}_$wl.err
python -c "import json;d=json.load(open('$O/${TAG}_${m}_$wl.json'));print('$m $wl', 1e3*d['ms_per_step'], 'us/step', d['value'], d['config'].get('bucket_overflows'))" || tail -5 $O/${TAG}_${m}_$wl.err
done; done
