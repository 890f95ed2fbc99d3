#!/bin/bash
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; O=gpurun_out/r05c6; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_p2p.py tests/test_gpu_async.py -m gpu -q --timeout=300 -x 2>&1 | tail -8 ) > $O/pytest.log
tail -4 $O/pytest.log
bash tools/build_variant.sh tl -DKGE_TIMELINE > $O/build.log 2>&1; tail -1 $O/build.log
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 200 python tools/timeline.py --workload rotate_wide --graph-steps 10 > $O/timeline_rotate_wide.txt 2>&1
cat $O/timeline_rotate_wide.txt | grep -v amdgpu.ids
for wl in rotate_wide rotate_fb15k; do
timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-configs --no-async-update --steps 600 --warmup 40 > $O/bench_$wl.json 2> $O/bench_$wl.err
python -c "
import json;d=json.loads(open('$O/bench_$wl.json').read().strip().splitlines()[-1]);print('$wl', d['ms_per_step'], d['roofline']['frac'])"
done
