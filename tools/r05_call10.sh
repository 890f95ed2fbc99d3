#!/bin/bash
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05c10; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_sampler.py -m gpu -q --timeout=300 -x 2>&1 | grep -v amdgpu.ids | tail -25 ) > $O/pytest_sampler.log
tail -25 $O/pytest_sampler.log
for rep in 1 2; do
for m in serial fused; do
  timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-async-update --hogwild 0 --sampler-mode $m > $O/drv_${m}_$rep.json 2> $O/err_$m.txt
  python -c "
import json;d=json.loads(open('$O/drv_${m}_$rep.json').read().strip().splitlines()[-1]);print('driver-shape $m', d['ms_per_step'], d['roofline']['event_ms_per_step'], d['value'], d['mean_loss'])" || tail -5 $O/err_$m.txt
done; done
for m in serial fused; do
  timeout 120 python bench.py --steps 1200 --warmup 120 --no-cpu-baseline --no-configs --no-async-update --hogwild 0 --sampler-mode $m > $O/long_$m.json 2> $O/err_$m.txt
  python -c "
import json;d=json.loads(open('$O/long_$m.json').read().strip().splitlines()[-1]);print('long $m', d['ms_per_step'], d['roofline']['event_ms_per_step'], d['value'], d['mean_loss'])" || tail -5 $O/err_$m.txt
done
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; O=gpurun_out/r05c10; mkdir -p $O
bash tools/build_variant.sh tl -DKGE_TIMELINE > $O/build.log 2>&1; tail -1 $O/build.log
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 200 python tools/timeline.py --workload transe_l2_fb15k --graph-steps 10 --tail > $O/timeline_tail.txt 2>&1
grep -v amdgpu.ids $O/timeline_tail.txt
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 200 python tools/timeline.py --workload transe_l2_fb15k --graph-steps 10 > $O/timeline_notail.txt 2>&1
grep -v amdgpu.ids $O/timeline_notail.txt
