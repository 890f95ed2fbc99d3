#!/bin/bash
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05c15; mkdir -p $O
for rep in 1 2 3; do for m in serial fused; do
  timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-async-update --hogwild 0 --sampler-mode $m > $O/drv_${m}_$rep.json 2> $O/err_$m.txt
  python -c "
import json;d=json.loads(open('$O/drv_${m}_$rep.json').read().strip().splitlines()[-1]);print('driver-shape $m', d['ms_per_step'], d['roofline']['event_ms_per_step'], d['value'])" || tail -5 $O/err_$m.txt
done; done
for fm in 0 48 1000; do
  KGE_FUSED_MAX=$fm timeout 120 python bench.py --steps 1200 --warmup 120 --no-cpu-baseline --no-configs --no-async-update --hogwild 0 --sampler-mode fused > $O/long_fm$fm.json 2> $O/err.txt
  python -c "
import json;d=json.loads(open('$O/long_fm$fm.json').read().strip().splitlines()[-1]);print('long fused_max=$fm', d['ms_per_step'], d['roofline']['event_ms_per_step'], d['value'])" || tail -5 $O/err.txt
done
for g in 20 40; do for fm in 0 1000; do
  KGE_FUSED_MAX=$fm timeout 120 python bench.py --steps 1200 --warmup 120 --graph-steps $g --no-cpu-baseline --no-configs --no-async-update --hogwild 0 --sampler-mode fused > $O/long_g${g}_fm$fm.json 2> $O/err.txt
  python -c "
import json;d=json.loads(open('$O/long_g${g}_fm$fm.json').read().strip().splitlines()[-1]);print('long G=$g fused_max=$fm', d['ms_per_step'], d['value'])" || tail -5 $O/err.txt
done; done
