#!/bin/bash
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05c11; mkdir -p $O
for g in 20 40 120; do for m in serial fused; do
  timeout 120 python bench.py --steps 1200 --warmup 120 --graph-steps $g --no-cpu-baseline --no-configs --no-async-update --hogwild 0 --sampler-mode $m > $O/long_${m}_$g.json 2> $O/err_$m.txt
  python -c "
import json;d=json.loads(open('$O/long_${m}_$g.json').read().strip().splitlines()[-1]);print('G=$g $m', d['ms_per_step'], d['roofline']['event_ms_per_step'], d['value'])" || tail -5 $O/err_$m.txt
done; done
