#!/bin/bash
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05c14; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q --timeout=300 -x 2>&1 | grep -v amdgpu.ids | tail -12 ) > $O/pytest_all.log; tail -6 $O/pytest_all.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['launch'][:160]); print({k:(v.get('us_per_step') if isinstance(v,dict) else v) for k,v in d['configs'].items()}); print(d['cpu_baseline']['kind'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
