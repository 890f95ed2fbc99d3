#!/bin/bash
# full GPU check: pytest -m gpu, smoke, the driver-shape bench line.  usage: gpu_check.sh [outdir-name]
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/${1:-check}; mkdir -p $O
( timeout 1700 python -m pytest tests -m gpu -q --timeout=300 2>&1 | grep -v amdgpu.ids | tail -8 ) > $O/pytest_all.log; tail -4 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['config']['launch'][:100]); print({k:v.get('us_per_step') for k,v in d['configs'].items()})"
