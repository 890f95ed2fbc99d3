#!/bin/bash
# A/B of library variants on one workload: tools/ab_rescal.sh WORKLOAD "" variants/libkge_x.so ...   ("" = the in-tree library)
cd $GRAFT_REPO_ROOT
W=${1:-rescal_fb15k}; shift
K=${AB_TESTS:-rescal or RESCAL}
timeout 300 python -m pytest tests -m gpu -q -x -k "$K" --timeout=300 2>&1 | grep -v amdgpu.ids | tail -2
for L in "$@"; do
  for i in 1 2; do
  KGE_LIB=${L:+$GRAFT_REPO_ROOT/dgl-ke_amd/$L} timeout 200 python bench.py --workload $W --no-cpu-baseline --no-configs --hogwild 0 --no-async-update --steps 300 --warmup 30 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=${L:-main}', d['ms_per_step'])"
  done
done
