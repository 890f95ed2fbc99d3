#!/bin/bash
# one GPU round-trip: parity tests, bench, rocprof kernel stats.  usage: tools/gpu_cycle.sh TAG [bench args]
TAG=$1; shift
[ -z "$GRAFT_REPO_ROOT" ] && export GRAFT_REPO_ROOT=$(pwd)
mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -q --timeout=120 -x 2>&1 | tail -15 > gpurun_out/${TAG}_pytest.log; tail -3 gpurun_out/${TAG}_pytest.log
timeout 300 python bench.py "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err | grep -v amdgpu.ids
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -- python $R/bench.py --no-cpu-baseline --no-configs --steps 1200 --warmup 120 "$@" > $R/gpurun_out/${TAG}_prof.log 2>&1
cd $R && python tools/rocpd_stats.py $(ls gpurun_out/${TAG}_prof/*/*_results.db | head -1) > gpurun_out/${TAG}_kernel_stats.txt 2>&1; head -12 gpurun_out/${TAG}_kernel_stats.txt
rm -rf gpurun_out/${TAG}_prof
