#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update --steps 20 --warmup 5"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('%-44s wall %.3f events %.3f' % ('$1', 1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step']))"; }
run() { n="$1"; shift; timeout 100 env "$@" python bench.py $B 2>/dev/null | grep "^{" | p "$n"; }
nproc; python -c "import os;print('affinity', len(os.sched_getaffinity(0)))"
run base A=1
run GPU_MAX_HW_QUEUES=1 GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=2 GPU_MAX_HW_QUEUES=2
run taskset_2cores A=1
taskset -c 2,3 timeout 100 python bench.py $B 2>/dev/null | grep "^{" | p taskset_c2_3
run OMP1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1
run HSA_ENABLE_INTERRUPT=0 HSA_ENABLE_INTERRUPT=0
run NOINTR_MAXQ1 HSA_ENABLE_INTERRUPT=0 GPU_MAX_HW_QUEUES=1
