#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update --steps 20 --warmup 5"
p() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('%-44s wall %.3f events %.3f' % ('$1', 1e3*d['ms_per_step'],1e3*d['roofline']['event_ms_per_step']))"; }
run() { n="$1"; shift; timeout 100 env "$@" python bench.py $B 2>/dev/null | grep "^{" | p "$n"; }
run base A=1
run HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=1
run GRAPH_PACKET_CAPTURE=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run KERNARG_HDP_FLUSH_WA=0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run KERNARG_HDP_FLUSH_WA=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
run ROC_CPU_WAIT_FOR_SIGNAL=0 ROC_CPU_WAIT_FOR_SIGNAL=0
run AMD_DIRECT_DISPATCH=0 AMD_DIRECT_DISPATCH=0
run ROC_USE_FGS_KERNARG=0 ROC_USE_FGS_KERNARG=0
run KERNARG_COPY_OPT=0 DEBUG_HIP_KERNARG_COPY_OPT=0
run SKIP_KERNEL_ARG_COPY=1 ROC_SKIP_KERNEL_ARG_COPY=1
run AQL_QUEUE_SIZE=65536 ROC_AQL_QUEUE_SIZE=65536
run ACTIVE_WAIT=100000 ROC_ACTIVE_WAIT_TIMEOUT=100000
