#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03n}
B="--no-cpu-baseline --hogwild 0 --no-async-update"
run() { n=$1; shift
  timeout 200 python bench.py $B "$@" > $O/${TAG}_$n.json 2> $O/${TAG}_$n.err
  python -c "import json;d=json.load(open('$O/${TAG}_$n.json'));print('%-34s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline']['event_ms_per_step']))" || tail -3 $O/${TAG}_$n.err
}
run base_long
run base_drv --steps 20 --warmup 5
HIP_FORCE_DEV_KERNARG=1 run devkernarg_long
HIP_FORCE_DEV_KERNARG=1 run devkernarg_drv --steps 20 --warmup 5
HIP_FORCE_DEV_KERNARG=0 run nodevkernarg_long
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 run pktcap1_long
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run pktcap0_long
GPU_MAX_HW_QUEUES=1 run hwq1_long
GPU_MAX_HW_QUEUES=1 run hwq1_drv --steps 20 --warmup 5
HSA_ENABLE_INTERRUPT=0 run noint_long
HSA_ENABLE_INTERRUPT=0 run noint_drv --steps 20 --warmup 5
AMD_SERIALIZE_KERNEL=0 HIP_LAUNCH_BLOCKING=0 run base2_long
