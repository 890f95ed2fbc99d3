#!/bin/bash
# VALU issue-rate probe + host wait knobs on the 20-step driver shape
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 120 tools/valu_rate_probe.bin > $O/valu_rate_probe.txt 2>&1; cat $O/valu_rate_probe.txt
B="--no-cpu-baseline --hogwild 0 --no-async-update --steps 20 --warmup 5"
run() { n=$1; shift
  timeout 200 env "$@" python bench.py $B > $O/c22_$n.json 2> $O/c22_$n.err
  python -c "import json;d=json.load(open('$O/c22_$n.json'));print('%-28s wall %.3f us  events %.3f us' % ('$n', 1e3*d['ms_per_step'], 1e3*d['roofline']['event_ms_per_step']))" || tail -3 $O/c22_$n.err
}
run base A=1
run base2 A=1
run nointr HSA_ENABLE_INTERRUPT=0
run nointr2 HSA_ENABLE_INTERRUPT=0
run activewait ROC_ACTIVE_WAIT_TIMEOUT=1000
run spin HIP_FORCE_SPIN=1 GPU_FORCE_BLIT_COPY_SIZE=0
