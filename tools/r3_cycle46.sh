#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_w && timeout 300 rocprofv3 --sys-trace -d /tmp/prof_w -- python $R/bench.py $B --steps 20 --warmup 5 > /tmp/prof_w.log 2>&1
ls -la /tmp/prof_w/*/ | head
python $R/tools/stall_context.py $(ls /tmp/prof_w/*/*_results.db | head -1) 2>&1 | cut -c1-200 | tee $O/c46_stall_context.txt | tail -90
