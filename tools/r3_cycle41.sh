#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
B="--no-cpu-baseline --hogwild 0 --no-async-update"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_w && timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_w -- python $R/bench.py $B --steps 20 --warmup 5 > /tmp/prof_w.log 2>&1
python $R/tools/first_steps.py $(ls /tmp/prof_w/*/*_results.db | head -1) 20 | tee $O/c41_first_steps.txt
