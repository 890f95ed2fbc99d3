#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R; export GRAFT_REPO_ROOT=$R
bash tools/pmc_cycle.sh r03_v3_distmult --hogwild 0 --no-async-update --workload distmult_fb15k 2>&1 | tail -30
rm -f $O/r03_v3_distmult_pmc_*.db
