#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03i}
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -m gpu -q --timeout=300 -k "dist or route or gather_rows_req or apply_merged or world" 2>&1 | grep -v "amdgpu.ids" | tail -40 > $O/${TAG}_pytest.log; tail -30 $O/${TAG}_pytest.log
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_d
KGE_DIST_MODE=a2a timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -- python $R/bench.py --gpus 1 --workload transe_l2_freebase --steps 1200 --warmup 120 > $O/${TAG}_prof.log 2>&1
cd $R && python tools/rocpd_stats.py $(ls /tmp/prof_d/*/*_results.db | head -1) > $O/${TAG}_kernel_stats_a2a_w1.txt 2>&1; head -12 $O/${TAG}_kernel_stats_a2a_w1.txt | cut -c1-70,73-120
