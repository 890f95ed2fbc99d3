#!/bin/bash
# round 6: 8 wavefronts per workgroup in the shared-pair backward at cfg-R's width (lc_wpb) vs 4 (-DLC_NO_WPB8): parity, per-kernel times,
# FETCH_SIZE / WRITE_SIZE at cfg-R's shape
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_wpb8; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=300 -k "wide_rows or real_table or config_shapes or balanced_split or shared_pair" 2>&1 | tail -3
for rep in 1 2; do bash tools/ab_kstats.sh rotate_wide "neg_|edge_|update|loss|==" "" variants/libkge_nowpb8.so 2>&1 | tee -a $O/kstats_rotate_wide.txt; done
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do for L in "" variants/libkge_nowpb8.so; do
  rm -rf /tmp/pmc_x
  KGE_LIB=${L:+$R/dgl-ke_amd/$L} timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_x -- python $R/bench.py --workload rotate_wide --no-cpu-baseline --no-configs --no-async-update --hogwild 0 --sampler-mode serial --steps 240 --warmup 120 --graph-steps 120 > /tmp/pmc_x.log 2>&1
  echo "== $C ${L:-main}"; python $R/tools/rocpd_stats.py $(ls /tmp/pmc_x/*/*_results.db | head -1) --pmc 2>&1 | egrep "neg_|edge_|update|loss" | tee $O/${C}_${L:+nowpb8}.txt
done; done
