#!/bin/bash
# round 6: XCD-aware block order of the pairwise kernels (kge_neg_bcast.hip sb_block) vs hardware order (-DSB_NO_XCD variant):
# per-kernel times on three workloads and the FETCH_SIZE pass at cfg-R's shape
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_xcd; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -q -x --timeout=300 -k "world1_local_shortcut or group_graph" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=300 -k "RotatE or TransE_l1 or rotate or transe_l1" 2>&1 | tail -3
for W in rotate_wide rotate_fb15k transe_l1_fb15k; do
  bash tools/ab_kstats.sh $W "neg_|edge_|update|loss|==" "" variants/libkge_noxcd.so 2>&1 | tee $O/kstats_$W.txt
done
cd /tmp; export TMPDIR=/tmp
for L in "" variants/libkge_noxcd.so; do
  rm -rf /tmp/pmc_x
  KGE_LIB=${L:+$R/dgl-ke_amd/$L} timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_x -- python $R/bench.py --workload rotate_wide --no-cpu-baseline --no-configs --no-async-update --hogwild 0 --sampler-mode serial --steps 240 --warmup 120 --graph-steps 120 > /tmp/pmc_x.log 2>&1
  echo "== FETCH ${L:-main}"; python $R/tools/rocpd_stats.py $(ls /tmp/pmc_x/*/*_results.db | head -1) --pmc 2>&1 | egrep "neg_|edge_|update|loss" | tee $O/fetch_${L:+noxcd}.txt
done
cd $R
for M in a2a p2p; do KGE_DIST_MODE=$M KGE_DIST_OTHER_LEG=0 timeout 200 python bench.py --workload rotate_freebase --steps 600 --warmup 120 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$M', d['ms_per_step'], d['config'].get('mode'), d['config'].get('launch'))"; done
