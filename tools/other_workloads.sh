#!/bin/bash
# tools/other_workloads.sh TAG: edges/s + per-kernel stats of every bench workload, MFMA counters of the GEMM models
TAG=$1; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out/${TAG}_workloads.txt
mkdir -p $R/gpurun_out; : > $O
for W in transe_l2_fb15k distmult_fb15k complex_wikikg2 rotate_fb15k transe_l1_fb15k simple_fb15k rescal_fb15k transr_fb15k; do
  echo "== $W" >> $O
  timeout 200 python $R/bench.py --no-cpu-baseline --no-configs --hogwild 0 --steps 600 --warmup 120 --workload $W 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        au, ar = d.get('async_update', {}), d.get('async_update_rel', {})
        print('edges/s %.1f  ms/step %.5f  hbm frac %.4f  mfma %s  | --async_update us/step: entity-only %s, relation deferred %s' % (
            d['value'], d['ms_per_step'], r['frac'], r.get('mfma', {}).get('achieved'), au.get('us_per_step'), ar.get('us_per_step')))
" >> $O
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_w && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_w -- python $R/bench.py --no-cpu-baseline --no-configs --hogwild 0 --no-async-update --steps 600 --warmup 120 --workload $W > /tmp/prof_w.log 2>&1
  python $R/tools/rocpd_stats.py $(ls /tmp/prof_w/*/*_results.db | head -1) | head -9 | cut -c1-64,73-118 >> $O
  cd $R
done
for W in transe_l2_fb15k distmult_fb15k complex_wikikg2; do
  echo "== MFMA counters $W (per launch means)" >> $O
  bash $R/tools/pmc_sq.sh "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" --workload $W 2>&1 | grep -i "gemm" >> $O
done
cat $O
