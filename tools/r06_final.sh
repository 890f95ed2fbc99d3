#!/bin/bash
# round 6: the profile set of the final build - bench lines (driver shape, long run), kernel stats at cfg-T and cfg-R's shape,
# FETCH_SIZE / WRITE_SIZE passes of five workloads / legs, the forced-exchange proxy legs
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06final; mkdir -p $O; cd $R
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err
timeout 400 python bench.py --no-configs > $O/bench_long.json 2> $O/bench_long.err
for f in bench_driver_shape bench_long; do python -c "
import json;d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['kind'], d['cpu_baseline']['value'])"; done
cd /tmp && export TMPDIR=/tmp
for W in transe_l2_fb15k rotate_wide; do
rm -rf /tmp/prof_f; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -- python $R/bench.py --workload $W --no-cpu-baseline --no-configs --hogwild 0 --no-async-update --steps 1200 --warmup 120 > $O/prof_$W.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_f/*/*_results.db | head -1) > $O/kernel_stats_$W.txt 2>&1; head -8 $O/kernel_stats_$W.txt | cut -c1-140
done
cd $R
bash tools/pmc_cycle.sh r06_transe_l2_fb15k --workload transe_l2_fb15k | grep -c kernel
bash tools/pmc_cycle.sh r06_complex_wikikg2 --workload complex_wikikg2 | grep -c kernel
bash tools/pmc_cycle.sh r06_rotate_wide --workload rotate_wide | grep -c kernel
KGE_DIST_MODE=a2a KGE_DIST_OTHER_LEG=0 bash tools/pmc_cycle.sh r06_rotate_freebase_a2a --workload rotate_freebase | grep -c kernel
KGE_DIST_MODE=p2p KGE_DIST_OTHER_LEG=0 bash tools/pmc_cycle.sh r06_rotate_freebase_p2p --workload rotate_freebase | grep -c kernel
rm -f gpurun_out/r06_*_pmc_*.db
ls gpurun_out/r06_*_pmc_*.txt
