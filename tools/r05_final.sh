#!/bin/bash
# round 5: the profile set of the final build (kernel stats, driver-shape trace, timeline, default bench line)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05final; mkdir -p $O; cd $R
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err
timeout 400 python bench.py --no-configs > $O/bench_long.json 2> $O/bench_long.err
for f in bench_driver_shape bench_long; do python -c "
import json;d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['kind'], d['cpu_baseline']['value'])"; done
cd /tmp && export TMPDIR=/tmp
for W in transe_l2_fb15k rotate_wide; do
rm -rf /tmp/prof_f; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -- python $R/bench.py --workload $W --no-cpu-baseline --no-configs --hogwild 0 --no-async-update --steps 1200 --warmup 120 > $O/prof_$W.log 2>&1
python $R/tools/rocpd_stats.py $(ls /tmp/prof_f/*/*_results.db | head -1) > $O/kernel_stats_$W.txt 2>&1; head -8 $O/kernel_stats_$W.txt | cut -c1-140
done
rm -rf /tmp/prof_d; timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_d -- python $R/bench.py --no-cpu-baseline --no-configs --hogwild 0 --no-async-update --steps 20 --warmup 5 > $O/prof_driver.log 2>&1
python $R/tools/first_steps.py $(ls /tmp/prof_d/*/*_results.db | head -1) 20 > $O/driver_shape_trace.txt 2>&1; tail -12 $O/driver_shape_trace.txt
cd $R; [ -f dgl-ke_amd/variants/libkge_tl.so ] || bash tools/build_variant.sh tl -DKGE_TIMELINE > /dev/null 2>&1
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 100 python tools/timeline.py > $O/timeline.txt 2>&1
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 100 python tools/timeline.py --tail > $O/timeline_tail.txt 2>&1
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 100 python tools/timeline.py --workload rotate_wide --graph-steps 10 > $O/timeline_rotate_wide.txt 2>&1
KGE_DIST_FORCE_COLL=1 KGE_DIST_MODE=a2a KGE_DIST_OTHER_LEG=0 KGE_DIST_PIPELINE=0 KGE_DIST_EAGER_LEG=1 timeout 200 python bench.py --workload rotate_freebase --steps 600 --warmup 40 --no-cpu-baseline > $O/proxy_forced_exchange.json 2> $O/proxy.err
python -c "
import json;d=json.loads(open('$O/proxy_forced_exchange.json').read().strip().splitlines()[-1]);print('proxy', d['ms_per_step'], d.get('a2a_eager'), d['config'].get('launch'))"
timeout 300 python tools/ab_driver_shape.py 40 2>&1 | grep -v amdgpu.ids | tail -4 > $O/ab_driver_shape.txt; cat $O/ab_driver_shape.txt
