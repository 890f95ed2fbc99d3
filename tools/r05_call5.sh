#!/bin/bash
# cfg-R per-GPU shape on one local table: per-wavefront timeline + rocprof kernel stats
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; O=gpurun_out/r05c5; mkdir -p $O
bash tools/build_variant.sh tl -DKGE_TIMELINE > $O/build.log 2>&1; tail -1 $O/build.log
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 200 python tools/timeline.py --workload rotate_wide --graph-steps 10 > $O/timeline_rotate_wide.txt 2>&1
cat $O/timeline_rotate_wide.txt | grep -v amdgpu.ids
timeout 200 python bench.py --workload rotate_wide --no-cpu-baseline --no-configs --no-async-update --steps 600 --warmup 40 > $O/bench_rotate_wide.json 2> $O/bench_rotate_wide.err
python -c "
import json;d=json.loads(open('$O/bench_rotate_wide.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python $R/bench.py --workload rotate_wide --no-cpu-baseline --no-configs --no-async-update --steps 600 --warmup 40 > $R/$O/prof.log 2>&1
cd $R && python tools/rocpd_stats.py $(ls $O/prof/*/*_results.db | head -1) > $O/kernel_stats_rotate_wide.txt 2>&1; head -12 $O/kernel_stats_rotate_wide.txt
rm -rf $O/prof
