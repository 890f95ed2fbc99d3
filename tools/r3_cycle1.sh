#!/bin/bash
# round-3 GPU cycle 1: parity suite, driver-like + long bench (merged vs split forward), kernel stats, timeline
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03a}
timeout 900 python -m pytest tests -m gpu -q --timeout=180 -x 2>&1 | tail -15 > $O/${TAG}_pytest.log; tail -4 $O/${TAG}_pytest.log
B="--no-cpu-baseline --hogwild 0 --no-async-update"
for rep in 1 2; do
timeout 200 python bench.py --steps 20 --warmup 5 $B > $O/${TAG}_driverlike_$rep.json 2> $O/${TAG}_driverlike.err; python -c "import json;d=json.load(open('$O/${TAG}_driverlike_$rep.json'));print('driver-like merged',d['ms_per_step'],d['roofline']['event_ms_per_step'])"
done
timeout 200 python bench.py --steps 20 --warmup 5 $B --flags 128 > $O/${TAG}_driverlike_split.json 2>> $O/${TAG}_driverlike.err; python -c "import json;d=json.load(open('$O/${TAG}_driverlike_split.json'));print('driver-like split',d['ms_per_step'],d['roofline']['event_ms_per_step'])"
timeout 200 python bench.py $B > $O/${TAG}_long.json 2> $O/${TAG}_long.err; python -c "import json;d=json.load(open('$O/${TAG}_long.json'));print('long merged',d['ms_per_step'],d['roofline']['event_ms_per_step'])"
timeout 200 python bench.py $B --flags 128 > $O/${TAG}_long_split.json 2>> $O/${TAG}_long.err; python -c "import json;d=json.load(open('$O/${TAG}_long_split.json'));print('long split',d['ms_per_step'],d['roofline']['event_ms_per_step'])"
timeout 200 python bench.py $B --workload distmult_fb15k > $O/${TAG}_long_distmult.json 2>> $O/${TAG}_long.err; python -c "import json;d=json.load(open('$O/${TAG}_long_distmult.json'));print('distmult merged',d['ms_per_step'])"
timeout 200 python bench.py $B --workload distmult_fb15k --flags 128 > $O/${TAG}_long_distmult_split.json 2>> $O/${TAG}_long.err; python -c "import json;d=json.load(open('$O/${TAG}_long_distmult_split.json'));print('distmult split',d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_f
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -- python $R/bench.py $B --steps 1200 --warmup 120 > $O/${TAG}_prof.log 2>&1
cd $R && python tools/rocpd_stats.py $(ls /tmp/prof_f/*/*_results.db | head -1) > $O/${TAG}_kernel_stats.txt 2>&1; head -9 $O/${TAG}_kernel_stats.txt | cut -c1-70,73-120
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 100 python tools/timeline.py > $O/${TAG}_timeline.txt 2>&1; tail -8 $O/${TAG}_timeline.txt
