#!/bin/bash
# tools/final_profiles.sh TAG: the profile set committed under profiles/ for one build (run on the GPU box through gpurun):
# kernel stats (rocprofv3 --kernel-trace --stats), FETCH_SIZE / WRITE_SIZE PMC passes, wavefront timeline, bench line
TAG=$1; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O
W=transe_l2_fb15k
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_f
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -- python $R/bench.py --no-cpu-baseline --no-configs --hogwild 0 --no-async-update --steps 1200 --warmup 120 > $O/${TAG}_prof.log 2>&1
cd $R && python tools/rocpd_stats.py $(ls /tmp/prof_f/*/*_results.db | head -1) > $O/${TAG}_kernel_stats_$W.txt 2>&1; head -12 $O/${TAG}_kernel_stats_$W.txt
bash $R/tools/pmc_cycle.sh $TAG --hogwild 0 --no-async-update > /dev/null 2>&1
cp $O/${TAG}_pmc_FETCH_SIZE.txt $O/${TAG}_pmc_fetch_size_$W.txt; cp $O/${TAG}_pmc_WRITE_SIZE.txt $O/${TAG}_pmc_write_size_$W.txt
rm -f $O/${TAG}_pmc_*.db
# (the timeline build: tools/build_variant.sh tl -DKGE_TIMELINE - hipcc, no GPU needed; built here when it is missing)
[ -f $R/dgl-ke_amd/variants/libkge_tl.so ] || bash $R/tools/build_variant.sh tl -DKGE_TIMELINE > /dev/null 2>&1
cd $R && KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 100 python tools/timeline.py > $O/${TAG}_timeline.txt 2>&1; tail -8 $O/${TAG}_timeline.txt
KGE_LIB=$R/dgl-ke_amd/variants/libkge_tl.so timeout 100 python tools/timeline.py --skew > $O/${TAG}_timeline_skew.txt 2>&1
