#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=r03_v1
bash tools/final_profiles.sh $TAG 2>&1 | grep -v amdgpu.ids | tail -30
python tools/kernel_stats_json.py $O/${TAG}_kernel_stats_transe_l2_fb15k.txt $TAG transe_l2_fb15k > /dev/null
python tools/traffic_json.py $O/${TAG}_pmc_fetch_size_transe_l2_fb15k.txt $O/${TAG}_pmc_write_size_transe_l2_fb15k.txt $TAG transe_l2_fb15k 120
cp profiles/latest_kernel_stats.json $O/${TAG}_latest_kernel_stats.json; cp profiles/latest_traffic.json $O/${TAG}_latest_traffic.json
timeout 300 python bench.py > $O/${TAG}_bench_transe_l2_fb15k.json 2> $O/${TAG}_bench.err; cut -c1-600 $O/${TAG}_bench_transe_l2_fb15k.json
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_like.json 2>> $O/${TAG}_bench.err; cut -c1-300 $O/${TAG}_bench_driver_like.json
