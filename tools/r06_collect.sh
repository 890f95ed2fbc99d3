#!/bin/bash
# after `gpurun -- bash tools/r06_final.sh`: copy the round-6 final set from gpurun_out/ into profiles/ and rebuild profiles/latest_*.json
# (their source_hash = the tree they are collected in: run it on the tree the profiles were taken from)
cd "$(dirname "$0")/.."
O=gpurun_out/r06final
cp $O/bench_driver_shape.json profiles/r06_bench_driver_shape.json
cp $O/bench_long.json profiles/r06_bench_long.json
for W in transe_l2_fb15k rotate_wide; do cp $O/kernel_stats_$W.txt profiles/r06_kernel_stats_$W.txt; done
python tools/kernel_stats_json.py profiles/r06_kernel_stats_transe_l2_fb15k.txt r06 transe_l2_fb15k > /dev/null
for L in transe_l2_fb15k complex_wikikg2 rotate_wide rotate_freebase_a2a rotate_freebase_p2p; do
  for c in FETCH_SIZE WRITE_SIZE; do cp gpurun_out/r06_${L}_pmc_$c.txt profiles/r06_${L}_pmc_$c.txt; done
  python tools/traffic_json.py profiles/r06_${L}_pmc_FETCH_SIZE.txt profiles/r06_${L}_pmc_WRITE_SIZE.txt r06 $L 120 profiles/latest_traffic_$L.json
done
python tools/traffic_json.py profiles/r06_transe_l2_fb15k_pmc_FETCH_SIZE.txt profiles/r06_transe_l2_fb15k_pmc_WRITE_SIZE.txt r06 transe_l2_fb15k 120 profiles/latest_traffic.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob('profiles/latest_*.json')):
    d = json.load(open(f)); print(f, d.get('source_hash'), d.get('hbm_bytes_per_step'))
PY
