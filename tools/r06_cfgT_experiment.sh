#!/bin/bash
# round 6, VERDICT r05 next-3: bounded experiment on the cfg-T step.
#  (1) the driver's shape (120 untimed, 20 timed) against a longer untimed run-in (--min-untimed 1000) and the long run: where the
#      driver-vs-own-box gap comes from;
#  (2) the SAME 4-launch step on 5/8 of the chip (HSA_CU_MASK, 160 of 256 CUs): what an XCD-partitioned persistent kernel - chunk c's
#      forward -> loss -> backward pinned to XCD c, C = 5 chunks on 8 XCDs - would pay for leaving three XCDs idle, before it gains
#      anything from XCD-local operands and the two launch boundaries it removes (<= 2 x 1.4 us + ~1.5 us of L2-local first loads).
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_cfgT; mkdir -p $O; cd $R
B="python bench.py --no-cpu-baseline --no-configs --hogwild 0 --no-async-update"
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', 'us/step', round(1e3*d['ms_per_step'],3), 'event', round(1e3*d['roofline']['event_ms_per_step'],3), d['config'].get('sampler_groups_timed'))"; }
for rep in 1 2 3; do timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "driver-shape(120 untimed) rep$rep"; done
for rep in 1 2 3; do timeout 200 $B --steps 20 --warmup 5 --min-untimed 1000 2>/dev/null | line "driver-shape(1000 untimed) rep$rep"; done
for rep in 1 2; do timeout 200 $B --steps 3000 --warmup 300 2>/dev/null | line "long-run rep$rep"; done
for M in "0:0-159" "0:0-191" "0:0-127"; do
  for rep in 1 2; do HSA_CU_MASK=$M timeout 200 $B --steps 3000 --warmup 300 2>/dev/null | line "HSA_CU_MASK=$M long-run rep$rep"; done
done
cd /tmp; export TMPDIR=/tmp
for M in "" "0:0-159"; do
  rm -rf /tmp/prof_m
  HSA_CU_MASK=$M timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_m -- $B --steps 1200 --warmup 120 > /tmp/prof_m.log 2>&1 || ( cd $R; HSA_CU_MASK=$M timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_m -- python $R/bench.py --no-cpu-baseline --no-configs --hogwild 0 --no-async-update --steps 1200 --warmup 120 > /tmp/prof_m.log 2>&1 )
  echo "== kernel stats, HSA_CU_MASK='$M'"; python $R/tools/rocpd_stats.py $(ls /tmp/prof_m/*/*_results.db | head -1) | head -6 | cut -c1-70,76-130
done
