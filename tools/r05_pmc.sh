#!/bin/bash
# round 5: FETCH_SIZE / WRITE_SIZE passes at cfg-T, cfg-C (real wikikg2 table), cfg-R (local 1 M-row table, a2a engine, shard map)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT; cd $R
bash tools/pmc_cycle.sh r05_transe_l2_fb15k --workload transe_l2_fb15k | grep -c kernel
bash tools/pmc_cycle.sh r05_complex_wikikg2 --workload complex_wikikg2 | grep -c kernel
bash tools/pmc_cycle.sh r05_rotate_wide --workload rotate_wide | grep -c kernel
KGE_DIST_MODE=a2a KGE_DIST_OTHER_LEG=0 bash tools/pmc_cycle.sh r05_rotate_freebase_a2a --workload rotate_freebase | grep -c kernel
KGE_DIST_MODE=p2p KGE_DIST_OTHER_LEG=0 bash tools/pmc_cycle.sh r05_rotate_freebase_p2p --workload rotate_freebase | grep -c kernel
rm -f gpurun_out/r05_*_pmc_*.db
ls -la gpurun_out/r05_*_pmc_*.txt
