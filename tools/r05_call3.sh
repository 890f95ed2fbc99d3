#!/bin/bash
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05c3; mkdir -p $O
env KGE_FAULTHANDLER=45 KGE_DIST_FORCE_COLL=1 KGE_DIST_MODE=a2a KGE_DIST_OTHER_LEG=0 KGE_DIST_GRAPH=1 KGE_DIST_GRAPH_TIMEOUT=60 timeout -s KILL 100 python bench.py --workload rotate_freebase --steps 600 --warmup 40 --no-cpu-baseline > $O/proxy_graph.json 2> $O/proxy_graph.err
echo "rc=$?"; grep -v "amdgpu.ids" $O/proxy_graph.err | tail -80; cut -c1-300 $O/proxy_graph.json
