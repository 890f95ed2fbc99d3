#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...] : build dgl-ke_amd/variants/libkge_NAME.so for A/B runs
NAME=$1; shift
cd "$(dirname "$0")/../dgl-ke_amd" && mkdir -p variants build/var_$NAME
for f in kge_rowwise kge_neg_gemm kge_neg_pair kge_neg_bcast kge_sampler kge_eval kge_rank_gemm kge_rescal kge_transr kge_route kge_api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c csrc/$f.hip -o build/var_$NAME/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/var_$NAME/*.o -o variants/libkge_$NAME.so && echo built variants/libkge_$NAME.so
