#!/bin/bash
# tools/build_variant1.sh NAME FILE [-DFLAG ...] : variant library that differs from the in-tree build in ONE source file
# (the other objects are the in-tree build's: run __graft_entry__.build() first)
NAME=$1; F=$2; shift; shift
cd "$(dirname "$0")/../dgl-ke_amd" && mkdir -p variants build/var_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c csrc/$F.hip -o build/var_$NAME/$F.o || exit 1
OBJS=$(ls build/*.o | grep -v "/$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/var_$NAME/$F.o -o variants/libkge_$NAME.so && echo built variants/libkge_$NAME.so
