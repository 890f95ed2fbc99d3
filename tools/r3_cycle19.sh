#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
for envs in "" "NCCL_GRAPH_REGISTER=0" "NCCL_GRAPH_MIXING_SUPPORT=0" "TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0" "RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0"; do
echo "== env: [$envs]"
env $envs MASTER_PORT=$((29600 + RANDOM % 200)) timeout 45 python tools/dbg/rccl_capture_probe.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
echo "rc=$?"
done
