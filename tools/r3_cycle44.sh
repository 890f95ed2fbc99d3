#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/wake_probe.hip -o /tmp/wake_probe.bin 2>/dev/null
timeout 120 /tmp/wake_probe.bin 2>&1 | tee $O/wake_probe.txt
