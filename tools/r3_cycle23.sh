#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o /tmp/valu_rate_probe.bin 2>/dev/null
timeout 120 /tmp/valu_rate_probe.bin > $O/valu_rate_probe.txt 2>&1; cat $O/valu_rate_probe.txt
