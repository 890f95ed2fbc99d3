#!/usr/bin/env python3
"""tools/kernel_stats_json.py STATS.txt BUILD WORKLOAD -> profiles/latest_kernel_stats.json
(per-kernel average launch durations from the rocprofv3 --kernel-trace --stats summary printed by tools/rocpd_stats.py;
bench.py copies the dominant kernel's duration into `roofline.dominant_kernel_us` so that the line is self-checking)."""
import json, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import source_hash
out = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if m and ("_kernel" in m.group(1)) and "at::native" not in m.group(1) and "rocprim" not in m.group(1) and \
            "anonymous namespace" not in m.group(1) and "rocclr" not in m.group(1):
        out[m.group(1).strip()] = {"calls": int(m.group(2)), "avg_us": float(m.group(4)), "pct": float(m.group(7))}
json.dump({"build": sys.argv[2], "source_hash": source_hash(), "workload": sys.argv[3], "kernels": out,
           "source": "rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline --steps 1200 --warmup 120`"},
          open("profiles/latest_kernel_stats.json", "w"), indent=1)
print(json.dumps({k: v["avg_us"] for k, v in out.items()}, indent=1))
