#!/usr/bin/env python3
"""tools/traffic_json.py FETCH.txt WRITE.txt BUILD WORKLOAD [G] [OUT] -> profiles/latest_traffic.json (or OUT, e.g.
profiles/latest_traffic_<leg>.json: what bench.py reports as that leg's roofline.traffic)
(per-launch PMC means from tools/rocpd_stats.py --pmc; FETCH_SIZE doubled per MI355X_MICROARCH.md).
Only the kernels of the training step are summed; the sampler kernel builds G batches per launch
(G = --graph-steps of the profiled run, default 120) and is added as 1/G of its per-launch traffic."""
import json, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import source_hash
def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m and "kernel<" in m.group(1) or (m and "_kernel" in m.group(1)):
            out[m.group(1).strip()] = float(m.group(5))
    return out
f, w = parse(sys.argv[1]), parse(sys.argv[2])
G = int(sys.argv[5]) if len(sys.argv) > 5 else 120
# (once-per-GROUP kernels - the sampler and the a2a engine's routing - count 1 / G of their per-launch traffic)
ours = lambda d: {k: (v / G if ("sample_plan" in k or "route_" in k) else v) for k, v in d.items()
                  if "at::native" not in k and "rocclr" not in k and "reduce_acc" not in k and "rocprim" not in k
                  and "randperm" not in k and "elementwise" not in k and "anonymous namespace" not in k}
f, w = ours(f), ours(w)
tot = sum(2 * v * 1024 for v in f.values()) + sum(v * 1024 for v in w.values())
json.dump({"build": sys.argv[3], "source_hash": source_hash(), "workload": sys.argv[4], "fetch_kb_per_launch": f, "write_kb_per_launch": w,
           "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE uncorrected",
           "hbm_bytes_per_step": tot}, open(sys.argv[6] if len(sys.argv) > 6 else "profiles/latest_traffic.json", "w"), indent=1)
print("hbm MB/step", tot / 1e6)
