#!/usr/bin/env python3
"""tools/kernel_gaps.py DB [skip]: per kernel name of a rocprofv3 kernel trace - calls, mean duration and the mean gap between
the previous kernel's end and this kernel's start on the time axis (all queues merged; negative = overlap with the previous
kernel, i.e. concurrency).  The step kernels only (names with neg_/loss_/update_/edge_/sample_)."""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kt)]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, start, end from %s order by start" % (name_col, kt)).fetchall()
keys = ("neg_", "loss_", "update_", "edge_", "sample_", "gather_req", "apply_merged", "route_", "gn_reduce", "transr_", "rescal_")
seq = [(n.split("(")[0].replace("void ", "")[:44], s, e) for n, s, e in rows if any(k in n for k in keys)]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(seq) // 3
seq = seq[skip:]
st = {}
prev_end = None
for n, s, e in seq:
    d = st.setdefault(n, {"dur": [], "gap": []})
    d["dur"].append((e - s) / 1e3)
    if prev_end is not None:
        d["gap"].append((s - prev_end) / 1e3)
    prev_end = max(prev_end or 0, e)
print("%-44s %6s %8s %8s %8s" % ("kernel", "calls", "dur_us", "gap_p50", "gap_mean"))
tot = 0.0
for n, d in sorted(st.items(), key=lambda kv: -len(kv[1]["dur"])):
    g = np.array(d["gap"]) if d["gap"] else np.zeros(1)
    print("%-44s %6d %8.2f %8.2f %8.2f" % (n, len(d["dur"]), np.mean(d["dur"]), np.percentile(g, 50), g.mean()))
span = (seq[-1][2] - seq[0][1]) / 1e3
print("span %.1f us over %d dispatches" % (span, len(seq)))
