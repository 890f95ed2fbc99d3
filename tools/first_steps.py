#!/usr/bin/env python3
"""tools/first_steps.py DB: kernel durations and gaps of the LAST n steps of a rocprofv3 kernel trace of `bench.py --steps n`,
in launch order - do the first steps after the synchronise that opens the timed region run slower than the later ones?"""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels")][0] if any(t.startswith("kernels") for t in tabs) else None
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kt)]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, start, end from %s order by start" % (name_col, kt)).fetchall()
upd = [i for i, r in enumerate(rows) if "update_kernel" in r[0]]
first = [i for i, r in enumerate(rows) if "neg_fwd_edge_kernel" in r[0] or "edge_fwd_kernel" in r[0]]
last_steps = first[-n:]
t0 = rows[last_steps[0]][1]
print("step  start_us   first    loss     bwd   update   step_span  gap_to_next")
for k, i in enumerate(last_steps):
    seg = rows[i:i + 4]
    d = [(e - s) / 1e3 for _, s, e in seg]
    nxt = rows[i + 4][1] if i + 4 < len(rows) else seg[-1][2]
    print("%3d  %8.2f  %6.2f  %6.2f  %6.2f  %6.2f   %7.2f   %6.2f   %s" % (k, (seg[0][1] - t0) / 1e3, d[0], d[1], d[2], d[3],
          (seg[-1][2] - seg[0][1]) / 1e3, (nxt - seg[-1][2]) / 1e3, "" if k else seg[0][0][:30]))
# what ran right before the first of these steps
for j in range(max(0, last_steps[0] - 3), last_steps[0]):
    print("before: %-40s start %9.2f dur %7.2f" % (rows[j][0][:40], (rows[j][1] - t0) / 1e3, (rows[j][2] - rows[j][1]) / 1e3))
# every dispatch around the largest gap inside the window
lo, hi = last_steps[0], min(len(rows), last_steps[-1] + 4)
gaps = [(rows[j + 1][1] - rows[j][2], j) for j in range(lo, hi - 1)]
g, j = max(gaps)
print("largest gap inside the window: %.2f us after dispatch %d" % (g / 1e3, j - lo))
for q in range(max(lo, j - 6), min(hi, j + 7)):
    print("  %3d %-44s start %9.2f  dur %7.2f  gap_before %6.2f" % (q - lo, rows[q][0][:44], (rows[q][1] - t0) / 1e3, (rows[q][2] - rows[q][1]) / 1e3,
          (rows[q][1] - rows[q - 1][2]) / 1e3))
# every kernel between the first dispatch of the window and the end of the trace, by name (round 5: is there a sampler launch among or
# behind the timed steps?)
cnt = {}
for q in range(lo, len(rows)):
    k = rows[q][0].split("(")[0].replace("void ", "")[:48]
    cnt[k] = cnt.get(k, 0) + 1
print("dispatches from the window's first kernel to the end of the trace:")
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("  %4d  %s" % (v, k))
