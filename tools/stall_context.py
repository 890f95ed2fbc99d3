#!/usr/bin/env python3
"""tools/stall_context.py DB: the largest stall between consecutive kernel dispatches inside the last burst of a rocprofv3
--sys-trace run, and every traced API call / memory copy that overlaps it (who was talking to the GPU while it waited?)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print("tables/views:", [t for t in tabs if not t.startswith("rocpd_")][:40])
kt = [t for t in tabs if t.startswith("kernels")][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kt)]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, start, end from %s order by start" % (name_col, kt)).fetchall()
first = [i for i, r in enumerate(rows) if "neg_fwd_edge_kernel" in r[0]]
lo = first[-20]
gaps = [(rows[j + 1][1] - rows[j][2], j) for j in range(lo, len(rows) - 1) if "neg_" in rows[j + 1][0] or "update" in rows[j + 1][0] or "loss" in rows[j + 1][0]]
g, j = max(gaps)
ta, tb = rows[j][2], rows[j + 1][1]
print("stall %.1f us after %s (+%.1f us since the burst's first step kernel)" % (g / 1e3, rows[j][0][:30], (ta - rows[lo][1]) / 1e3))
for t in tabs:
    if t.startswith("rocpd_") or t == kt:
        continue
    try:
        c = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
        if "start" in c and "end" in c:
            nm = "name" if "name" in c else ([x for x in c if "name" in x] or [c[0]])[0]
            q = cur.execute("select %s, start, end from %s where end >= ? and start <= ? order by start" % (nm, t), (ta - 100000, tb + 20000)).fetchall()
            for r in q[:60]:
                print("  %-22s %-40s start %+9.1f us  dur %8.1f us" % (t[:22], str(r[0])[:40], (r[1] - ta) / 1e3, (r[2] - r[1]) / 1e3))
    except Exception as e:
        print("  (%s: %s)" % (t, e))
