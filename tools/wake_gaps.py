#!/usr/bin/env python3
"""tools/wake_gaps.py DB: every stall of >= 5 us between consecutive dispatches of a rocprofv3 kernel trace, with the time since the
GPU last woke up (= since the last idle period of >= 100 us) and the number of dispatches since then."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels")][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kt)]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, start, end from %s order by start" % (name_col, kt)).fetchall()
wake_t, wake_i = rows[0][1], 0
print("%-10s %-12s %-9s %-9s %s" % ("gap_us", "since_wake_us", "dispatch#", "busy_us", "next kernel"))
busy = 0.0
for i in range(1, len(rows)):
    gap = (rows[i][1] - rows[i - 1][2]) / 1e3
    busy += (rows[i - 1][2] - rows[i - 1][1]) / 1e3
    if gap >= 100.0:
        print("---- idle %.0f us; previous burst: %d dispatches, %.0f us busy" % (gap, i - wake_i, busy))
        wake_t, wake_i, busy = rows[i][1], i, 0.0
    elif gap >= 5.0:
        print("%-10.2f %-12.2f %-9d %-9.1f %s" % (gap, (rows[i][1] - wake_t) / 1e3, i - wake_i, busy, rows[i][0][:40]))
