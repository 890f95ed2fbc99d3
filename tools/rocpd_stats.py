#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (`--kernel-trace --stats` output) as a per-kernel
table: calls, total / average / min / max duration, share of GPU time.  Usage:
    python tools/rocpd_stats.py gpurun_out/prof/run/NNN_results.db > profiles/xxx_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        d = e - s
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print("%-72s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) <= 72 else name[:69] + "..."
        print("%-72s %8d %12.1f %10.3f %10.3f %10.3f %6.2f" % (short, a[0], a[1] / 1e3, a[1] / a[0] / 1e3,
                                                             a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))
    print("total GPU kernel time: %.1f us over %d dispatches" % (tot / 1e3, len(rows)))


def pmc(path):
    """per-kernel mean of every collected counter (rocprofv3 --pmc ...)."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    print("# counters_collection columns:", cols)
    namec = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "counter" not in c][0]
    cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    val = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    rows = cur.execute("select %s, %s, count(*), sum(%s) from counters_collection group by %s, %s"
                       % (namec, cname, val, namec, cname)).fetchall()
    print("%-64s %-14s %8s %16s %14s" % ("kernel", "counter", "samples", "sum", "mean"))
    for k, c, n, sm in sorted(rows, key=lambda r: -(r[3] or 0)):
        short = k if len(k) <= 64 else k[:61] + "..."
        print("%-64s %-14s %8d %16.1f %14.2f" % (short, c, n, sm, sm / max(n, 1)))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--pmc":
        pmc(sys.argv[1])
    else:
        main(sys.argv[1])
