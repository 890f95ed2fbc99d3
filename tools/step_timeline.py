#!/usr/bin/env python3
"""tools/step_timeline.py DB: per-dispatch view of a rocprofv3 kernel trace of bench.py - for every kernel of the
training step the distribution of durations and of the GAP to the previous kernel's end, split by step parity
(odd steps corrupt tails, even steps heads) - where does the step time that is not in the kernel averages go?"""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
names = ["edge_fwd", "neg_fwd_gemm", "loss_kernel", "neg_bwd_gemm", "update_kernel"]
seq = [(n, s, e) for n, s, e in rows if any(k in n for k in names)]
# steps = consecutive groups starting at edge_fwd
steps, cur_ = [], []
for n, s, e in seq:
    if "edge_fwd" in n and cur_:
        steps.append(cur_); cur_ = []
    cur_.append((n, s, e))
steps.append(cur_)
steps = [st for st in steps if len(st) == 5][50:]
print("steps analysed:", len(steps))
dur = np.array([[e - s for _, s, e in st] for st in steps]) / 1e3
gap = np.array([[st[i][1] - st[i - 1][2] for i in range(1, 5)] for st in steps]) / 1e3
inter = np.array([steps[i + 1][0][1] - steps[i][4][2] for i in range(len(steps) - 1)]) / 1e3
span = np.array([st[4][2] - st[0][1] for st in steps]) / 1e3
print("kernel        mean   p10   p50   p90   (us)")
for i, n in enumerate(names):
    d = dur[:, i]
    print("%-13s %5.2f %5.2f %5.2f %5.2f" % (n, d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90)))
print("gaps inside a step (end -> next start): mean", gap.mean(0).round(2), " p50", np.percentile(gap, 50, axis=0).round(2))
print("gap between steps (update end -> next edge_fwd start): mean %.2f p50 %.2f p90 %.2f" % (inter.mean(), np.percentile(inter, 50), np.percentile(inter, 90)))
print("step span (edge_fwd start -> update end): mean %.2f p50 %.2f;  sum of kernel means %.2f;  step period %.2f" % (
    span.mean(), np.percentile(span, 50), dur.mean(0).sum(), span.mean() + inter.mean()))
odd = dur[0::2].mean(0); even = dur[1::2].mean(0)
print("by alternate steps (head / tail corruption): ", odd.round(2), even.round(2))
