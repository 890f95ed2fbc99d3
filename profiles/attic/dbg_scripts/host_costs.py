#!/usr/bin/env python3
"""tools/dbg/host_costs.py - host-side cost of the calls that bracket bench.py's timed region (idle GPU): torch.cuda.synchronize(),
Event.record(), a stream synchronise, an empty hipGraph replay + synchronise, time.perf_counter()."""
import time
import torch
dev = torch.device("cuda", 0)
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()


def t(fn, n=200):
    fn()
    a = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - a) / n * 1e6


ev = torch.cuda.Event(enable_timing=True)
s = torch.cuda.current_stream()
print("perf_counter            %.2f us" % t(time.perf_counter))
print("torch.cuda.synchronize  %.2f us (idle)" % t(torch.cuda.synchronize))
print("stream.synchronize      %.2f us (idle)" % t(s.synchronize))
print("event.record            %.2f us" % t(ev.record))
print("event.record + sync     %.2f us" % t(lambda: (ev.record(), torch.cuda.synchronize())))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(80):
        x.add_(1.0)
torch.cuda.synchronize()
print("80-kernel graph replay (host call only) %.2f us" % t(g.replay, 50))
torch.cuda.synchronize()
print("80-kernel graph replay + synchronize     %.2f us" % t(lambda: (g.replay(), torch.cuda.synchronize()), 50))
k1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(k1):
    x.add_(1.0)
torch.cuda.synchronize()
print("1-kernel graph replay + synchronize      %.2f us" % t(lambda: (k1.replay(), torch.cuda.synchronize()), 200))
print("1 eager kernel + synchronize             %.2f us" % t(lambda: (x.add_(1.0), torch.cuda.synchronize()), 200))
