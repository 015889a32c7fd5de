#!/usr/bin/env python3
"""Where does the host time of one proving step go?  (python challenge generation vs C-ABI calls)"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench
from zkdl_amd.capi import Zk
zk = Zk(0)
wl = bench.Workload(zk, bench.MLP18M, 256, 0)
for i in range(2):
    wl.prove(wl.challenges(i)); wl.tr.flush(); zk.sync()
t0 = time.perf_counter(); ch = [wl.challenges(10 + i) for i in range(5)]; t1 = time.perf_counter()
print("challenges: %.2f ms/step" % ((t1 - t0) * 200))
zk.sync()
t0 = time.perf_counter()
for c in ch:
    wl.prove(c); wl.tr.flush()
t1 = time.perf_counter(); zk.sync(); t2 = time.perf_counter()
print("prove+flush enqueue: %.2f ms/step ; drain after: %.2f ms total" % ((t1 - t0) * 200, (t2 - t1) * 1e3))
# per-call timing
import collections
acc = collections.Counter()
for c in ch[:2]:
    wl.tr.reset()
    L = len(wl.shapes)
    for kind, i, cc in c:
        t = time.perf_counter()
        if kind == "fc":
            x_in = wl.A[i - 1] if i > 0 else wl.X
            z_out = wl.Yhat if i == L - 1 else wl.Z[i]
            wl.fcs[i].prove(x_in, z_out, wl.gens[i], cc[0], cc[1], cc[2], wl.tr)
        else:
            wl.relus[i].prove(wl.Z[i], wl.A[i], cc, wl.tr)
        acc[kind] += time.perf_counter() - t
    t = time.perf_counter(); wl.tr.flush(); acc["flush"] += time.perf_counter() - t
    zk.sync()
print({k: round(v * 500, 2) for k, v in acc.items()}, "ms/step")
