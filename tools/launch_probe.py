#!/usr/bin/env python3
"""Host-side cost per launch through the C ABI: N tiny zk_fr_add calls, enqueue time vs total."""
import sys, time, ctypes, numpy as np
sys.path.insert(0, ".")
from zkdl_amd.capi import Zk
zk = Zk(0)
a = zk.fr_random(64); b = zk.fr_random(64); out = zk.empty(64, 8)
lib = zk.lib
fn = lib.zk_fr_add
for N in (2000, 2000):
    zk.sync()
    t0 = time.perf_counter()
    for _ in range(N):
        fn(zk.ctx, a.ptr, b.ptr, out.ptr, 64)
    t1 = time.perf_counter(); zk.sync(); t2 = time.perf_counter()
    print(f"zk_fr_add x{N}: enqueue {1e6 * (t1 - t0) / N:.2f} us/call, total {1e6 * (t2 - t0) / N:.2f} us/call")
# sumcheck on small tensors (launch-count heavy)
for k in (10, 14, 18):
    t = zk.fr_random(1 << k); u = zk.random_vec(k); v = zk.random_vec(k)
    zk.sumcheck_binary(t, u, v); zk.sync()
    t0 = time.perf_counter()
    for _ in range(20): zk.sumcheck_binary(t, u, v)
    zk.sync(); t1 = time.perf_counter()
    print(f"binary sumcheck k={k}: {1e3 * (t1 - t0) / 20:.3f} ms each (includes proof download)")
