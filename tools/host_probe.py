#!/usr/bin/env python3
"""Host-side time of each prove call of the bench workload (no device wait inside the calls)."""
import os, sys, time
sys.path.insert(0, ".")
import torch  # noqa: F401  (loads the HIP runtime first, like bench.py)
from zkdl_amd.capi import Zk
import bench
zk = Zk(0)
dims, batch = bench.WORKLOADS[os.environ.get("WORKLOAD", "mlp18m_b256")]
wl = bench.Workload(zk, dims, batch, seed=0)
L = len(wl.shapes)
import os
for it in range(int(os.environ.get('ITERS','4'))):
    t0 = time.perf_counter()
    chal = wl.challenges(it)
    t1 = time.perf_counter()
    wl.tr.reset()
    rows = []
    for kind, i, ch in chal:
        a = time.perf_counter()
        if kind == "fc":
            x_in = wl.A[i - 1] if i > 0 else wl.X
            z_out = wl.Yhat if i == L - 1 else wl.Z[i]
            wl.fcs[i].prove(x_in, z_out, wl.gens[i], ch[0], ch[1], ch[2], wl.tr)
        else:
            wl.relus[i].prove(wl.Z[i], wl.A[i], ch, wl.tr)
        rows.append((kind, i, (time.perf_counter() - a) * 1e3))
    t2 = time.perf_counter()
    zk.sync()                      # Fr phase drained (worker lanes joined)
    t3 = time.perf_counter()
    wl.tr.flush()
    zk.sync()                      # G1 phase drained
    t4 = time.perf_counter()
    worst = max(rows, key=lambda r: r[2])
    print(f"iter {it}: worst call {worst[0]}{worst[1]} {worst[2]:.2f} ms; challenges {1e3*(t1-t0):.2f} ms, prove calls {1e3*(t2-t1):.2f} ms, Fr drain {1e3*(t3-t2):.2f} ms (Fr phase {1e3*(t3-t1):.2f}), G1 phase {1e3*(t4-t3):.2f} ms, total {1e3*(t4-t0):.2f}")
print(" ".join(f"{k}{i}:{ms:.2f}" for k, i, ms in rows))
