#!/usr/bin/env python3
"""ONE sealed proof of the bench workload with nothing else in flight, in the bench's latency configuration (digit tables, g1_batch 1,
one G1 stream), with the library's event profiler on: ZKDL_PROF_TRACE=<file> gets the per-launch timeline (tools/prof_timeline.py).
DERIVE=1 derives the round commitments instead of running them as MSMs."""
import os, sys, time
sys.path.insert(0, ".")
import torch  # noqa: F401
from zkdl_amd.capi import MSM_PIPPENGER, Zk
import bench
zk = Zk(0); zk.set_msm_mode(MSM_PIPPENGER)
dims, batch = bench.WORKLOADS["mlp18m_b256"]
wl = bench.Workload(zk, dims, batch, seed=0, inflight=2, full_tables=int(os.environ.get("FT", "12")))
zk.set_flag("derive_rounds", int(os.environ.get("DERIVE", "0")))
zk.set_flag("g1_batch", 1); zk.set_flag("g1_streams", 1)
rows = torch.zeros((16, 8), dtype=torch.int64, device="cuda")
for it in range(4):
    tr = wl.trs[it & 1]
    wl.prove(wl.challenges(it), tr); tr.seal(rows.data_ptr() + 64 * it); zk.sync()
if os.environ.get("REPS"):
    import numpy as np
    ts = []
    for it in range(int(os.environ["REPS"])):
        tr = wl.trs[it & 1]
        zk.sync(); t0 = time.perf_counter()
        wl.prove(wl.challenges(100 + it), tr); tr.seal(rows.data_ptr() + 64 * (it & 7)); zk.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("latency ms: median %.3f min %.3f" % (float(np.median(ts)), min(ts)))
    sys.exit(0)
path = os.environ.get("ZKDL_PROF_TRACE")
if path and os.path.exists(path): os.remove(path)
zk.prof_reset(); zk.prof_enable(True)
t0 = time.perf_counter()
tr = wl.trs[0]
wl.prove(wl.challenges(10), tr)
t1 = time.perf_counter()
tr.seal(rows.data_ptr() + 64 * 8)
t2 = time.perf_counter()
zk.sync()
t3 = time.perf_counter()
print("host: record %.3f ms, seal (enqueue) %.3f ms, wait %.3f ms, total %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3))
zk.prof_read(); zk.prof_enable(False)
