#!/usr/bin/env python3
"""One proof of the bench workload with the library's event profiler on; with ZKDL_PROF_TRACE=<file> set the
library appends a per-launch timeline that tools/prof_timeline.py summarises."""
import os, sys
sys.path.insert(0, ".")
import torch  # noqa: F401
from zkdl_amd.capi import Zk
import bench
zk = Zk(0)
dims, batch = bench.WORKLOADS[os.environ.get("WORKLOAD", "mlp18m_b256")]
wl = bench.Workload(zk, dims, batch, seed=0)
if os.environ.get('LANES') == '0':
    zk.set_flag('lanes', 0)                # every launch alone on the chip
for it in range(3):                      # warm pools
    wl.prove(wl.challenges(it)); wl.tr.flush(); zk.sync()
path = os.environ.get("ZKDL_PROF_TRACE")
if path and os.path.exists(path): os.remove(path)
zk.prof_reset(); zk.prof_enable(True)
pipelined = os.environ.get("PIPE", "0") == "1"
for it in range(4 if pipelined else 1):
    tr = wl.trs[it & 1]
    wl.prove(wl.challenges(10 + it), tr)
    if pipelined: tr.seal()
    else: tr.flush()
zk.sync()
zk.prof_read(); zk.prof_enable(False)
