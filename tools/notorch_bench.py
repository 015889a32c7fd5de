"""The bench loop without torch in the process (ctypes only): does torch's GPU context cost the pipeline anything?
usage: python tools/notorch_bench.py [steps] [inflight]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (module level: no torch import)
from zkdl_amd.capi import Zk  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 4
assert "torch" not in sys.modules
zk = Zk(0)
zk.set_flag("derive_rounds", 1)
dims, batch = bench.WORKLOADS["mlp18m_b256"]
wl = bench.Workload(zk, dims, batch, seed=0, inflight=inflight)
rows = zk.empty(2 * (steps + 16), 8)      # 64 bytes per proof


def step(i):
    tr = wl.trs[i % len(wl.trs)]
    wl.prove(wl.challenges(i), tr)
    tr.seal(rows.ptr + 64 * (i % (steps + 16)))


for i in range(120):
    step(-1 - i)
    if i % 8 == 7:
        zk.sync()
zk.sync()
t0 = time.perf_counter()
for i in range(steps):
    if i == steps - 1:
        zk.set_flag("derive_rounds", 0)
    step(i)
zk.sync()
dt = time.perf_counter() - t0
print("no torch in the process: %.3f ms per proof over %d proofs (torch loaded: %s)" % (dt * 1e3 / steps, steps, "torch" in sys.modules))
