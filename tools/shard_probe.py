#!/usr/bin/env python3
"""Where a sharded proof's time goes at N = 1: host time of every phase of zkdl_amd/shard.py with a device sync behind it."""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch  # noqa: F401
from zkdl_amd.capi import MSM_PIPPENGER, Transcript, Zk
from zkdl_amd.shard import Shard
import bench
zk = Zk(0); zk.set_msm_mode(MSM_PIPPENGER)
dims, batch = bench.WORKLOADS["mlp18m_b256"]
wl = bench.Workload(zk, dims, batch, seed=0)
shard = Shard(wl, 0, 1)
trA, trB = Transcript(zk), Transcript(zk)


def chal(step):
    zk.seed(0xFACE + step)
    L = len(wl.shapes)
    kb = bench.ceil_log2(batch)
    out = []
    for kind, i in [("fc", L - 1)] + [x for i in range(L - 2, -1, -1) for x in (("relu", i), ("fc", i))]:
        s = wl.shapes[i]
        if kind == "fc":
            out.append((kind, i, [zk.random_vec(kb), zk.random_vec(bench.ceil_log2(s["in_p"])), zk.random_vec(bench.ceil_log2(s["out_p"]))]))
        else:
            k = kb + bench.ceil_log2(s["out_p"])
            out.append((kind, i, [zk.random_vec(n) for n in (k + 5, k + 5, k + 4, k + 4, k, k, k)]))
    return out


def timed(name, fn, acc):
    zk.sync(); t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); zk.sync(); t2 = time.perf_counter()
    acc.setdefault(name, []).append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    return r


acc = {}
for it in range(6):
    c = timed("challenges", lambda: chal(it), acc)
    stA = timed("local_enqueue", lambda: shard.local_enqueue(c, trA), acc)
    g = [timed("local_collect", lambda: shard.local_collect(stA, trA, keep_on_device=True), acc)]
    stB = timed("owner_enqueue", lambda: shard.owner_enqueue(c, g, trB), acc)
    own = timed("owner_collect", lambda: shard.owner_collect(stB, trB), acc)
    timed("assemble", lambda: shard.assemble(c, g, own), acc)
# per-kernel device time of one owner phase (worker streams off: every launch alone)
c = chal(99)
stA = shard.local_enqueue(c, trA)
g = [shard.local_collect(stA, trA, keep_on_device=True)]
zk.sync()
zk.set_flag("lanes", 0)
zk.prof_reset(); zk.prof_enable(True)
stB = shard.owner_enqueue(c, g, trB)
zk.sync()
p = zk.prof_read(); zk.prof_enable(False)
zk.set_flag("lanes", 1)
shard.owner_collect(stB, trB)
for k, v in sorted(p.items(), key=lambda kv: -kv[1]["ms"])[:14]:
    print("   owner phase  %-22s %8.3f ms  %4d launches" % (k, v["ms"], v["launches"]))
for k, v in acc.items():
    v = v[2:]
    print("%-16s host %7.3f ms   host + device %7.3f ms" % (k, sum(a for a, _ in v) / len(v), sum(b for _, b in v) / len(v)))
