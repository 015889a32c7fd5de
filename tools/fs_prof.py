#!/usr/bin/env python3
"""Per-kernel-family device time of one 18M / batch-256 proof in the sound mode (round-by-round Fiat-Shamir) next to the throughput mode, standalone launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
from zkdl_amd.capi import MSM_PIPPENGER, Transcript, Zk
from zkdl_amd.proof import fs_challenges, prove_mlp_fs, statement_digest

zk = Zk(0); zk.set_msm_mode(MSM_PIPPENGER)
dims, batch = bench.WORKLOADS["mlp18m_b256"]
wl = bench.Workload(zk, dims, batch, seed=21, full_tables=12)
coms = [fc.com() for fc in wl.fcs]
seed = statement_digest(zk, dims, batch, wl.gens, coms, wl.X, wl.Yhat)
chal = fs_challenges(seed, dims, batch)
zk.set_flag("derive_rounds", 1)
tr = Transcript(zk)
for sound in (1, 0):
    for rep in range(2):
        if sound: prove_mlp_fs(wl.fcs, wl.relus, wl.gens, wl.X, wl.Z, wl.A, wl.Yhat, seed, tr)
        else: wl.prove(chal, tr)
        tr.flush()
    zk.sync()
    zk.set_flag("lanes", 0)
    zk.prof_reset(); zk.prof_enable(True)
    if sound: prove_mlp_fs(wl.fcs, wl.relus, wl.gens, wl.X, wl.Z, wl.A, wl.Yhat, seed, tr)
    else: wl.prove(chal, tr)
    tr.flush(); zk.sync()
    p = zk.prof_read(); zk.prof_enable(False); zk.set_flag("lanes", 1)
    print("sound" if sound else "throughput mode", "total %.2f ms, %d launches" % (sum(v["ms"] for v in p.values()), sum(v["launches"] for v in p.values())))
    for k, v in sorted(p.items(), key=lambda kv: -kv[1]["ms"])[:22]:
        print("   %-22s %7.3f ms  n=%d" % (k, v["ms"], v["launches"]))
