#!/usr/bin/env python3
"""Per-kernel-family device time of one 18M / batch-256 proof with and without the activation openings (zk_act_open), serial launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
from zkdl_amd.capi import MSM_PIPPENGER, Transcript, Zk
from zkdl_amd.proof import fs_challenges, prove_activation_openings, statement_digest

zk = Zk(0); zk.set_msm_mode(MSM_PIPPENGER)
dims, batch = bench.WORKLOADS["mlp18m_b256"]
wl = bench.Workload(zk, dims, batch, seed=21)
coms = [fc.com() for fc in wl.fcs]
seed = statement_digest(zk, dims, batch, wl.gens, coms, wl.X, wl.Yhat)
chal = fs_challenges(seed, dims, batch)
zk.set_flag("derive_rounds", 1)
tr = Transcript(zk)
for with_act in (1, 0, 1):
    for rep in range(2):
        wl.prove(chal, tr)
        if with_act: prove_activation_openings(zk, dims, batch, wl.gens, wl.Z, wl.A, chal, seed, tr)
        tr.flush()
    zk.sync()
    zk.set_flag("lanes", 0)
    zk.prof_reset(); zk.prof_enable(True)
    wl.prove(chal, tr)
    if with_act: prove_activation_openings(zk, dims, batch, wl.gens, wl.Z, wl.A, chal, seed, tr)
    tr.flush(); zk.sync()
    p = zk.prof_read(); zk.prof_enable(False); zk.set_flag("lanes", 1)
    print("with_act", with_act, "total %.2f ms" % sum(v["ms"] for v in p.values()))
    for k, v in sorted(p.items(), key=lambda kv: -kv[1]["ms"])[:16]:
        print("   %-22s %7.3f ms  n=%d" % (k, v["ms"], v["launches"]))

import time
for with_act in (0, 1, 1):
    for rep in range(3):
        zk.sync()
        t0 = time.perf_counter()
        wl.prove(chal, tr)
        t1 = time.perf_counter()
        if with_act: prove_activation_openings(zk, dims, batch, wl.gens, wl.Z, wl.A, chal, seed, tr)
        t2 = time.perf_counter()
        tr.flush()
        t3 = time.perf_counter()
        zk.sync()
        t4 = time.perf_counter()
        print("with_act %d: record layers %.2f ms, record act %.2f ms, flush call %.2f ms, sync %.2f ms, total %.2f" % (with_act, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t4-t0)*1e3))
os.environ["ZKDL_TRACE_ALLOC"] = "1"
