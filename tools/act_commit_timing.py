#!/usr/bin/env python3
"""Cost of binding the activations of the 18M-parameter MLP at batch 256 (zkdl_amd/proof.py: activation commitments)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench
from zkdl_amd.capi import MSM_PIPPENGER, Transcript, Zk
from zkdl_amd.proof import (activation_claims, claimed_activation_values, commit_activations, fs_challenges, open_activations, statement_digest,
                            verify_activation_openings, verify_mlp)

zk = Zk(0)
zk.set_msm_mode(MSM_PIPPENGER)
dims, batch = bench.WORKLOADS["mlp18m_b256"]
wl = bench.Workload(zk, dims, batch, seed=1)
coms = [fc.com() for fc in wl.fcs]
chal = fs_challenges(statement_digest(zk, dims, batch, wl.gens, coms, wl.X, wl.Yhat), dims, batch)
tr = Transcript(zk)
wl.prove(chal, tr)
blob = tr.serialize()
ok, msg, open_claims = verify_mlp(zk, dims, batch, wl.gens, coms, wl.X, wl.Yhat, blob, chal=chal)
assert ok, msg
for rep in range(2):
    zk.sync(); t0 = time.perf_counter()
    committed = commit_activations(zk, wl.gens, wl.Z, wl.A)
    zk.sync(); t1 = time.perf_counter()
    claims = activation_claims(dims, batch, chal)
    openings = open_activations(zk, committed, claims)
    zk.sync(); t2 = time.perf_counter()
public = {k: (g, com) for k, (g, t, com) in committed.items()}
ok, msg = verify_activation_openings(zk, public, claims, openings, claimed_activation_values(dims, open_claims))
t3 = time.perf_counter()
pairs = sum(t.n for _, t, _ in committed.values())
print("activation commitments: %d tensors, %.1f M pairs in %.1f ms (%.0f M pairs/s); %d openings in %.1f ms (synchronous API, one at a time); verification %s in %.2f s"
      % (len(committed), pairs / 1e6, (t1 - t0) * 1e3, pairs / (t1 - t0) / 1e6, len(claims), (t2 - t1) * 1e3, "OK" if ok else "FAILED: " + msg, t3 - t2))
zk.close()
