"""Prove -> serialise -> verify, through the C ABI.  The verifier (csrc/verify.hip via zkdl_amd/proof.py) sees only
public data: generators, the layers' commitments, the network's input and output, the hash-derived challenges and the
serialised proof.  An honest proof verifies; a proof with ONE flipped limb anywhere does not; a proof replayed against
another statement does not."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def zk():
    from zkdl_amd.capi import MSM_PIPPENGER, Zk
    z = Zk(0)
    z.set_msm_mode(MSM_PIPPENGER)
    yield z
    z.close()


def build_and_prove(zk, oracle, dims, batch, seed):
    from mlp_util import enqueue_proof, setup_mlp_gpu, synth_mlp
    from zkdl_amd.capi import Transcript
    from zkdl_amd.proof import fs_challenges, statement_digest
    model = synth_mlp(oracle, dims, batch, seed=seed)
    st = setup_mlp_gpu(zk, model)
    coms = [fc.com() for fc in st["fcs"]]
    digest = statement_digest(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"])
    chal = fs_challenges(digest, dims, batch)
    tr = Transcript(zk)
    enqueue_proof(st, chal, tr)
    blob = tr.serialize()
    return model, st, coms, chal, blob


@pytest.mark.parametrize("dims,batch", [([6, 5, 4], 3), ([16, 32, 24, 8], 4), ([5, 3], 1)])
def test_honest_proof_verifies_and_tampering_is_rejected(zk, oracle, dims, batch):
    from zkdl_amd.proof import parse_transcript, verify_mlp
    model, st, coms, chal, blob = build_and_prove(zk, oracle, dims, batch, seed=31)
    ok, msg, open_claims = verify_mlp(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"], blob)   # challenges re-derived from the statement
    assert ok, msg
    L = len(dims) - 1
    assert len(open_claims) == 2 * (L - 1) + 5 * (L - 1)
    fr, aff, inf = parse_transcript(blob)
    n_fr, n_g1 = fr.shape[0], aff.shape[0]
    rng = np.random.default_rng(5)
    # one flipped bit in EVERY Fr value that enters a check, and in a sample of the G1 coordinates.
    # (The 32 + 16 partial_me values of each ReLU are outputs the reference's protocol never constrains: skipped.)
    from zkdl_amd.proof import layer_shapes, call_order, ceil_log2
    sh = layer_shapes(dims)
    kb = ceil_log2(1 << ceil_log2(batch))
    unconstrained = set()
    off = 0
    for kind, i in call_order(L):
        if kind == "fc":
            off += 3 * ceil_log2(sh[i]["in_p"]) + 4
        else:
            k = kb + ceil_log2(sh[i]["out_p"])
            a = off + 3 * (k + 5) + 1
            unconstrained |= set(range(a, a + 32))
            b = a + 32 + 3 * (k + 4) + 1
            unconstrained |= set(range(b, b + 16))
            off = b + 16 + 3 * k + 2
    assert off == n_fr
    bad = bytearray(blob)
    for idx in range(n_fr):
        if idx in unconstrained:
            continue
        pos = 24 + 32 * idx + int(rng.integers(0, 32))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        ok2, msg2, _ = verify_mlp(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"], bytes(bad), chal=chal)
        assert not ok2, ("Fr value %d accepted after tampering" % idx)
        assert "verification failed" in msg2 or "zkFC" in msg2 or "zkReLU" in msg2
        bad[pos] = blob[pos]
    for idx in range(n_g1):
        pos = 24 + 32 * n_fr + 96 * idx + int(rng.integers(0, 96))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        ok2, _, _ = verify_mlp(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"], bytes(bad), chal=chal)
        assert not ok2, ("G1 value %d accepted after tampering" % idx)
        bad[pos] = blob[pos]
    # an infinity flag
    pos = 24 + 32 * n_fr + 96 * n_g1
    bad[pos] ^= 1
    assert not verify_mlp(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"], bytes(bad), chal=chal)[0]
    bad[pos] = blob[pos]
    assert verify_mlp(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"], bytes(bad), chal=chal)[0]      # restored: verifies again
    # truncated / wrong-shape blobs
    assert not verify_mlp(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"], blob[:-1])[0]


def test_proof_is_bound_to_its_statement(zk, oracle):
    """The same proof against a different public output, a different input, or another layer's commitment fails:
    the challenges change with the statement digest, and the first / last call open the public tensors."""
    from zkdl_amd.proof import verify_mlp
    dims, batch = [16, 32, 24, 8], 4
    model, st, coms, chal, blob = build_and_prove(zk, oracle, dims, batch, seed=32)
    assert verify_mlp(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"], blob)[0]
    y = st["Yhat"].download()
    y[1, 0] ^= 1
    assert not verify_mlp(zk, dims, batch, st["gens"], coms, st["X"], zk.upload(y), blob)[0]                # statement digest changes
    assert not verify_mlp(zk, dims, batch, st["gens"], coms, st["X"], zk.upload(y), blob, chal=chal)[0]     # same challenges: Z~ check fails
    x = st["X"].download()
    x[2, 0] ^= 1
    assert not verify_mlp(zk, dims, batch, st["gens"], coms, zk.upload(x), st["Yhat"], blob, chal=chal)[0]  # X~ check fails
    c = coms[1].download()
    c[0], c[1] = c[1].copy(), c[0].copy()
    swapped = [coms[0], zk.upload(c), coms[2]]
    assert not verify_mlp(zk, dims, batch, st["gens"], swapped, st["X"], st["Yhat"], blob, chal=chal)[0]    # com(u_hi) check fails


def test_verifier_accepts_the_full_size_proof(zk):
    """BASELINE configs[2] (18M-parameter MLP, batch 256): prove with hash-derived challenges, serialise, verify."""
    import bench
    from zkdl_amd.capi import Transcript
    from zkdl_amd.proof import fs_challenges, statement_digest, verify_mlp
    dims, batch = bench.WORKLOADS["mlp18m_b256"]
    wl = bench.Workload(zk, dims, batch, seed=3)
    coms = [fc.com() for fc in wl.fcs]
    chal = fs_challenges(statement_digest(zk, dims, batch, wl.gens, coms, wl.X, wl.Yhat), dims, batch)
    tr = Transcript(zk)
    wl.prove(chal, tr)
    blob = tr.serialize()
    ok, msg, _ = verify_mlp(zk, dims, batch, wl.gens, coms, wl.X, wl.Yhat, blob)
    assert ok, msg
    bad = bytearray(blob)
    bad[24 + 32 * 100 + 5] ^= 4
    assert not verify_mlp(zk, dims, batch, wl.gens, coms, wl.X, wl.Yhat, bytes(bad), chal=chal)[0]


def test_activation_commitments_close_the_inter_layer_claims(zk, oracle):
    """The reference's future work, on top of its proof: commit Z_i / A_i, open them at the points the layer proofs leave as open claims,
    verify the openings against the commitments (zk_verify_open).  Honest: verifies.  An opening against another tensor's commitment, a
    tampered opening, or a claim that is not the committed tensor's evaluation: rejected."""
    from zkdl_amd.proof import (activation_claims, claimed_activation_values, commit_activations, open_activations, verify_activation_openings,
                                verify_mlp)
    dims, batch = [16, 32, 24, 8], 4
    model, st, coms, chal, blob = build_and_prove(zk, oracle, dims, batch, seed=33)
    ok, msg, open_claims = verify_mlp(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"], blob, chal=chal)
    assert ok, msg
    committed = commit_activations(zk, st["gens"], st["Z"], st["A"])
    claims = activation_claims(dims, batch, chal)
    assert [n for n, _ in claims] == ["A1", "Z1", "Z1", "A0", "Z0", "Z0"]          # fc2 | relu1, fc1 | relu0, fc0 (call order)
    openings = open_activations(zk, committed, claims)
    values = claimed_activation_values(dims, open_claims)
    public = {k: (g, com) for k, (g, t, com) in committed.items()}
    ok, msg = verify_activation_openings(zk, public, claims, openings, values)
    assert ok, msg
    # the opened values ARE the layer proofs' claims (same numbers from two independent computations: sumcheck transcript vs opening)
    for (name, val, _, _), want in zip(openings, values):
        assert np.array_equal(np.asarray(val).reshape(8), np.asarray(want).reshape(8)), name
    # another tensor's commitment
    swapped = dict(public)
    swapped["Z1"], swapped["A1"] = (public["Z1"][0], public["A1"][1]), (public["A1"][0], public["Z1"][1])
    assert not verify_activation_openings(zk, swapped, claims, openings, values)[0]
    # a tampered opening point
    bad = [(n, v, a.copy(), i) for n, v, a, i in openings]
    bad[2][2][3, 5] ^= 1
    assert not verify_activation_openings(zk, public, claims, bad, values)[0]
    # a false claim: the layer proof's value replaced (the opening then opens something else)
    wrong = [v.copy() for v in values]
    wrong[4][0] ^= 1
    assert not verify_activation_openings(zk, public, claims, openings, wrong)[0]
    # a cheating opening for that false claim: the prover cannot produce one from the committed tensor -- opening a different tensor fails T_0
    other = zk.fr_add(st["Z"][0], st["Z"][0])
    ret, com_eval, proof = zk.open(st["gens"][0], other, committed["Z0"][2], claims[4][1])
    aff, inf = zk.g1_to_affine(zk.upload(np.concatenate([com_eval.reshape(1, 36), proof])))
    ok, _ = zk.verify_open(st["gens"][0], committed["Z0"][2], claims[4][1], ret, aff, inf)
    assert not ok
