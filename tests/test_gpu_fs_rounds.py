"""Round-by-round Fiat-Shamir ("sound mode"; csrc/fs_rounds.hip): zk_fc_prove_fs / zk_relu_prove_fs derive every
sumcheck round challenge from a hash of the rounds before it, on the device, and the verifier re-derives them from the serialised
proof.  Checked here:

  * prove -> serialise -> verify (public data only) on small models whose tables exercise both the big-round kernels and the
    LDS-resident tail, ragged sizes included;
  * the transcript IS a reference transcript: fed the challenges this mode derived, the reference-parity entry points
    (zk_fc_prove / zk_relu_prove) and the CPU oracle reproduce it bit for bit (Fr) / point for point (G1, affine-canonical);
  * tampering: one flipped bit in any round message is rejected; so is a round message changed CONSISTENTLY with its own round
    check (c0 + d, c1 - 2d keeps p(0) + p(1)) -- the later challenges are recomputed from the changed proof, as an adaptive prover
    would have to, and the proof still fails;
  * the 18M-parameter MLP at batch 256 proves and verifies in this mode (its time per proof is printed and recorded by bench.py).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


@pytest.fixture(scope="module")
def zk():
    from zkdl_amd.capi import MSM_PIPPENGER, Zk
    z = Zk(0)
    z.set_msm_mode(MSM_PIPPENGER)
    yield z
    z.close()


def setup(zk, oracle, dims, batch, seed):
    from mlp_util import setup_mlp_gpu, synth_mlp
    from zkdl_amd.proof import statement_digest
    model = synth_mlp(oracle, dims, batch, seed=seed)
    st = setup_mlp_gpu(zk, model)
    coms = [fc.com() for fc in st["fcs"]]
    digest = statement_digest(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"])
    return model, st, coms, digest


def prove_fs(zk, st, digest):
    from zkdl_amd.capi import Transcript
    from zkdl_amd.proof import prove_mlp_fs
    tr = Transcript(zk)
    prove_mlp_fs(st["fcs"], st["relus"], st["gens"], st["X"], st["Z"], st["A"], st["Yhat"], digest, tr)
    blob = tr.serialize()
    fr, _ = tr.read()
    aff, inf = tr.read_affine()
    tr.close()
    return blob, fr, aff, inf


# [1100, 9] batch 2: in_p = 2048 > 1024 -> the inner-product sumcheck has a big round; [40, 70, 30] batch 20: 4096-element Hadamard
# tables and 2^17 / 2^16 bit tables (7 and 6 big rounds); [5, 3] batch 1: every table fits the tail kernel, k_bs = 0
@pytest.mark.parametrize("dims,batch", [([6, 5, 4], 3), ([40, 70, 30], 20), ([1100, 9], 2), ([5, 3], 1)])
def test_sound_mode_proves_verifies_and_is_a_reference_transcript(zk, oracle, dims, batch):
    from mlp_util import enqueue_proof, prove_mlp_oracle
    from zkdl_amd.capi import Transcript
    from zkdl_amd.proof import fs_round_mode_challenges, verify_mlp, verify_mlp_fs
    model, st, coms, digest = setup(zk, oracle, dims, batch, seed=41)
    blob, fr, aff, inf = prove_fs(zk, st, digest)
    ok, msg, claims = verify_mlp_fs(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"], blob)
    assert ok, msg
    # deterministic: the same statement gives the same proof
    blob2, *_ = prove_fs(zk, st, digest)
    assert blob2 == blob
    # the derived challenges, handed to the reference-parity entry points and to the oracle
    chal = fs_round_mode_challenges(digest, dims, batch, fr)
    tr = Transcript(zk)
    enqueue_proof(st, chal, tr)
    fr_r, _ = tr.read()
    aff_r, inf_r = tr.read_affine()
    assert eq(fr_r, fr) and eq(aff_r, aff) and eq(inf_r, inf)
    model["chal"] = chal
    fr_o, g1_o = prove_mlp_oracle(oracle, model)
    aff_o, inf_o = oracle.g1_to_affine(g1_o)
    assert eq(fr_o, fr), "Fr transcript differs from the oracle run on the derived challenges"
    assert eq(aff_o, aff) and eq(inf_o, inf)
    # with immediate execution (no recorder, no worker streams) the same bytes
    zk.set_flag("batch_layers", 0)
    zk.set_flag("lanes", 0)
    try:
        blob3, *_ = prove_fs(zk, st, digest)
    finally:
        zk.set_flag("batch_layers", 1)
        zk.set_flag("lanes", 1)
    assert blob3 == blob
    # the non-adaptive verifier with the SAME challenge values accepts too (the checks are the same; only where the coins come from differs)
    assert verify_mlp(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"], blob, chal=chal)[0]


def round_message_offsets(dims, batch):
    """(offset of c0 of round r in the Fr transcript, rounds) for every sumcheck of the proof, in call order."""
    from zkdl_amd.proof import call_order, ceil_log2, layer_shapes
    sh = layer_shapes(dims)
    kb = ceil_log2(1 << ceil_log2(batch))
    out, off = [], 0
    for kind, i in call_order(len(sh)):
        if kind == "fc":
            k_in = ceil_log2(sh[i]["in_p"])
            out.append(("fc%d ip" % i, off, k_in))
            off += 3 * k_in + 4
        else:
            k = kb + ceil_log2(sh[i]["out_p"])
            out.append(("relu%d mag" % i, off, k + 5))
            off += 3 * (k + 5) + 1 + 32
            out.append(("relu%d rem" % i, off, k + 4))
            off += 3 * (k + 4) + 1 + 16
            out.append(("relu%d hp" % i, off, k))
            off += 3 * k + 2
    return out


def fr_int(limbs):
    return int.from_bytes(np.ascontiguousarray(limbs, dtype=np.uint32).tobytes(), "little")


def int_limbs(v):
    return np.frombuffer(int(v % R_MOD).to_bytes(32, "little"), dtype=np.uint32).copy()


def test_changed_round_messages_are_rejected_even_with_recomputed_challenges(zk, oracle):
    from zkdl_amd.proof import verify_mlp_fs
    dims, batch = [16, 32, 24, 8], 4
    model, st, coms, digest = setup(zk, oracle, dims, batch, seed=42)
    blob, fr, aff, inf = prove_fs(zk, st, digest)
    args = (zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"])
    assert verify_mlp_fs(*args, blob)[0]
    rng = np.random.default_rng(9)
    # Montgomery form: the limbs hold v * R mod r, and addition is linear in that representation, so d can be added to the limbs' integer
    for name, off, k in round_message_offsets(dims, batch):
        for r in range(k):
            # (a) one flipped bit in each coefficient of the round
            for q in range(3):
                bad = bytearray(blob)
                pos = 24 + 32 * (off + 3 * r + q) + int(rng.integers(0, 32))
                bad[pos] ^= 1 << int(rng.integers(0, 8))
                ok, msg, _ = verify_mlp_fs(*args, bytes(bad))
                assert not ok, (name, r, q)
            # (b) a change the round's own check cannot see: c0 + d, c1 - 2 d keeps p(0) + p(1) = 2 c0 + c1 + c2 -- and, for the eq-weighted
            # rounds, (1 - u) p(0) + u p(1) changes, so only the inner-product sumcheck has this freedom; there the NEXT round (or the final
            # product) must catch it, with challenges recomputed from the changed message
            if name.endswith("ip"):
                d = int(rng.integers(1, 1 << 62))
                f2 = fr.copy()
                f2[off + 3 * r] = int_limbs(fr_int(fr[off + 3 * r]) + d)
                f2[off + 3 * r + 1] = int_limbs(fr_int(fr[off + 3 * r + 1]) - 2 * d)
                bad = bytearray(blob)
                bad[24:24 + 32 * f2.shape[0]] = f2.tobytes()
                ok, msg, _ = verify_mlp_fs(*args, bytes(bad))
                assert not ok, (name, r, "consistent change")
                assert ("round %d:" % r) not in msg            # round r's own check passes; a LATER check catches it
    # a proof made for another statement does not verify against this one
    model2, st2, coms2, digest2 = setup(zk, oracle, dims, batch, seed=43)
    blob_other, *_ = prove_fs(zk, st2, digest2)
    assert not verify_mlp_fs(*args, blob_other)[0]


def test_sound_mode_at_full_size(zk):
    """BASELINE configs[2] in the sound mode: proves, verifies, equals the reference-parity transcript made from the derived challenges."""
    import time
    import bench
    from zkdl_amd.capi import Transcript
    from zkdl_amd.proof import fs_round_mode_challenges, prove_mlp_fs, statement_digest, verify_mlp_fs
    dims, batch = bench.WORKLOADS["mlp18m_b256"]
    wl = bench.Workload(zk, dims, batch, seed=13)
    coms = [fc.com() for fc in wl.fcs]
    digest = statement_digest(zk, dims, batch, wl.gens, coms, wl.X, wl.Yhat)
    tr = Transcript(zk)
    prove_mlp_fs(wl.fcs, wl.relus, wl.gens, wl.X, wl.Z, wl.A, wl.Yhat, digest, tr)       # first: materialises the bit tensors, warms the pools
    blob = tr.serialize()
    fr, _ = tr.read()
    aff, inf = tr.read_affine()
    ok, msg, _ = verify_mlp_fs(zk, dims, batch, wl.gens, coms, wl.X, wl.Yhat, blob)
    assert ok, msg
    zk.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        prove_mlp_fs(wl.fcs, wl.relus, wl.gens, wl.X, wl.Z, wl.A, wl.Yhat, digest, tr)
        tr.flush()
    zk.sync()
    ms = (time.perf_counter() - t0) * 1e3 / 3
    assert tr.serialize() == blob
    print("sound mode (round-by-round Fiat-Shamir), 18M MLP batch 256: %.1f ms per proof (synchronous)" % ms)
    assert ms < 25600.0                                       # north_star: <= 0.1 s per example
    chal = fs_round_mode_challenges(digest, dims, batch, fr)
    wl.prove(chal, tr)
    fr_r, _ = tr.read()
    aff_r, inf_r = tr.read_affine()
    assert eq(fr_r, fr) and eq(aff_r, aff) and eq(inf_r, inf)
    bad = bytearray(blob)
    bad[24 + 32 * 40 + 2] ^= 8
    assert not verify_mlp_fs(zk, dims, batch, wl.gens, coms, wl.X, wl.Yhat, bytes(bad))[0]
    tr.close()
