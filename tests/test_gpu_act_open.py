"""Activation commitments inside the proof (csrc/act_open.hip): the three activation-side claims of a hidden layer are reduced
to one evaluation of [Z_i ; A_i] by an inner-product sumcheck and opened ONCE; the openings' MSMs run in the proof's deferred batch."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    from zkdl_amd.capi import MSM_PIPPENGER, Zk
    z = Zk(0)
    z.set_msm_mode(MSM_PIPPENGER)
    yield z
    z.close()


def concat_coms(zk, gens, Z, A):
    """[com(Z_i) ; com(A_i)] per hidden layer (the public activation commitments)"""
    out = []
    for g, z, a in zip(gens, Z, A):
        out.append(zk.upload(np.concatenate([zk.commit(g, z).download(), zk.commit(g, a).download()])))
    return out


@pytest.mark.parametrize("dims,batch,derive", [([16, 32, 24, 8], 4, 0), ([16, 32, 24, 8], 4, 1), ([40, 70, 30], 20, 1), ([6, 5, 4], 3, 0)])
def test_activation_openings_inside_the_proof(zk, oracle, dims, batch, derive):
    import bigint_model as bm
    from mlp_util import enqueue_proof, setup_mlp_gpu, synth_mlp
    from zkdl_amd.capi import Transcript
    from zkdl_amd.proof import (act_open_lengths, act_statement_seed, fs_challenges, parse_transcript, prove_activation_openings, statement_digest, verify_mlp,
                                verify_mlp_with_activation_openings)
    model = synth_mlp(oracle, dims, batch, seed=51)
    st = setup_mlp_gpu(zk, model)
    coms = [fc.com() for fc in st["fcs"]]
    seed = statement_digest(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"])
    chal = fs_challenges(seed, dims, batch)
    act_coms = concat_coms(zk, st["gens"], st["Z"], st["A"])
    aseed = act_statement_seed(zk, seed, act_coms)         # the openings' challenges depend on the activation commitments too
    zk.set_flag("derive_rounds", derive)
    try:
        tr = Transcript(zk)
        enqueue_proof(st, chal, tr)
        base_counts = None
        prove_activation_openings(zk, dims, batch, st["gens"], st["Z"], st["A"], chal, aseed, tr)
        blob = tr.serialize()
        # the same through a sealed (pipelined) submission
        tr2 = Transcript(zk)
        enqueue_proof(st, chal, tr2)
        prove_activation_openings(zk, dims, batch, st["gens"], st["Z"], st["A"], chal, aseed, tr2)
        tr2.seal()
        assert tr2.serialize() == blob
    finally:
        zk.set_flag("derive_rounds", 0)
    args = (zk, dims, batch, st["gens"], coms, act_coms, st["X"], st["Yhat"])
    ok, msg, remaining = verify_mlp_with_activation_openings(*args, blob)
    assert ok, msg
    L = len(dims) - 1
    assert len(remaining) == 4 * (L - 1)              # per ReLU: mag_bin~, rem_bin~, (Z.sign)~, sign~ -- what the reference's ReLU leaves unbound
    fr, aff, inf = parse_transcript(blob)
    lens = act_open_lengths(dims, batch)
    nf_act = sum(a for a, _ in lens)
    # the opened value IS the random combination of the two tensors at the reduction's point (big-integer model on the downloaded tensors)
    from zkdl_amd.proof import act_open_challenges
    off = fr.shape[0] - nf_act
    for i, (alpha, fold) in enumerate(act_open_challenges(aseed, dims, batch)):
        nf = lens[i][0]
        t0 = bm.fr_from_mont(fr[off + nf - 1])
        r = [bm.fr_from_mont(x) for x in fold]
        zt = [bm.fr_from_mont(x) for x in st["Z"][i].download()]
        at = [bm.fr_from_mont(x) for x in st["A"][i].download()]
        want = ((1 - r[-1]) * bm.mle(zt, r[:-1]) + r[-1] * bm.mle(at, r[:-1])) % bm.R_MOD
        assert t0 == want, i
        off += nf
    # tampering: every Fr value of the openings' slices, a G1 coordinate of each, a swapped commitment, a false layer claim
    rng = np.random.default_rng(4)
    n_fr, n_g1 = fr.shape[0], aff.shape[0]
    for idx in range(n_fr - nf_act, n_fr):
        bad = bytearray(blob)
        bad[24 + 32 * idx + int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
        assert not verify_mlp_with_activation_openings(*args, bytes(bad))[0], idx
    ng_act = sum(b for _, b in lens)
    for idx in range(n_g1 - ng_act, n_g1, 3):
        bad = bytearray(blob)
        bad[24 + 32 * n_fr + 96 * idx + int(rng.integers(0, 96))] ^= 1 << int(rng.integers(0, 8))
        assert not verify_mlp_with_activation_openings(*args, bytes(bad))[0], idx
    if L >= 3:
        swapped = [act_coms[1], act_coms[0]] + act_coms[2:]
        if swapped[0].n == act_coms[0].n:
            assert not verify_mlp_with_activation_openings(zk, dims, batch, st["gens"], coms, swapped, st["X"], st["Yhat"], blob)[0]
    # a commitment to a different tensor: the opening no longer matches
    other = [zk.upload(np.concatenate([zk.commit(st["gens"][0], st["A"][0]).download(), zk.commit(st["gens"][0], st["Z"][0]).download()]))] + act_coms[1:]
    assert not verify_mlp_with_activation_openings(zk, dims, batch, st["gens"], coms, other, st["X"], st["Yhat"], blob)[0]


def test_activation_openings_at_full_size(zk):
    """18M-parameter MLP, batch 256: 7 claim reductions + 7 openings of 2^20-entry tensors recorded into the proof; verifies; cost printed."""
    import time
    import bench
    from zkdl_amd.capi import Transcript
    from zkdl_amd.proof import act_statement_seed, fs_challenges, prove_activation_openings, statement_digest, verify_mlp_with_activation_openings
    dims, batch = bench.WORKLOADS["mlp18m_b256"]
    wl = bench.Workload(zk, dims, batch, seed=21)
    coms = [fc.com() for fc in wl.fcs]
    seed = statement_digest(zk, dims, batch, wl.gens, coms, wl.X, wl.Yhat)
    chal = fs_challenges(seed, dims, batch)
    zk.sync()
    t0 = time.perf_counter()
    act_coms = concat_coms(zk, wl.gens[:-1], wl.Z, wl.A)
    zk.sync()
    t_commit = (time.perf_counter() - t0) * 1e3
    aseed = act_statement_seed(zk, seed, act_coms)
    zk.set_flag("derive_rounds", 1)
    try:
        tr = Transcript(zk)
        times = {}
        for with_act in (0, 1, 0, 1):
            zk.sync()
            t0 = time.perf_counter()
            for rep in range(3):
                wl.prove(chal, tr)
                if with_act:
                    prove_activation_openings(zk, dims, batch, wl.gens, wl.Z, wl.A, chal, aseed, tr)
                tr.flush()
            zk.sync()
            times[with_act] = (time.perf_counter() - t0) * 1e3 / 3
        blob = tr.serialize()
    finally:
        zk.set_flag("derive_rounds", 0)
    print("activation openings inside the proof, 18M MLP batch 256 (synchronous proofs): %.2f ms per proof without, %.2f ms with 7 reductions + 7 openings; "
          "14 activation commitments (2^18-2^19 entries each, incl. the host round trips of this test): %.1f ms" % (times[0], times[1], t_commit))
    ok, msg, remaining = verify_mlp_with_activation_openings(zk, dims, batch, wl.gens, coms, act_coms, wl.X, wl.Yhat, blob, chal=chal)
    assert ok, msg
    assert len(remaining) == 4 * 7
    bad = bytearray(blob)
    bad[-200] ^= 1
    assert not verify_mlp_with_activation_openings(zk, dims, batch, wl.gens, coms, act_coms, wl.X, wl.Yhat, bytes(bad), chal=chal)[0]
    tr.close()


@pytest.mark.parametrize("dims,batch", [([16, 32, 24, 8], 4), ([40, 70, 30], 20), ([600, 900, 700, 300], 64)])
def test_sound_layer_proofs_with_hashed_activation_openings(zk, oracle, dims, batch):
    """(i) the openings' challenges must not be known before com(Z_i), com(A_i) are fixed -- they are derived from a seed that absorbs the
    activation commitments; (ii) in the sound mode alpha depends on the claimed evaluations and the reduction's round challenges on its own round
    messages (zk_act_open_fs: the hash runs on the device); (iii) a sealed transcript that mixes sound-mode layer proofs (their openings sit in a
    chain that is NOT early) with activation openings (an early chain) must not start its G1 phase behind the early chains only."""
    from mlp_util import setup_mlp_gpu, synth_mlp
    from zkdl_amd.capi import Transcript
    from zkdl_amd.proof import (act_open_lengths, act_statement_seed, fs_round_mode_challenges, parse_transcript, prove_activation_openings,
                                prove_activation_openings_fs, prove_mlp_fs, statement_digest, verify_mlp_with_activation_openings)
    model = synth_mlp(oracle, dims, batch, seed=77)
    st = setup_mlp_gpu(zk, model)
    coms = [fc.com() for fc in st["fcs"]]
    seed = statement_digest(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"])
    act_coms = concat_coms(zk, st["gens"], st["Z"], st["A"])
    aseed = act_statement_seed(zk, seed, act_coms)
    args = (zk, dims, batch, st["gens"], coms, act_coms, st["X"], st["Yhat"])
    tr = Transcript(zk)
    prove_mlp_fs(st["fcs"], st["relus"], st["gens"], st["X"], st["Z"], st["A"], st["Yhat"], seed, tr)
    prove_activation_openings_fs(zk, dims, batch, st["gens"], st["Z"], st["A"], seed, aseed, tr)
    blob = tr.serialize()
    ok, msg, remaining = verify_mlp_with_activation_openings(*args, blob, sound=True)
    assert ok, msg
    assert len(remaining) == 4 * (len(dims) - 2)
    # the statement-derived schedule does not verify this proof, and other activation commitments change every challenge
    assert not verify_mlp_with_activation_openings(*args, blob, sound=False)[0]
    fr, aff, inf = parse_transcript(blob)
    lens = act_open_lengths(dims, batch)
    nf_act = sum(a for a, _ in lens)
    rng = np.random.default_rng(9)
    for idx in list(range(fr.shape[0] - nf_act, fr.shape[0], 2)) + [3, fr.shape[0] - nf_act - 2]:
        bad = bytearray(blob)
        bad[24 + 32 * idx + int(rng.integers(0, 31))] ^= 1 << int(rng.integers(0, 8))
        assert not verify_mlp_with_activation_openings(*args, bytes(bad), sound=True)[0], idx
    # adaptive tampering of a reduction's round message: c0 + d, c1 - 2d keeps that round's own check; the challenges derived from it change
    import bigint_model as bm
    off = fr.shape[0] - nf_act
    d = 12345
    c0, c1 = bm.limbs_to_int(fr[off]), bm.limbs_to_int(fr[off + 1])
    bad = bytearray(blob)
    bad[24 + 32 * off:24 + 32 * off + 32] = np.array(bm.int_to_limbs((c0 + d) % bm.R_MOD, 8), dtype=np.uint32).tobytes()
    bad[24 + 32 * (off + 1):24 + 32 * (off + 1) + 32] = np.array(bm.int_to_limbs((c1 - 2 * d) % bm.R_MOD, 8), dtype=np.uint32).tobytes()
    assert not verify_mlp_with_activation_openings(*args, bytes(bad), sound=True)[0]
    # a non-canonical encoding (c + r, same field element) of a proof value is refused outright
    small = [i for i in range(fr.shape[0]) if bm.limbs_to_int(fr[i]) + bm.R_MOD < (1 << 256)]
    if small:
        i0 = small[-1]
        bad = bytearray(blob)
        bad[24 + 32 * i0:24 + 32 * i0 + 32] = np.array(bm.int_to_limbs(bm.limbs_to_int(fr[i0]) + bm.R_MOD, 8), dtype=np.uint32).tobytes()
        assert not verify_mlp_with_activation_openings(*args, bytes(bad), sound=True)[0]
    # (iii) sound layer proofs + statement-scheduled activation openings, SEALED (pipelined submission, batched G1 phases) == the same recorded synchronously
    chal = None
    blobs = []
    for sealed in (False, True):
        zk.set_flag("g1_batch", 2 if sealed else 1)
        try:
            trs = [Transcript(zk) for _ in range(3)]
            for t in trs:
                prove_mlp_fs(st["fcs"], st["relus"], st["gens"], st["X"], st["Z"], st["A"], st["Yhat"], seed, t)
                if chal is None:
                    t.flush()
                    chal = fs_round_mode_challenges(seed, dims, batch, t.read()[0])
                    prove_mlp_fs(st["fcs"], st["relus"], st["gens"], st["X"], st["Z"], st["A"], st["Yhat"], seed, t)
                prove_activation_openings(zk, dims, batch, st["gens"], st["Z"], st["A"], chal, aseed, t)
                if sealed:
                    t.seal()
                else:
                    t.flush()
            blobs.append([t.serialize() for t in trs])
            for t in trs:
                t.close()
        finally:
            zk.set_flag("g1_batch", 1)
    assert blobs[0] == blobs[1] and blobs[0][0] == blobs[0][2]
    ok, msg, _ = verify_mlp_with_activation_openings(*args, blobs[1][1], chal=chal)
    assert ok, msg
    tr.close()


@pytest.mark.parametrize("dims,batch,tables", [([16, 32, 24, 8], 4, 0), ([40, 70, 30], 20, 8), ([600, 900, 700, 300], 64, 0)])
def test_activation_commitments_inside_the_pipeline(zk, oracle, dims, batch, tables):
    """the per-input activation commitments as part of the proving pipeline.  zk_act_commit defers the row MSMs of com(Z_i), com(A_i) into a
    transcript (digit tables when the generators have them); prove_with_activation_commitments_stream keeps two sealed stages per proof and three
    proofs in flight.  The points equal zk_commit's (affine-canonical), every streamed proof verifies against them, a proof made for other
    commitments does not, and all proofs of one statement are identical."""
    from mlp_util import setup_mlp_gpu, synth_mlp
    from zkdl_amd.capi import Transcript
    from zkdl_amd.proof import act_coms_from_points, prove_with_activation_commitments_stream, statement_digest, verify_mlp_with_activation_openings
    model = synth_mlp(oracle, dims, batch, seed=91)
    st = setup_mlp_gpu(zk, model)
    if tables:
        for g in st["gens"]:
            g.prepare_full(tables)
    coms = [fc.com() for fc in st["fcs"]]
    seed = statement_digest(zk, dims, batch, st["gens"], coms, st["X"], st["Yhat"])
    # the deferred commitment == the synchronous one
    tr = Transcript(zk)
    for i in range(len(dims) - 2):
        zk.act_commit(st["gens"][i], st["Z"][i], tr)
        zk.act_commit(st["gens"][i], st["A"][i], tr)
    aff, inf = tr.read_affine()
    want = concat_coms(zk, st["gens"], st["Z"], st["A"])
    og = 0
    for w in want:
        wa, wi = zk.g1_to_affine(w)
        assert np.array_equal(aff[og:og + w.n], wa) and np.array_equal(inf[og:og + w.n], wi)
        og += w.n
    assert og == aff.shape[0]
    tr.close()
    # with a stated bound on the scalars (activations are < 2^47 in magnitude) only their low windows are walked: same points ...
    tr = Transcript(zk)
    for i in range(len(dims) - 2):
        zk.act_commit(st["gens"][i], st["Z"][i], tr, 48)
        zk.act_commit(st["gens"][i], st["A"][i], tr, 48)
    aff48, inf48 = tr.read_affine()
    assert np.array_equal(aff48, aff) and np.array_equal(inf48, inf)
    zk.sync()
    # ... and a bound the tensor violates is caught on the device and reported by the next sync, not trusted
    from zkdl_amd.capi import ZkError
    tr.reset()
    zk.act_commit(st["gens"][0], st["Z"][0], tr, 3)
    tr.flush()
    with pytest.raises(ZkError) as err:
        zk.sync()
    assert err.value.code == 5            # ZK_ERR_RANGE
    zk.sync()                             # reported once
    tr.close()
    # the streamed prover
    got = list(prove_with_activation_commitments_stream(zk, dims, batch, st["gens"], st["fcs"], st["relus"], st["X"], st["Z"], st["A"], st["Yhat"], seed, 5))
    assert len(got) == 5
    args = lambda ac: (zk, dims, batch, st["gens"], coms, ac, st["X"], st["Yhat"])
    for a_k, i_k, blob in got:
        assert np.array_equal(a_k, aff) and np.array_equal(i_k, inf) and blob == got[0][2]
    act_coms = act_coms_from_points(zk, dims, batch, got[2][0], got[2][1])
    ok, msg, remaining = verify_mlp_with_activation_openings(*args(act_coms), got[2][2])
    assert ok, msg
    assert len(remaining) == 4 * (len(dims) - 2)
    ok2, msg2, _ = verify_mlp_with_activation_openings(*args(want), got[4][2])          # the same points, uploaded as zk_commit returned them
    assert ok2, msg2
    if len(dims) >= 4 and want[0].n == want[1].n:
        assert not verify_mlp_with_activation_openings(*args([want[1], want[0]] + want[2:]), got[0][2])[0]
    bad = bytearray(got[0][2])
    bad[-40] ^= 1
    assert not verify_mlp_with_activation_openings(*args(act_coms), bytes(bad))[0]
