"""The BENCHMARKED configuration at full size: bench.py proves the 18M-parameter MLP at batch 256 with
derived round commitments (`derive_rounds` = 1: T_r from the folding relation behind the MSM batch), four transcripts in flight and
sealed G1 phases on two G1 streams -- k_lo = 11, 2048 generators, 121 derived products per layer.  These tests pin exactly that
configuration:

  * every proof made that way equals, value for value, the proof made with the library defaults (every T_r an MSM) synchronously;
  * the verifier accepts a derived, pipelined proof made with hash-derived challenges;
  * the CPU oracle (the restatement of the reference algorithm, zkfc.cu:128-145 / zkrelu.cu:79-100 / commitment.cu:43-92), run on the
    same activations, weights, generators, commitments and challenges, returns the same Fr values bit for bit and the same G1 points
    (affine-canonical) for zkFC 7, zkFC 6 and zkReLU 6 at batch 256 -- through the very function bench.py's cpu_baseline leg uses
    for its `parity_checked` object.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


@pytest.fixture(scope="module")
def zk():
    from zkdl_amd.capi import MSM_PIPPENGER, Zk
    z = Zk(0)
    z.set_msm_mode(MSM_PIPPENGER)
    yield z
    z.close()


@pytest.fixture(scope="module")
def wl(zk):
    import bench
    dims, batch = bench.WORKLOADS["mlp18m_b256"]
    return bench.Workload(zk, dims, batch, seed=11, inflight=4)


def pipelined_proofs(zk, wl, steps, derive):
    """`steps` proofs enqueued back to back on the rotating transcripts and sealed (bench.py's timed loop); returns each proof's
    (Fr, affine G1, infinity flags, digest lanes) read after ALL of them have been submitted."""
    from zkdl_amd.dist import lanes_to_fr
    assert len(steps) <= len(wl.trs)
    zk.set_flag("derive_rounds", 1 if derive else 0)
    rows = zk.empty(2 * len(steps), 8)                       # 8 x uint64 per proof
    for j, st in enumerate(steps):
        tr = wl.trs[j]
        wl.prove(wl.challenges(st), tr)
        tr.seal(rows.ptr + 64 * j)
    out = []
    for j, st in enumerate(steps):
        fr, _ = wl.trs[j].read()
        aff, inf = wl.trs[j].read_affine()
        out.append((fr, aff, inf))
    zk.sync()
    lanes = rows.download().reshape(-1).view(np.uint64).reshape(len(steps), 8)
    zk.set_flag("derive_rounds", 0)
    return out, [lanes_to_fr([int(x) for x in row]) for row in lanes]


@pytest.mark.parametrize("g1_batch,early_g1", [(1, 1), (1, 0), (2, 1), (3, 1), (4, 0)])
def test_derived_pipelined_proofs_equal_the_default_synchronous_ones(zk, wl, g1_batch, early_g1):
    """g1_batch: the G1 phases of that many sealed proofs run as one batch (3 leaves the fourth proof waiting for a partner until it
    is read); early_g1: the G1 phase starts behind the openings' scalar preparation instead of behind the whole Fr phase."""
    from zkdl_amd.capi import Transcript
    steps = [0, 1, 2, 3]
    zk.set_flag("g1_batch", g1_batch)
    zk.set_flag("early_g1", early_g1)
    try:
        got, digests = pipelined_proofs(zk, wl, steps, derive=True)
    finally:
        zk.set_flag("g1_batch", 1)
        zk.set_flag("early_g1", 1)
    zk.set_flag("derive_rounds", 0)
    tr = Transcript(zk)
    for j, st in enumerate(steps):
        wl.prove(wl.challenges(st), tr)                      # library defaults: every T_r an MSM, flushed on the main stream
        fr, _ = tr.read()
        aff, inf = tr.read_affine()
        d = int.from_bytes(np.ascontiguousarray(tr.digest()).tobytes(), "little")
        assert eq(fr, got[j][0]), ("Fr transcript", st)
        assert eq(aff, got[j][1]) and eq(inf, got[j][2]), ("G1 transcript", st)
        assert d == digests[j], ("digest", st)
    tr.close()
    # k_lo = 11 layers exist in this model: 121 derived products each
    assert max(s["n_gens"] for s in wl.shapes) == 2048


def test_verifier_accepts_a_derived_pipelined_proof(zk, wl):
    from zkdl_amd.proof import fs_challenges, statement_digest, verify_mlp
    coms = [fc.com() for fc in wl.fcs]
    chal = fs_challenges(statement_digest(zk, wl.dims, wl.batch, wl.gens, coms, wl.X, wl.Yhat), wl.dims, wl.batch)
    zk.set_flag("derive_rounds", 1)
    # three other proofs in flight around it
    for j in range(3):
        wl.prove(wl.challenges(100 + j), wl.trs[j])
        wl.trs[j].seal()
    wl.prove(chal, wl.trs[3])
    wl.trs[3].seal()
    blob = wl.trs[3].serialize()
    zk.set_flag("derive_rounds", 0)
    ok, msg, _ = verify_mlp(zk, wl.dims, wl.batch, wl.gens, coms, wl.X, wl.Yhat, blob)
    assert ok, msg
    bad = bytearray(blob)
    bad[24 + 32 * 7 + 3] ^= 2
    assert not verify_mlp(zk, wl.dims, wl.batch, wl.gens, coms, wl.X, wl.Yhat, bytes(bad), chal=chal)[0]


def test_oracle_matches_the_benchmarked_configuration_at_batch_256(zk, wl):
    import bench
    got, _ = pipelined_proofs(zk, wl, [7, 8, 9, 10], derive=True)
    fr, aff, inf = got[2]
    base, checked = bench.cpu_baseline(zk, wl, "sample", (wl.challenges(9), fr, aff, inf))
    assert checked["ok"], checked
    assert checked["calls"] == ["zkFC7", "zkFC6", "zkReLU6"]
    assert checked["g1"] == 2 * (3 * 11 + 2) and checked["fr"] > 300
    assert base["kind"] == "port" and base["cores"] >= 1
    # and a corrupted GPU value is reported, not swallowed
    fr2 = fr.copy()
    fr2[0, 0] ^= 1
    _, bad = bench.cpu_baseline_check_only(zk, wl, [("fc", len(wl.shapes) - 1)], (wl.challenges(9), fr2, aff, inf))
    assert not bad["ok"] and bad["mismatch"] == ["zkFC7"]


def test_pending_g1_batches_survive_reset_destroy_and_flag_changes(zk):
    """g1_batch > 1 leaves a sealed proof's G1 phase with the context until partners arrive.  Whoever touches such a transcript (read, reset,
    destroy), changes the batch size, or synchronises the context must get the same values as an unbatched proof -- and nothing may dangle."""
    import bench
    from zkdl_amd.capi import Transcript
    wl = bench.Workload(zk, [64, 96, 48, 32], 16, seed=3, inflight=4)
    want = []
    tr = Transcript(zk)
    for st in range(3):
        wl.prove(wl.challenges(st), tr)
        want.append((tr.read()[0], tr.read_affine()))
    zk.set_flag("g1_batch", 3)
    try:
        # (a) one pending proof, read at once: runs alone
        wl.prove(wl.challenges(0), wl.trs[0]); wl.trs[0].seal()
        fr, _ = wl.trs[0].read()
        assert eq(fr, want[0][0]) and eq(wl.trs[0].read_affine()[0], want[0][1][0])
        # (b) two pending, the batch size changes under them: the flag change flushes them
        wl.prove(wl.challenges(1), wl.trs[1]); wl.trs[1].seal()
        wl.prove(wl.challenges(2), wl.trs[2]); wl.trs[2].seal()
        zk.set_flag("g1_batch", 2)
        assert eq(wl.trs[1].read()[0], want[1][0]) and eq(wl.trs[2].read_affine()[0], want[2][1][0])
        # (c) a pending transcript is destroyed, another one reset, then the context synchronised; a fresh proof afterwards is still right
        t_a, t_b = Transcript(zk), Transcript(zk)
        wl.prove(wl.challenges(0), t_a); t_a.seal()
        t_a.close()
        wl.prove(wl.challenges(1), t_b); t_b.seal()
        t_b.reset()
        zk.sync()
        wl.prove(wl.challenges(2), t_b); t_b.seal()
        zk.sync()
        assert eq(t_b.read()[0], want[2][0]) and eq(t_b.read_affine()[0], want[2][1][0])
        t_b.close()
    finally:
        zk.set_flag("g1_batch", 1)
    tr.close()
