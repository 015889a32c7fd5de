"""LeakyReLU with negative slope 2^-shift (a non-ReLU activation; an extension -- the reference has zkReLU only): the HIP
forward pass and prover against the CPU oracle's restatement of the same definition (oracle/zkdl_oracle.c: zo_leaky_forward /
zo_leaky_prove, themselves pinned by tests/test_oracle_leaky.py against Python integers), the verifier, and a float model."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

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


@pytest.mark.parametrize("n,shift", [(1, 3), (24, 0), (700, 3), (5000, 7), (4096, 31)])
def test_leaky_forward_and_proof_match_the_oracle(zk, oracle, n, shift):
    from util import rand_fr
    from zkdl_amd.capi import ReLU, Transcript
    import bigint_model as bm
    rng = np.random.default_rng(100 + n + shift)
    vals = [int(x) for x in rng.integers(-(1 << 44), 1 << 44, size=n)]
    if n >= 24:
        vals[:8] = [0, 1, -1, 32768, -32768, (1 << 47) - 1, -(1 << 47) + 1, -65536]
    X = np.array([bm.fr_to_mont(v % bm.R_MOD) for v in vals], dtype=np.uint32).reshape(-1, 8)
    A_o, sign_o, mag_o, rem_o, low_o, nsign_o = oracle.leaky_forward(X, shift)
    relu = ReLU(zk)
    relu.set_activation(1, shift)
    xd = zk.upload(X)
    a = relu.forward(xd)
    assert eq(a.download(), A_o)
    sign, mag, rem = relu.aux()
    assert eq(sign.download(), sign_o) and eq(mag.download(), mag_o) and eq(rem.download(), rem_o)
    k = oracle.ceil_log2(n)
    ch = [rand_fr(rng, m) for m in (k + 5, k + 5, k + 4, k + 4, k, k, k, k, k)]
    want = oracle.leaky_prove(X, A_o, sign_o, mag_o, rem_o, low_o, nsign_o, *ch)
    for packed in (1, 0):
        zk.set_flag("relu_packed", packed)
        tr = Transcript(zk)
        relu.prove_leaky(xd, a, ch, tr)
        fr, _ = tr.read()
        tr.close()
        assert eq(fr, want), ("relu_packed", packed)
    zk.set_flag("relu_packed", 1)
    ok, msg, claims = zk.verify_leaky(n, ch, fr)
    assert ok, msg
    # the extra sumcheck's claim is the negative branch's magnitude (low o nsign)~(u_lk)
    prod = [bm.fr_from_mont(x) * bm.fr_from_mont(y) % bm.R_MOD for x, y in zip(low_o, nsign_o)]
    assert bm.fr_from_mont(claims[5]) == bm.mle(prod, [bm.fr_from_mont(u) for u in ch[7]])
    if k > 0:                                                # (with a single element the Hadamard sumchecks have no rounds: nothing to constrain)
        bad = fr.copy()
        bad[-3, 0] ^= 1                                      # the last round of the extra sumcheck
        assert not zk.verify_leaky(n, ch, bad)[0]
    bad = fr.copy()
    bad[5, 2] ^= 4                                           # the magnitude-bit sumcheck
    assert not zk.verify_leaky(n, ch, bad)[0]
    # a ReLU layer refuses the leaky prover, and a leaky layer still proves as a ReLU-shaped transcript prefix
    plain = ReLU(zk)
    plain.forward(xd)
    tr = Transcript(zk)
    with pytest.raises(Exception):
        plain.prove_leaky(xd, a, ch, tr)
    tr.close()


@pytest.mark.parametrize("n,shift", [(1, 3), (700, 3), (5000, 0)])
def test_leaky_sound_mode_is_a_reference_transcript_and_verifies(zk, oracle, n, shift):
    """zk_leaky_prove_fs: every sumcheck round challenge (v_z, v_r, v_hp, v_lk) is a hash of the rounds before it.  The verifier's derivation
    (statement digest + the proof's own round messages) gives nine vectors with which zk_leaky_prove AND the CPU oracle reproduce the
    transcript bit for bit, and zk_verify_leaky accepts; a changed round message changes the later challenges and is rejected."""
    from zkdl_amd.capi import ReLU, Transcript, fs_derive, fs_round_challenges
    import bigint_model as bm
    rng = np.random.default_rng(300 + n + shift)
    vals = [int(x) for x in rng.integers(-(1 << 40), 1 << 40, size=n)]
    X = np.array([bm.fr_to_mont(v % bm.R_MOD) for v in vals], dtype=np.uint32).reshape(-1, 8)
    A_o, sign_o, mag_o, rem_o, low_o, nsign_o = oracle.leaky_forward(X, shift)
    relu = ReLU(zk)
    relu.set_activation(1, shift)
    xd = zk.upload(X)
    a = relu.forward(xd)
    seed, layer = bytes(range(32)), 3
    tr = Transcript(zk)
    relu.prove_leaky_fs(xd, a, seed, layer, tr)
    fr, _ = tr.read()
    tr.close()
    k = oracle.ceil_log2(n)
    o_rem = 3 * (k + 5) + 1 + 32
    o_hp = o_rem + 3 * (k + 4) + 1 + 16
    o_lk = o_hp + 3 * k + 2
    assert fr.shape[0] == o_lk + 3 * k + 2

    def challenges(fr):
        return [fs_derive(seed, layer, 1, 0, k + 5), fs_round_challenges(seed, layer, 1, 1, fr[:3 * (k + 5)]),
                fs_derive(seed, layer, 1, 2, k + 4), fs_round_challenges(seed, layer, 1, 2, fr[o_rem:o_rem + 3 * (k + 4)]),
                fs_derive(seed, layer, 1, 4, k), fs_derive(seed, layer, 1, 5, k), fs_round_challenges(seed, layer, 1, 3, fr[o_hp:o_hp + 3 * k]),
                fs_derive(seed, layer, 1, 7, k), fs_round_challenges(seed, layer, 1, 4, fr[o_lk:o_lk + 3 * k])]

    ch = challenges(fr)
    assert eq(fr, oracle.leaky_prove(X, A_o, sign_o, mag_o, rem_o, low_o, nsign_o, *ch))
    tr = Transcript(zk)
    relu.prove_leaky(xd, a, ch, tr)
    assert eq(tr.read()[0], fr)
    tr.close()
    ok, msg, _ = zk.verify_leaky(n, ch, fr)
    assert ok, msg
    if k > 1:
        bad = fr.copy()
        bad[o_lk + 1, 0] ^= 1                                # round 0 of the extra sumcheck: c1
        assert not zk.verify_leaky(n, challenges(bad), bad)[0]
    # a ReLU layer has no leaky transcript
    plain = ReLU(zk)
    plain.forward(xd)
    tr = Transcript(zk)
    with pytest.raises(Exception):
        plain.prove_leaky_fs(xd, a, seed, layer, tr)
    tr.close()


def test_mlp_with_leaky_and_identity_activations_matches_the_float_model(zk, oracle):
    """Linear(20, 24) -> LeakyReLU(1/8) -> Linear(24, 16) -> Identity -> Linear(16, 5): quantised inference through the C ABI equals the
    oracle's bit for bit and de-quantises to the float network (floor rounding on the negative branch: one unit of 2^-16 per layer)."""
    from zkdl_amd.capi import FC, ReLU
    from mlp_util import layer_shapes
    from util import rand_fr
    import bigint_model as bm
    rng = np.random.default_rng(77)
    dims, batch, shifts = [20, 24, 16, 5], 6, [3, 0]
    sh = layer_shapes(dims)
    W = [rng.uniform(-0.4, 0.4, size=(s["n_in"], s["n_out"])).astype(np.float32) for s in sh]
    x = rng.standard_normal((batch, dims[0])).astype(np.float32)
    gens = [zk.gens(zk.upload(oracle.g1_mul(np.repeat(oracle.generator(), s["n_gens"], axis=0), rand_fr(rng, s["n_gens"])))) for s in sh]
    fcs = [FC.from_float(zk, s["n_in"], s["n_out"], zk.upload_f32(W[i]), gens[i]) for i, s in enumerate(sh)]
    cur = zk.fr_mont(zk.fc_load_input(batch, dims[0], zk.upload_f32(x)))
    B_p = 8
    cur_o = oracle.fr_mont(oracle.float_to_fr(x, B_p, sh[0]["in_p"]))
    acts = []
    for i in range(2):
        z = fcs[i].forward(cur)
        act = ReLU(zk)
        act.set_activation(1, shifts[i])
        cur = act.forward(z)
        acts.append(act)
        z_o = oracle.fc_forward(cur_o, oracle.fr_mont(oracle.float_to_fr(W[i], sh[i]["in_p"], sh[i]["out_p"])), B_p, sh[i]["in_p"], sh[i]["out_p"])
        assert eq(z.download(), z_o)
        cur_o = oracle.leaky_forward(z_o, shifts[i])[0]
        assert eq(cur.download(), cur_o)
    y = zk.fr_unmont(fcs[2].forward(cur)).download()
    h = x.astype(np.float64) @ W[0]
    h = np.where(h >= 0, h, h / 8.0)
    h = h @ W[1]
    want = h @ W[2]
    got = np.array([[(lambda v: (v - bm.R_MOD if v > bm.R_MOD // 2 else v) / 2.0 ** 32)(bm.limbs_to_int(y[b * sh[2]["out_p"] + o])) for o in range(dims[3])] for b in range(batch)])
    assert np.abs(got - want).max() < 2e-3
