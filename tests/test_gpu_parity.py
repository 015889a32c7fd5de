"""GPU parity tests proper: every call goes through the C ABI (libzkdl_amd.so) and is compared with
the CPU oracle on the same seeded inputs.  Bar: bit-exact Fr; G1 bit-exact in Jacobian limbs for the
operator-level kernels that keep the reference's association order, affine-canonical otherwise."""
import numpy as np
import pytest

from util import rand_fp, rand_fr, rand_points, small_signed_fr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    from zkdl_amd.capi import Zk
    z = Zk(0)
    yield z
    z.close()


def eq(a, b):
    return np.array_equal(np.asarray(a).reshape(-1), np.asarray(b).reshape(-1))


# ------------------------------------------------------------------ Fr tensor
@pytest.mark.parametrize("n", [1, 63, 64, 257, 5000])
def test_fr_elementwise(zk, oracle, rng, n):
    a, b = rand_fr(rng, n), rand_fr(rng, n)
    # edge operands in the first slots: 0, 1, r-1, R mod r
    c = oracle.constants()
    edge = np.stack([np.zeros(8, np.uint32), np.eye(1, 8, 0, dtype=np.uint32)[0], oracle.fr_neg(np.eye(1, 8, 0, dtype=np.uint32))[0], c["fr_one"]])
    m = min(n, 4)
    a[:m] = edge[:m]
    b[:m] = edge[::-1][:m]
    da, db = zk.upload(a), zk.upload(b)
    assert eq(zk.fr_add(da, db).download(), oracle.fr_add(a, b))
    assert eq(zk.fr_sub(da, db).download(), oracle.fr_sub(a, b))
    assert eq(zk.fr_mul(da, db).download(), oracle.fr_mul(a, b))
    assert eq(zk.fr_neg(da).download(), oracle.fr_neg(a))
    assert eq(zk.fr_mont(da).download(), oracle.fr_mont(a))
    assert eq(zk.fr_unmont(da).download(), oracle.fr_unmont(a))
    x = rand_fr(rng, 1)
    assert eq(zk.fr_add_scalar(da, x).download(), oracle.fr_add_s(a, x))
    assert eq(zk.fr_sub_scalar(da, x).download(), oracle.fr_sub_s(a, x))
    assert eq(zk.fr_mul_scalar(da, x).download(), oracle.fr_mul_s(a, x))
    assert eq(zk.fr_sum(da), oracle.fr_sum(a))


def test_fr_empty_and_shape_errors(zk, rng):
    from zkdl_amd.capi import ShapeError
    e = zk.upload(np.zeros((0, 8), np.uint32))
    assert zk.fr_add(e, e).download().shape[0] == 0
    assert not zk.fr_sum(e).any()
    a, b = zk.upload(rand_fr(rng, 4)), zk.upload(rand_fr(rng, 5))
    with pytest.raises(ShapeError):
        zk.fr_add(a, b)
    with pytest.raises(ShapeError):
        zk.fr_me(a, rand_fr(rng, 3))          # size <= 2^(k-1)
    with pytest.raises(ShapeError):
        zk.fr_me(b, rand_fr(rng, 2))          # size > 2^k
    with pytest.raises(ShapeError):
        zk.fr_partial_me(a, rand_fr(rng, 2), 4)
    with pytest.raises(ShapeError):
        zk.fr_split(a, 4)


@pytest.mark.parametrize("n,k", [(1, 0), (2, 1), (5, 3), (64, 6), (1000, 10), (1025, 11), (70000, 17)])
def test_fr_me(zk, oracle, rng, n, k):
    t, u = rand_fr(rng, n), rand_fr(rng, k)
    assert eq(zk.fr_me(zk.upload(t), u), oracle.fr_me(t, u))


@pytest.mark.parametrize("n,window,k", [(12, 3, 2), (4096, 1, 5), (40, 8, 3), (2 * 16, 16, 1), (6, 2, 0), (300 * 32, 32, 9), (37 * 5, 5, 6), (3000, 1, 12)])
def test_fr_partial_me(zk, oracle, rng, n, window, k):
    t, u = rand_fr(rng, n), rand_fr(rng, k)
    got = zk.fr_partial_me(zk.upload(t), u, window).download()
    assert eq(got, oracle.fr_partial_me(t, u, window))


def test_fr_split(zk, oracle, rng):
    t = rand_fr(rng, 37)
    o0, o1 = zk.fr_split(zk.upload(t), 4)
    w0, w1 = oracle.fr_split(t, 4)
    assert eq(o0.download(), w0) and eq(o1.download(), w1)


def test_fr_random_is_canonical_and_reproducible(zk):
    zk.seed(123)
    a = zk.fr_random(1000).download()
    v = zk.random_vec(5)
    zk.seed(123)
    b = zk.fr_random(1000).download()
    assert eq(a, b) and eq(v, zk.random_vec(5))
    assert (a[:, 7] < 1944954707).all() and len({x.tobytes() for x in a}) == 1000
    r = zk.fr_random_int(256, 8).download()
    # values are uniform 8-bit ints minus 128, as field elements: either < 128 or >= r - 128
    small = (r[:, 1:] == 0).all(axis=1)
    assert (r[small, 0] < 128).all() and (~small).any()


# ------------------------------------------------------------------ sumchecks
@pytest.mark.parametrize("n,k", [(1, 0), (8, 3), (6, 3), (2048, 11), (1500, 11), (5000, 13), (100000, 17)])
def test_sumcheck_ip(zk, oracle, rng, n, k):
    a, b, u = rand_fr(rng, n), rand_fr(rng, n), rand_fr(rng, k)
    assert eq(zk.sumcheck_ip(zk.upload(a), zk.upload(b), u), oracle.ip_sumcheck(a, b, u))


# (131072, 17) and (100000, 17): big rounds whose workgroups take the factored-weight path of k_sc_round29 (one high weight per workgroup)
@pytest.mark.parametrize("n,k", [(1, 0), (8, 3), (7, 3), (4096, 12), (3000, 12), (9000, 14), (131072, 17), (100000, 17)])
def test_sumcheck_hadamard(zk, oracle, rng, n, k):
    a, b, u, v = rand_fr(rng, n), rand_fr(rng, n), rand_fr(rng, k), rand_fr(rng, k)
    assert eq(zk.sumcheck_hadamard(zk.upload(a), zk.upload(b), u, v), oracle.hp_sumcheck(a, b, u, v))


@pytest.mark.parametrize("n,k", [(1, 0), (16, 4), (12, 4), (8192, 13), (5001, 13), (131072, 17), (100000, 17)])
def test_sumcheck_binary(zk, oracle, rng, n, k):
    bits = rng.integers(0, 2, size=n)
    one = oracle.constants()["fr_one"]
    a = np.where(bits[:, None] == 1, one[None, :], 0).astype(np.uint32)
    u, v = rand_fr(rng, k), rand_fr(rng, k)
    assert eq(zk.sumcheck_binary(zk.upload(a), u, v), oracle.bin_sumcheck(a, u, v))
    a2 = rand_fr(rng, n)                      # non-boolean table: generic coefficients
    assert eq(zk.sumcheck_binary(zk.upload(a2), u, v), oracle.bin_sumcheck(a2, u, v))


def test_sumcheck_shape_errors(zk, rng):
    from zkdl_amd.capi import ShapeError
    a = zk.upload(rand_fr(rng, 4))
    with pytest.raises(ShapeError):
        zk.sumcheck_ip(a, a, rand_fr(rng, 3))
    with pytest.raises(ShapeError):
        zk.sumcheck_binary(zk.upload(rand_fr(rng, 9)), rand_fr(rng, 3), rand_fr(rng, 3))
    with pytest.raises(ShapeError):
        zk.sumcheck_hadamard(a, a, rand_fr(rng, 2), rand_fr(rng, 3))


# ------------------------------------------------------------------ G1 tensor (Jacobian-bitwise)
def test_g1_elementwise_bitwise(zk, oracle, rng):
    n = 70
    P, Q = rand_points(oracle, rng, n), rand_points(oracle, rng, n)
    inf = np.zeros(36, np.uint32)
    inf[12:24] = oracle.constants()["fp_one"]
    # special cases: inf + Q, P + inf, P + P, P + (-P)
    P[0] = inf
    Q[1] = inf
    Q[2] = P[2]
    Q[3] = oracle.g1_neg(P[3:4])[0]
    dP, dQ = zk.upload(P), zk.upload(Q)
    assert eq(zk.g1_add(dP, dQ).download(), oracle.g1_add(P, Q))
    assert eq(zk.g1_sub(dP, dQ).download(), oracle.g1_sub(P, Q))
    assert eq(zk.g1_neg(dP).download(), oracle.g1_neg(P))
    assert eq(zk.g1_add_bcast(dP, Q[5]).download(), oracle.g1_add_s(P, Q[5:6]))
    assert eq(zk.g1_sub_bcast(dP, Q[5]).download(), oracle.g1_sub_s(P, Q[5:6]))
    aff, flags = oracle.g1_to_affine(Q)
    aff[1] = aff[4]
    dA = zk.upload(aff)
    assert eq(zk.g1_madd(dP, dA).download(), oracle.g1_madd(P, aff))
    assert eq(zk.g1_msub(dP, dA).download(), oracle.g1_msub(P, aff))
    assert eq(zk.g1_madd_bcast(dP, aff[7]).download(), oracle.g1_madd_s(P, aff[7:8]))
    assert eq(zk.g1_msub_bcast(dP, aff[7]).download(), oracle.g1_msub_s(P, aff[7:8]))
    assert eq(zk.g1_from_affine(dA).download(), oracle.g1_from_affine(aff))
    assert eq(zk.g1_fill(9, P[8]).download(), np.repeat(P[8:9], 9, axis=0))
    got_aff, got_inf = zk.g1_to_affine(dP)
    want_aff, want_inf = oracle.g1_to_affine(P)
    assert eq(got_aff, want_aff) and eq(got_inf, want_inf) and got_inf[0] == 1


@pytest.mark.parametrize("n", [1, 2, 64, 65, 129, 300])
def test_g1_sum_reference_tree(zk, oracle, rng, n):
    base = rand_points(oracle, rng, min(n, 40))
    P = base[np.arange(n) % base.shape[0]]
    if n > 3:
        P[3] = oracle.g1_double(P[2:3])[0]
    assert eq(zk.g1_sum(zk.upload(P)), oracle.g1_sum(P))


def test_g1_scalar_mul_and_me(zk, oracle, rng):
    n = 24
    P = rand_points(oracle, rng, n)
    s = rand_fr(rng, n)
    s[0] = 0
    s[1] = 0; s[1, 0] = 1
    s[2] = oracle.constants()["fr_p"]                 # r*P = infinity
    s[3] = np.full(8, 0xFFFFFFFF, np.uint32)          # all 256 raw bits set (g1-tensor.cu:422-430 uses them all)
    dP, ds = zk.upload(P), zk.upload(s)
    assert eq(zk.g1_mul(dP, ds).download(), oracle.g1_mul(P, s))
    s2 = rand_fr(rng, 3 * 8)
    assert eq(zk.g1_mul_bcast(zk.upload(P[:8]), zk.upload(s2)).download(), oracle.g1_mul_bcast(P[:8], s2))
    for m, k in [(1, 0), (2, 1), (6, 3), (8, 3)]:
        u = rand_fr(rng, k)
        assert eq(zk.g1_me(zk.upload(P[:m]), u), oracle.g1_me(P[:m], u))


# ------------------------------------------------------------------ commitment (literal back end: bitwise)
def test_commit_me_open_open_literal(zk, oracle, rng):
    from zkdl_amd.capi import MSM_LITERAL, ShapeError
    zk.set_msm_mode(MSM_LITERAL)
    n, m = 8, 4
    gens = rand_points(oracle, rng, n)
    t = small_signed_fr(oracle, rng, n * m)
    dg, dt = zk.gens(zk.upload(gens)), zk.upload(t)
    com = zk.commit(dg, dt)
    want_com = oracle.commit(gens, t)
    assert eq(com.download(), want_com)
    u = rand_fr(rng, 3)
    tt = rand_fr(rng, n)
    ret, proof = zk.me_open(zk.upload(tt), dg, u)
    wret, wproof = oracle.me_open(tt, gens, u)
    assert eq(ret, wret) and eq(proof, wproof)
    u5 = rand_fr(rng, 5)
    ret, ce, proof = zk.open(dg, dt, com, u5)
    wret, wce, wproof = oracle.open(gens, t, want_com, u5)
    assert eq(ret, wret) and eq(ce, wce) and eq(proof, wproof)
    with pytest.raises(ShapeError):
        zk.commit(dg, zk.upload(rand_fr(rng, 7)))


# ------------------------------------------------------------------ layers
def test_quantiser_matmul_relu(zk, oracle, rng):
    from zkdl_amd.capi import FC, ReLU
    B, n_in, n_out = 5, 6, 7
    W = rng.uniform(-0.4, 0.4, size=(n_in, n_out)).astype(np.float32)
    W[0, :4] = [0.0, -0.0, 0.5 / 65536, -1.5 / 65536]
    X = rng.standard_normal((B, n_in)).astype(np.float32)
    Wq = zk.float_to_fr(zk.upload_f32(W), n_in, 8, n_out, 8)
    want_Wq = oracle.float_to_fr(W, 8, 8)
    assert eq(Wq.download(), want_Wq)
    gens = rand_points(oracle, rng, 8)
    zk.set_msm_mode(0)
    fc = FC.from_float(zk, n_in, n_out, zk.upload_f32(W), zk.gens(zk.upload(gens)))
    Wm = oracle.fr_mont(want_Wq)
    assert eq(fc.weights().download(), Wm) and (fc.n_in, fc.n_out, fc.com_size) == (8, 8, 8)
    assert eq(fc.com().download(), oracle.commit(gens, Wm))
    Xq = oracle.fr_mont(oracle.float_to_fr(X, 8, 8))
    Z = fc.forward(zk.upload(Xq))
    want_Z = oracle.fc_forward(Xq, Wm, 8, 8, 8)
    assert eq(Z.download(), want_Z)
    relu = ReLU(zk)
    A = relu.forward(Z)
    wA, wsign, wmag, wrem = oracle.relu_forward(want_Z)
    sign, mag, rem = relu.aux()
    assert eq(A.download(), wA) and eq(sign.download(), wsign) and eq(mag.download(), wmag) and eq(rem.download(), wrem)


def test_relu_window_edges_and_range_error(zk, oracle):
    from zkdl_amd.capi import ReLU, ZkError
    import bigint_model as bm
    vals = [0, 1, 32767, 32768, 32769, 65536, (1 << 47) - 1, -1, -32768, -32769, -(1 << 47)]
    X = np.stack([bm.fr_to_mont(v % bm.R_MOD) for v in vals])
    relu = ReLU(zk)
    A = relu.forward(zk.upload(X))
    wA, wsign, wmag, wrem = oracle.relu_forward(X)
    sign, mag, rem = relu.aux()
    assert eq(A.download(), wA) and eq(sign.download(), wsign) and eq(mag.download(), wmag) and eq(rem.download(), wrem)
    bad = np.stack([bm.fr_to_mont(1 << 47)])            # first value outside the window
    with pytest.raises(ZkError):
        relu.forward(zk.upload(bad))


@pytest.mark.parametrize("dims,batch", [([6, 5, 4], 3), ([16, 32, 8], 4), ([5, 3], 1)])
def test_mlp_prove_matches_oracle(zk, oracle, dims, batch):
    from mlp_util import prove_mlp_gpu, prove_mlp_oracle, synth_mlp
    zk.set_msm_mode(0)
    model = synth_mlp(oracle, dims, batch, seed=11)
    fr_o, g1_o = prove_mlp_oracle(oracle, model)
    fr_g, aff_g, inf_g, info = prove_mlp_gpu(zk, model)
    assert eq(info["Yhat"], model["_oracle"]["Yhat"])
    for a, b in zip(info["coms"], model["_oracle"]["coms"]):
        assert eq(a, b)
    assert eq(fr_g, fr_o)
    aff_o, inf_o = oracle.g1_to_affine(g1_o)
    assert eq(aff_g, aff_o) and eq(inf_g, inf_o)
    _, g1_g = info["tr"].read()
    assert eq(g1_g, g1_o)                      # literal back end keeps the reference's Jacobian limbs


# ------------------------------------------------------------------ Pippenger / flattened back end (affine-canonical)
def affine_of(oracle, pts):
    return oracle.g1_to_affine(np.asarray(pts).reshape(-1, 36))


@pytest.mark.parametrize("n,m,kind", [(8, 4, "small"), (64, 16, "small"), (32, 8, "full"), (16, 5, "mixed"), (128, 3, "small")])
def test_commit_pippenger_affine_parity(zk, oracle, rng, n, m, kind):
    from zkdl_amd.capi import MSM_PIPPENGER
    zk.set_msm_mode(MSM_PIPPENGER)
    gens = rand_points(oracle, rng, n)
    if kind == "small":
        t = small_signed_fr(oracle, rng, n * m)
    elif kind == "full":
        t = rand_fr(rng, n * m)
    else:
        t = small_signed_fr(oracle, rng, n * m)
        t[:n] = 0                                 # an all-zero (padded) row commits to infinity
        t[n:n + 3] = rand_fr(rng, 3)              # a few full-width scalars force the wide path
        t[2 * n] = oracle.fr_mont(oracle.fr_neg(np.eye(1, 8, 0, dtype=np.uint32)))[0]   # r - 1
    got = zk.commit(zk.gens(zk.upload(gens)), zk.upload(t)).download()
    want = oracle.commit(gens, t)
    ga, gi = affine_of(oracle, got)
    wa, wi = affine_of(oracle, want)
    assert eq(ga, wa) and eq(gi, wi)
    if kind == "mixed":
        assert gi[0] == 1
    # canonical output form: Z = ONE (or the canonical infinity)
    one = oracle.constants()["fp_one"]
    for p, inf in zip(got, gi):
        assert eq(p[24:36], np.zeros(12, np.uint32) if inf else one)


@pytest.mark.parametrize("n,k", [(2, 1), (8, 3), (64, 6)])
def test_me_open_flattened_affine_parity(zk, oracle, rng, n, k):
    from zkdl_amd.capi import MSM_PIPPENGER
    zk.set_msm_mode(MSM_PIPPENGER)
    gens = rand_points(oracle, rng, n)
    t, u = rand_fr(rng, n), rand_fr(rng, k)
    ret, proof = zk.me_open(zk.upload(t), zk.gens(zk.upload(gens)), u)
    wret, wproof = oracle.me_open(t, gens, u)
    assert eq(ret, wret)
    ga, gi = affine_of(oracle, proof)
    wa, wi = affine_of(oracle, wproof)
    assert eq(ga, wa) and eq(gi, wi)


def test_open_and_msm_batch_pippenger(zk, oracle, rng):
    from zkdl_amd.capi import MSM_PIPPENGER
    import bigint_model as bm
    zk.set_msm_mode(MSM_PIPPENGER)
    n, m = 16, 8
    gens = rand_points(oracle, rng, n)
    t = small_signed_fr(oracle, rng, n * m)
    t[3 * n:4 * n] = 0                            # infinite commitment point inside com(u)
    dg = zk.gens(zk.upload(gens))
    com = zk.commit(dg, zk.upload(t))
    want_com = oracle.commit(gens, t)
    u = rand_fr(rng, 7)
    ret, ce, proof = zk.open(dg, zk.upload(t), com, u)
    wret, wce, wproof = oracle.open(gens, t, want_com, u)
    assert eq(ret, wret)
    for g, w in ((ce, wce), (proof, wproof)):
        ga, gi = affine_of(oracle, g)
        wa, wi = affine_of(oracle, w)
        assert eq(ga, wa) and eq(gi, wi)
    # raw-scalar MSM probe against the big-integer model
    q = 3
    s = rand_fr(rng, q * n)
    s[0] = 0
    s[1] = oracle.constants()["fr_p"] - np.eye(1, 8, 0, dtype=np.uint32)[0]      # r - 1
    out = zk.msm_batch(dg, zk.upload(s)).download()
    gi_ = [bm.jac_to_affine_int(g) for g in gens]
    for row in range(q):
        want = None
        for j in range(n):
            want = bm.g1_add(want, bm.g1_mul(gi_[j], bm.limbs_to_int(s[row * n + j])))
        assert bm.jac_to_affine_int(out[row]) == want


@pytest.mark.parametrize("window", [0, 4, 9, 11, 12])
def test_msm_window_sweep_gives_same_points(zk, oracle, rng, window):
    from zkdl_amd.capi import MSM_PIPPENGER
    zk.set_msm_mode(MSM_PIPPENGER)
    zk.set_msm_window(window)
    try:
        n, m = 32, 4
        gens = rand_points(oracle, rng, n)
        t = rand_fr(rng, n * m)
        got = zk.commit(zk.gens(zk.upload(gens)), zk.upload(t)).download()
        ga, gi = affine_of(oracle, got)
        wa, wi = affine_of(oracle, oracle.commit(gens, t))
        assert eq(ga, wa) and eq(gi, wi)
    finally:
        zk.set_msm_window(0)


@pytest.mark.parametrize("n", [2048, 4096])
def test_full_width_msm_batch_is_the_same_group_element_at_every_window_width(zk, oracle, rng, n):
    """Full-width scalars over 2048 / 4096 generators, 40 MSMs in one batch: the width the library picks by generator count (11 / 12 bits) and the forced widths
    10 .. 13 give the same points.  At 11 and 12 bits the partial top window crowds a quarter to a half of each MSM's points into a few buckets -- the
    chains k_msm_combine's rotated block order spreads over the chip.  One row is checked against the literal (reference-shaped) MSM."""
    from zkdl_amd.capi import MSM_PIPPENGER
    q = 40
    base = zk.upload(rand_points(oracle, rng, 1))
    pts = zk.g1_mul(zk.g1_fill(n, base.download()[0]), zk.fr_random(n))
    gens = zk.gens(pts)
    sc = zk.fr_random(n * q)
    zk.set_msm_mode(MSM_PIPPENGER)
    ref = None
    try:
        for w in (0, 10, 11, 12, 13):
            zk.set_msm_window(w)
            aff, inf = zk.g1_to_affine(zk.msm_batch(gens, sc))
            key = (aff.tobytes(), inf.tobytes())
            ref = ref or key
            assert key == ref, w
        zk.set_msm_window(0)
        la, li = zk.g1_to_affine(zk.upload(zk.g1_sum(zk.g1_mul(pts, sc.view(0, n))).reshape(1, 36)))   # the reference's shape: elementwise products, then the tree sum
        pa, pi = zk.g1_to_affine(zk.msm_batch(gens, sc.view(0, n)))
        assert eq(la, pa) and eq(li, pi)
    finally:
        zk.set_msm_window(0)


def test_msm_window_outside_the_supported_range_is_refused(zk):
    """The counting sort holds one LDS counter per bucket: windows of 2 .. 13 bits.  Wider ones used to be clamped to 13 silently."""
    from zkdl_amd.capi import ZkError
    for w in (1, 14, 16, 17, 64):
        with pytest.raises(ZkError):
            zk.set_msm_window(w)
    zk.set_msm_window(13)
    zk.set_msm_window(0)


def test_derived_round_commitments_equal_the_msm_ones(zk, oracle):
    """`derive_rounds`: T_1 .. T_{k-1} of every opening from the folding relation (scalar multiplications of T_0, T0_j, T1_j
    behind the MSM batch) instead of as MSMs -- the same group elements, hence the same affine transcript and the oracle's."""
    from mlp_util import prove_mlp_gpu, prove_mlp_oracle, synth_mlp
    from zkdl_amd.capi import MSM_PIPPENGER
    zk.set_msm_mode(MSM_PIPPENGER)
    model = synth_mlp(oracle, [16, 32, 24, 8], 4, seed=23)
    fr_o, g1_o = prove_mlp_oracle(oracle, model)
    aff_o, inf_o = oracle.g1_to_affine(g1_o)
    gens = [zk.gens(zk.upload(g)) for g in model["gens"]]
    try:
        for flag in (1, 0):
            zk.set_flag("derive_rounds", flag)
            fr_g, aff_g, inf_g, _ = prove_mlp_gpu(zk, model, gens_dev=gens)
            assert eq(fr_g, fr_o) and eq(aff_g, aff_o) and eq(inf_g, inf_o), flag
    finally:
        zk.set_flag("derive_rounds", 0)


# batch 1 (BASELINE configs[1]): k_bs = 0, partial_me with no challenges is the identity (fr-tensor.cu:370-374, App. D-3)
@pytest.mark.parametrize("dims,batch", [([6, 5, 4], 3), ([16, 32, 8], 4), ([5, 3], 1), ([12, 20, 9, 7], 1)])
def test_mlp_prove_pippenger_matches_oracle(zk, oracle, dims, batch):
    from mlp_util import prove_mlp_gpu, prove_mlp_oracle, synth_mlp
    from zkdl_amd.capi import MSM_PIPPENGER
    zk.set_msm_mode(MSM_PIPPENGER)
    model = synth_mlp(oracle, dims, batch, seed=12)
    fr_o, g1_o = prove_mlp_oracle(oracle, model)
    fr_g, aff_g, inf_g, info = prove_mlp_gpu(zk, model)
    assert eq(fr_g, fr_o)
    aff_o, inf_o = oracle.g1_to_affine(g1_o)
    assert eq(aff_g, aff_o) and eq(inf_g, inf_o)
    for a, b in zip(info["coms"], model["_oracle"]["coms"]):
        ga, gi = affine_of(oracle, a)
        wa, wi = affine_of(oracle, b)
        assert eq(ga, wa) and eq(gi, wi)


# ------------------------------------------------------------------ packed ReLU proofs vs the one-Fr-per-bit path
# three forms of the same proof: byte histograms (default), three-round lookup tables, one field element per bit
@pytest.mark.parametrize("n,spread", [(1, 46), (2, 46), (5, 46), (64, 46), (300, 46), (300, 20), (4096, 46), (20000, 30)])
def test_relu_prove_packed_equals_unpacked_and_oracle(zk, oracle, rng, n, spread):
    from zkdl_amd.capi import ReLU, Transcript
    import bigint_model as bm
    vals = rng.integers(-(1 << spread), 1 << spread, size=n).tolist()     # spread 20/30: high bytes of the magnitude all zero
    vals[0] = (1 << 47) - 1
    if n > 4:
        vals[3] = 0
        vals[4] = -(1 << 47) + 1
    X = np.stack([bm.fr_to_mont(v % bm.R_MOD) for v in vals])
    relu = ReLU(zk)
    dX = zk.upload(X)
    A = relu.forward(dX)
    k = zk.ceil_log2(n)
    ch = [rand_fr(rng, m) for m in (k + 5, k + 5, k + 4, k + 4, k, k, k)]
    out = {}
    for form, (packed, by) in {"bytes": (1, 1), "tables": (1, 0), "bits": (0, 0)}.items():
        zk.set_flag("relu_packed", packed)
        zk.set_flag("relu_bytes", by)
        tr = Transcript(zk)
        relu.prove(dX, A, ch, tr)
        out[form], _ = tr.read()
    zk.set_flag("relu_packed", 1)
    zk.set_flag("relu_bytes", 1)
    assert eq(out["tables"], out["bits"])
    assert eq(out["bytes"], out["bits"])
    if n <= 300:
        wA, wsign, wmag, wrem = oracle.relu_forward(X)
        assert eq(out["bytes"], oracle.relu_prove(X, wA, wsign, wmag, wrem, *ch))


def test_lanes_and_deferred_msm_do_not_change_the_transcript(zk, oracle):
    """Worker lanes (independent layers on separate HIP streams) and the deferred MSM batch are pure
    scheduling: the transcript must be identical with lanes on and off, run after run."""
    from mlp_util import prove_mlp_gpu, synth_mlp
    from zkdl_amd.capi import MSM_PIPPENGER
    zk.set_msm_mode(MSM_PIPPENGER)
    model = synth_mlp(oracle, [16, 32, 24, 8], 4, seed=21)
    gens = [zk.gens(zk.upload(g)) for g in model["gens"]]
    runs = []
    for lanes in (1, 0, 1):
        zk.set_flag("lanes", lanes)
        fr, aff, inf, info = prove_mlp_gpu(zk, model, gens_dev=gens)
        runs.append((fr, aff, inf, info["tr"].digest()))
    zk.set_flag("lanes", 1)
    for r in runs[1:]:
        assert eq(r[0], runs[0][0]) and eq(r[1], runs[0][1]) and eq(r[2], runs[0][2]) and eq(r[3], runs[0][3])


def test_sealed_transcripts_pipeline_proofs_without_changing_them(zk, oracle):
    """zk_transcript_seal finishes a proof on the G1 stream while the next proof's Fr phase is already
    running on the worker lanes (two transcripts alternated, as bench.py does).  Pure scheduling: every
    proof must equal the one obtained by proving it alone and flushing on the main stream, and the
    device-side digest lanes must equal the host digest."""
    import numpy as np
    from mlp_util import enqueue_proof, setup_mlp_gpu, synth_mlp
    from zkdl_amd.capi import MSM_PIPPENGER, Transcript
    from zkdl_amd.dist import lanes_to_fr
    zk.set_msm_mode(MSM_PIPPENGER)
    model = synth_mlp(oracle, [16, 32, 24, 8], 4, seed=33)
    st = setup_mlp_gpu(zk, model)
    chals = [model["chal"]] + [model["draw_chal"]() for _ in range(4)]
    # reference runs: one at a time, synchronous flush
    want = []
    tr0 = Transcript(zk)
    for ch in chals:
        tr0.reset()
        enqueue_proof(st, ch, tr0)
        fr, _ = tr0.read()
        aff, inf = tr0.read_affine()
        want.append((fr, aff, inf, tr0.digest()))
    # pipelined: seal, move on to the next proof immediately, read each transcript two proofs later
    trs = [Transcript(zk), Transcript(zk)]
    lanes = [zk.empty(2, 8), zk.empty(2, 8)]   # 8 x uint64 digest lanes each (2 x 32 bytes)
    got = [None] * len(chals)

    def collect(i):
        tr = trs[i & 1]
        fr, _ = tr.read()
        aff, inf = tr.read_affine()
        l64 = lanes[i & 1].download().reshape(-1).view(np.uint64)
        got[i] = (fr, aff, inf, l64.copy())

    for i, ch in enumerate(chals):
        if i >= 2:
            collect(i - 2)
        trs[i & 1].reset()
        enqueue_proof(st, ch, trs[i & 1])
        trs[i & 1].seal(lanes[i & 1].ptr)
    collect(len(chals) - 2)
    collect(len(chals) - 1)
    for w, g in zip(want, got):
        assert eq(g[0], w[0]) and eq(g[1], w[1]) and eq(g[2], w[2])
        d = lanes_to_fr([int(x) for x in g[3]])
        assert d == int.from_bytes(np.ascontiguousarray(w[3]).tobytes(), "little")
    # a seal without lanes degrades to the synchronous path with the same values
    zk.set_flag("lanes", 0)
    trs[0].reset()
    enqueue_proof(st, chals[0], trs[0])
    trs[0].seal(lanes[0].ptr)
    fr, _ = trs[0].read()
    zk.set_flag("lanes", 1)
    assert eq(fr, want[0][0])


def test_paired_sealing_zips_two_proofs_without_changing_them(zk, oracle):
    """zk_transcript_seal_pair: two proofs recorded on two transcripts, their Fr chains merged into one set of batched launches, each
    G1 phase on its own stream.  Pure scheduling: both transcripts and both device digests equal the ones of proving each proof alone;
    pairs can follow each other back to back, with derived round commitments on or off, and degrade to two plain seals without lanes."""
    import numpy as np
    from mlp_util import enqueue_proof, setup_mlp_gpu, synth_mlp
    from zkdl_amd.capi import MSM_PIPPENGER, Transcript
    from zkdl_amd.dist import lanes_to_fr
    zk.set_msm_mode(MSM_PIPPENGER)
    model = synth_mlp(oracle, [16, 32, 24, 8], 4, seed=35)
    st = setup_mlp_gpu(zk, model)
    chals = [model["chal"]] + [model["draw_chal"]() for _ in range(5)]
    want = []
    tr0 = Transcript(zk)
    for ch in chals:
        tr0.reset()
        enqueue_proof(st, ch, tr0)
        fr, _ = tr0.read()
        aff, inf = tr0.read_affine()
        want.append((fr, aff, inf, tr0.digest()))
    trs = [Transcript(zk) for _ in range(4)]
    lanes = [zk.empty(2, 8) for _ in range(4)]

    def check(slot, i):
        fr, _ = trs[slot].read()
        aff, inf = trs[slot].read_affine()
        assert eq(fr, want[i][0]) and eq(aff, want[i][1]) and eq(inf, want[i][2])
        l64 = lanes[slot].download().reshape(-1).view(np.uint64)
        assert lanes_to_fr([int(x) for x in l64]) == int.from_bytes(np.ascontiguousarray(want[i][3]).tobytes(), "little")

    for derive in (0, 1):
        zk.set_flag("derive_rounds", derive)
        for base in (0, 2):                      # two pairs enqueued back to back before anything is read
            for j in (0, 1):
                trs[base + j].reset()
                enqueue_proof(st, chals[base + j], trs[base + j])
            trs[base].seal_pair(trs[base + 1], lanes[base].ptr, lanes[base + 1].ptr)
        for slot in range(4):
            check(slot, slot)
    zk.set_flag("derive_rounds", 0)
    # an empty partner, and no worker lanes: same values
    trs[0].reset()
    trs[1].reset()
    enqueue_proof(st, chals[4], trs[0])
    trs[0].seal_pair(trs[1], lanes[0].ptr, None)
    check(0, 4)
    zk.set_flag("lanes", 0)
    for j in (0, 1):
        trs[j].reset()
        enqueue_proof(st, chals[4 + j], trs[j])
    trs[0].seal_pair(trs[1], lanes[0].ptr, lanes[1].ptr)
    zk.set_flag("lanes", 1)
    check(0, 4)
    check(1, 5)
    # a transcript cannot be its own partner; the failed call leaves both usable
    from zkdl_amd.capi import ZkError
    with pytest.raises(ZkError):
        trs[0].seal_pair(trs[0], lanes[0].ptr, lanes[0].ptr)
    trs[0].reset()
    enqueue_proof(st, chals[2], trs[0])
    trs[0].seal(lanes[0].ptr)
    check(0, 2)


def test_recorded_batched_layers_equal_immediate_execution(zk, oracle):
    """zk_fc_prove / zk_relu_prove only RECORD their launches; the flush zips the same kernel of different
    layers into one launch (blockIdx.z = job) and spreads independent pieces over streams.  Pure scheduling:
    the transcript must equal the one produced with batching off (every launch enqueued immediately), also
    when other entry points are called between the prove calls (they settle the recorded work first)."""
    from mlp_util import enqueue_proof, setup_mlp_gpu, synth_mlp
    from zkdl_amd.capi import MSM_PIPPENGER, Transcript
    zk.set_msm_mode(MSM_PIPPENGER)
    model = synth_mlp(oracle, [24, 40, 16, 56, 8], 6, seed=77)   # ragged widths: chains of different lengths
    st = setup_mlp_gpu(zk, model)
    runs = {}
    for batch, lanes in ((0, 0), (1, 0), (1, 1), (0, 1)):
        zk.set_flag("batch_layers", batch)
        zk.set_flag("lanes", lanes)
        tr = Transcript(zk)
        enqueue_proof(st, model["chal"], tr)
        fr, _ = tr.read()
        aff, inf = tr.read_affine()
        runs[(batch, lanes)] = (fr, aff, inf)
    zk.set_flag("batch_layers", 1)
    zk.set_flag("lanes", 1)
    ref = runs[(0, 0)]
    for key, r in runs.items():
        assert eq(r[0], ref[0]) and eq(r[1], ref[1]) and eq(r[2], ref[2]), key
    # interleave unrelated entry points between the prove calls
    tr = Transcript(zk)
    scratch = zk.fr_random(100)
    L = st["L"]
    for kind, i, ch in model["chal"]:
        if kind == "fc":
            x_in = st["A"][i - 1] if i > 0 else st["X"]
            z_out = st["Yhat"] if i == L - 1 else st["Z"][i]
            st["fcs"][i].prove(x_in, z_out, st["gens"][i], ch[0], ch[1], ch[2], tr)
        else:
            st["relus"][i].prove(st["Z"][i], st["A"][i], ch, tr)
        scratch = zk.fr_add(scratch, scratch)            # settles what has been recorded so far
    fr, _ = tr.read()
    assert eq(fr, ref[0])


def test_int16_weight_contractions_equal_the_fr_path(zk, oracle):
    """Quantised layers keep an int16 copy of W; the prover's two contractions of W (partial_me(u_out, 1) and
    the row fold of Commitment::open) then run as 8x1-limb products accumulated unreduced.  Same field
    elements: the transcript must not depend on the flag, and must still equal the oracle's."""
    from mlp_util import enqueue_proof, prove_mlp_oracle, setup_mlp_gpu, synth_mlp
    from zkdl_amd.capi import MSM_PIPPENGER, Transcript
    zk.set_msm_mode(MSM_PIPPENGER)
    model = synth_mlp(oracle, [48, 96, 40, 24], 5, seed=91)
    out = {}
    for flag in (1, 0):
        zk.set_flag("weights_q16", flag)
        st = setup_mlp_gpu(zk, model)
        tr = Transcript(zk)
        enqueue_proof(st, model["chal"], tr)
        fr, _ = tr.read()
        aff, inf = tr.read_affine()
        out[flag] = (fr, aff, inf)
    zk.set_flag("weights_q16", 1)
    assert eq(out[1][0], out[0][0]) and eq(out[1][1], out[0][1]) and eq(out[1][2], out[0][2])
    want_fr = prove_mlp_oracle(oracle, model)[0]
    assert eq(out[1][0], want_fr)


def test_transcript_serialisation_round_trip(zk, oracle):
    """zk_transcript_serialize: header, Fr entries, affine G1 entries and infinity flags equal what read() /
    read_affine() return, and the bytes do not depend on how the proof was scheduled."""
    import struct
    from mlp_util import enqueue_proof, setup_mlp_gpu, synth_mlp
    from zkdl_amd.capi import MSM_PIPPENGER, Transcript
    zk.set_msm_mode(MSM_PIPPENGER)
    model = synth_mlp(oracle, [16, 32, 8], 4, seed=123)
    st = setup_mlp_gpu(zk, model)
    blobs = []
    for batch in (1, 0):
        zk.set_flag("batch_layers", batch)
        tr = Transcript(zk)
        enqueue_proof(st, model["chal"], tr)
        blob = tr.serialize()
        fr, _ = tr.read()
        aff, inf = tr.read_affine()
        assert blob[:8] == b"ZKDLTR01"
        n_fr, n_g1 = struct.unpack("<QQ", blob[8:24])
        assert (n_fr, n_g1) == (fr.shape[0], aff.shape[0])
        body = np.frombuffer(blob, dtype=np.uint8, offset=24)
        assert eq(body[:32 * n_fr].view(np.uint32).reshape(-1, 8), fr)
        g = body[32 * n_fr:32 * n_fr + 96 * n_g1].view(np.uint32).reshape(-1, 24)
        assert eq(g, np.asarray(aff).reshape(-1, 24))
        assert eq(body[32 * n_fr + 96 * n_g1:], np.asarray(inf).reshape(-1))
        assert len(blob) == 24 + 32 * n_fr + 97 * n_g1
        blobs.append(blob)
    zk.set_flag("batch_layers", 1)
    assert blobs[0] == blobs[1]


# ---- larger oracle sizes on the G1 paths: g1_me at 256 / 300 points, the flattened opening at 512 generators
# (bucket method and digit tables), zk_open with 256 generators x 128 rows
@pytest.mark.parametrize("n", [256, 300])
def test_g1_me_against_the_oracle_at_hundreds_of_points(zk, oracle, rng, n):
    pts = rand_points(oracle, rng, n)
    k = (n - 1).bit_length()                                  # n in (2^(k-1), 2^k]  (g1-tensor.cu:489)
    u = rand_fr(rng, k)
    got = zk.g1_me(zk.upload(pts), u)
    want = oracle.g1_me(pts, u)
    assert eq(got.reshape(1, 36), np.asarray(want).reshape(1, 36))          # literal chain: Jacobian limbs equal


@pytest.mark.parametrize("tables", [0, 6])
def test_me_open_and_open_against_the_oracle_at_512_generators(zk, oracle, rng, tables):
    from zkdl_amd.capi import MSM_PIPPENGER
    zk.set_msm_mode(MSM_PIPPENGER)
    n, k = 512, 9
    gens = rand_points(oracle, rng, n)
    dg = zk.gens(zk.upload(gens))
    if tables:
        dg.prepare_full(tables)
    t, u = rand_fr(rng, n), rand_fr(rng, k)
    ret, proof = zk.me_open(zk.upload(t), dg, u)
    wret, wproof = oracle.me_open(t, gens, u)
    assert eq(ret, wret)
    ga, gi = affine_of(oracle, proof)
    wa, wi = affine_of(oracle, wproof)
    assert eq(ga, wa) and eq(gi, wi)
    # Commitment::open over 64 rows of these generators
    m = 64
    w = small_signed_fr(oracle, rng, n * m)
    com = zk.commit(dg, zk.upload(w))
    want_com = oracle.commit(gens, w)
    u2 = rand_fr(rng, k + 6)
    ret, ce, proof = zk.open(dg, zk.upload(w), com, u2)
    wret, wce, wproof = oracle.open(gens, w, want_com, u2)
    assert eq(ret, wret)
    for g, ww in ((ce, wce), (proof, wproof)):
        ga, gi = affine_of(oracle, g)
        wa, wi = affine_of(oracle, ww)
        assert eq(ga, wa) and eq(gi, wi)
