"""Protocol-level known-answer tests for the CPU oracle: the reference ships no
tests, so the restated kernels/drivers are pinned by reference-independent invariants evaluated with
the big-integer model (tests/bigint_model.py) acting as a mini-verifier."""
import numpy as np
import pytest

import bigint_model as bm
from util import rand_fr, rand_points, small_signed_fr

R = bm.R_MOD


def ints(a):
    """Montgomery limbs -> field values."""
    return [bm.fr_from_mont(x) for x in np.asarray(a).reshape(-1, 8)]


def raw(a):
    return [bm.limbs_to_int(x) for x in np.asarray(a).reshape(-1, 8)]


def to_mont(vals):
    return np.stack([bm.fr_to_mont(v) for v in vals])


@pytest.mark.parametrize("n,k", [(1, 0), (2, 1), (5, 3), (8, 3), (13, 4), (64, 6)])
def test_fr_me_is_multilinear_extension(oracle, rng, n, k):
    t, u = rand_fr(rng, n), rand_fr(rng, k)
    assert bm.fr_from_mont(oracle.fr_me(t, u)) == bm.mle(ints(t), ints(u))


def test_fr_me_rejects_bad_sizes(oracle, rng):
    with pytest.raises(ValueError):
        oracle.fr_me(rand_fr(rng, 4), rand_fr(rng, 3))      # size <= 2^(k-1)
    with pytest.raises(ValueError):
        oracle.fr_me(rand_fr(rng, 9), rand_fr(rng, 3))      # size > 2^k


@pytest.mark.parametrize("rows,window,k", [(4, 3, 2), (8, 1, 3), (5, 4, 3), (2, 16, 1), (3, 2, 0)])
def test_partial_me_folds_window_index(oracle, rng, rows, window, k):
    t, u = rand_fr(rng, rows * window), rand_fr(rng, k)
    got = oracle.fr_partial_me(t, u, window)
    ti, w = ints(t), bm.eq_weights(ints(u))
    if k == 0:
        assert np.array_equal(got, t)
        return
    assert got.shape[0] == window
    for j in range(window):
        want = sum(w[r] * ti[r * window + j] for r in range(rows)) % R
        assert bm.fr_from_mont(got[j]) == want


def test_fr_sum_and_split(oracle, rng):
    t = rand_fr(rng, 37)
    assert bm.limbs_to_int(oracle.fr_sum(t)) == sum(raw(t)) % R
    a0, a1 = oracle.fr_split(t, 4)
    assert a0.shape[0] == 19
    for g in range(19):
        w, i = divmod(g, 4)
        g0, g1 = 2 * w * 4 + i, (2 * w + 1) * 4 + i
        assert bm.limbs_to_int(a0[g]) == (bm.limbs_to_int(t[g0]) if g0 < 37 else 0)
        assert bm.limbs_to_int(a1[g]) == (bm.limbs_to_int(t[g1]) if g1 < 37 else 0)
    with pytest.raises(ValueError):
        oracle.fr_split(t, 37)


def poly(c, x):
    return (c[0] + c[1] * x + c[2] * x * x) % R


@pytest.mark.parametrize("n,k", [(8, 3), (6, 3), (32, 5), (1, 0)])
def test_inner_product_sumcheck_invariants(oracle, rng, n, k):
    a, b, u = rand_fr(rng, n), rand_fr(rng, n), rand_fr(rng, k)
    pr = ints(oracle.ip_sumcheck(a, b, u))
    ai, bi, ui = ints(a), ints(b), ints(u)
    claim = sum(x * y for x, y in zip(ai, bi)) % R
    for r in range(k):
        c = pr[3 * r:3 * r + 3]
        assert (poly(c, 0) + poly(c, 1)) % R == claim
        claim = poly(c, ui[r])
    fa, fb = pr[3 * k], pr[3 * k + 1]
    assert fa * fb % R == claim
    assert fa == bm.mle(ai, ui) and fb == bm.mle(bi, ui)


@pytest.mark.parametrize("n,k", [(8, 3), (7, 3), (16, 4)])
def test_hadamard_sumcheck_invariants(oracle, rng, n, k):
    a, b, u, v = rand_fr(rng, n), rand_fr(rng, n), rand_fr(rng, k), rand_fr(rng, k)
    pr = ints(oracle.hp_sumcheck(a, b, u, v))
    ai, bi, ui, vi = ints(a), ints(b), ints(u), ints(v)
    claim = bm.mle([x * y % R for x, y in zip(ai, bi)], ui)
    for r in range(k):
        c = pr[3 * r:3 * r + 3]
        assert ((1 - ui[r]) * poly(c, 0) + ui[r] * poly(c, 1)) % R == claim
        claim = poly(c, vi[r])
    fa, fb = pr[3 * k], pr[3 * k + 1]
    assert fa * fb % R == claim
    assert fa == bm.mle(ai, vi) and fb == bm.mle(bi, vi)


@pytest.mark.parametrize("n,k", [(16, 4), (12, 4)])
def test_binary_sumcheck_invariants(oracle, rng, n, k):
    bits = rng.integers(0, 2, size=n)
    a = to_mont(bits.tolist())
    u, v = rand_fr(rng, k), rand_fr(rng, k)
    pr = ints(oracle.bin_sumcheck(a, u, v))
    ui, vi = ints(u), ints(v)
    claim = 0                                             # (a^2 - a)~(u) vanishes on boolean a
    for r in range(k):
        c = pr[3 * r:3 * r + 3]
        if r == 0:
            assert c[0] == 0
        assert ((1 - ui[r]) * poly(c, 0) + ui[r] * poly(c, 1)) % R == claim
        claim = poly(c, vi[r])
    fa = pr[3 * k]
    assert (fa * fa - fa) % R == claim
    assert fa == bm.mle(bits.tolist(), vi)
    # a non-boolean table makes the claimed value non-zero and the recurrence still hold
    a2 = rand_fr(rng, n)
    pr = ints(oracle.bin_sumcheck(a2, u, v))
    ai = ints(a2)
    claim = bm.mle([(x * x - x) % R for x in ai], ui)
    for r in range(k):
        c = pr[3 * r:3 * r + 3]
        assert ((1 - ui[r]) * poly(c, 0) + ui[r] * poly(c, 1)) % R == claim
        claim = poly(c, vi[r])


def test_sumcheck_shape_errors(oracle, rng):
    with pytest.raises(ValueError):
        oracle.ip_sumcheck(rand_fr(rng, 4), rand_fr(rng, 5), rand_fr(rng, 3))
    with pytest.raises(ValueError):
        oracle.ip_sumcheck(rand_fr(rng, 4), rand_fr(rng, 4), rand_fr(rng, 3))
    with pytest.raises(ValueError):
        oracle.bin_sumcheck(rand_fr(rng, 9), rand_fr(rng, 3), rand_fr(rng, 3))
    with pytest.raises(ValueError):
        oracle.hp_sumcheck(rand_fr(rng, 8), rand_fr(rng, 8), rand_fr(rng, 3), rand_fr(rng, 2))


def test_commit_is_row_msm_and_linear(oracle, rng):
    n, m = 4, 3
    gens = rand_points(oracle, rng, n)
    ta = small_signed_fr(oracle, rng, n * m)
    tb = rand_fr(rng, n * m)
    ca, cb, cab = oracle.commit(gens, ta), oracle.commit(gens, tb), oracle.commit(gens, oracle.fr_add(ta, tb))
    gi = [bm.jac_to_affine_int(g) for g in gens]
    vals = ints(ta)                                        # commit un-Montgomery's its scalars
    for row in range(m):
        want = None
        for j in range(n):
            want = bm.g1_add(want, bm.g1_mul(gi[j], vals[row * n + j]))
        assert bm.jac_to_affine_int(ca[row]) == want
        lhs = bm.g1_add(bm.jac_to_affine_int(ca[row]), bm.jac_to_affine_int(cb[row]))
        assert lhs == bm.jac_to_affine_int(cab[row])
    with pytest.raises(ValueError):
        oracle.commit(gens, rand_fr(rng, 7))


def test_commit_as_launched_is_the_documented_stripe(oracle, rng):
    # only out[64b] is written and it mixes a diagonal stripe of 64 rows.
    n, m = 64, 64
    gens = rand_points(oracle, rng, n)
    t = small_signed_fr(oracle, rng, n * m, bits=4)
    out, written = oracle.commit_as_launched(gens, t)
    assert written.tolist() == [1] + [0] * 63
    gi = [bm.jac_to_affine_int(g) for g in gens]
    vals = ints(t)
    want = None
    for k in range(64):
        want = bm.g1_add(want, bm.g1_mul(gi[k], vals[k * n + k]))
    assert bm.jac_to_affine_int(out[0]) == want


def test_me_open_round_relation(oracle, rng):
    n, k = 8, 3
    gens = rand_points(oracle, rng, n)
    t, u = rand_fr(rng, n), rand_fr(rng, k)
    ret, proof = oracle.me_open(t, gens, u)
    gi = [bm.jac_to_affine_int(g) for g in gens]
    s = raw(t)                                   # scalar-muls use the RAW (Montgomery-form) limbs
    uh = ints(u)                                 # unmont(u)
    rinv = pow(bm.RR, -1, R)

    def msm(gs, ss):
        acc = None
        for g, x in zip(gs, ss):
            acc = bm.g1_add(acc, bm.g1_mul(g, x))
        return acc

    for r in range(k):
        T, T0, T1 = [bm.jac_to_affine_int(p) for p in proof[3 * r:3 * r + 3]]
        g0, g1, s0, s1 = gi[0::2], gi[1::2], s[0::2], s[1::2]
        assert T == msm(gi, s)
        assert T0 == msm(g1, s0) and T1 == msm(g0, s1)
        # fold exactly as me_open_step does: s' = s0 + u (x) (s1 - s0) with (x) the Montgomery product
        um = bm.limbs_to_int(u[r])
        s = [(a + um * ((b - a) % R) * rinv) % R for a, b in zip(s0, s1)]
        gi = [bm.g1_add(b, bm.g1_mul(bm.g1_add(a, bm.g1_neg(b)), uh[r])) for a, b in zip(g0, g1)]
        lhs = msm(gi, s)
        rhs = bm.g1_add(bm.g1_add(bm.g1_mul(T, uh[r] * (1 - uh[r]) % R), bm.g1_mul(T0, (1 - uh[r]) ** 2 % R)),
                        bm.g1_mul(T1, uh[r] ** 2 % R))
        assert lhs == rhs
    assert bm.jac_to_affine_int(proof[3 * k]) == gi[0]
    assert bm.limbs_to_int(ret) == s[0]
    with pytest.raises(ValueError):
        oracle.me_open(rand_fr(rng, 4), gens, u)


def test_open_splits_challenges_and_folds_rows(oracle, rng):
    n, m = 4, 4                                    # 16 weights, 4 generators, 4 commitment rows
    gens = rand_points(oracle, rng, n)
    t = small_signed_fr(oracle, rng, n * m)
    com = oracle.commit(gens, t)
    u = rand_fr(rng, 4)
    ret, com_eval, proof = oracle.open(gens, t, com, u)
    folded = oracle.fr_partial_me(t, u[2:], n)      # last ceilLog2(m) challenges fold the row index
    ret2, proof2 = oracle.me_open(folded, gens, u[:2])
    assert np.array_equal(ret, ret2) and np.array_equal(proof, proof2)
    assert np.array_equal(com_eval, oracle.g1_me(com, u[2:]))
    # binding identity: com(u_hi) commits to the row-folded tensor
    fv = ints(folded)
    want = None
    for j in range(n):
        want = bm.g1_add(want, bm.g1_mul(bm.jac_to_affine_int(gens[j]), fv[j]))
    assert bm.jac_to_affine_int(com_eval) == want


def test_float_to_fr_quantiser(oracle):
    fs = np.array([[0.0, -0.0, 1.0, -1.0, 0.5 / 65536, -0.5 / 65536, 1.5 / 65536, -2.5 / 65536, 0.0357, -4.63, 65535.99]],
                  dtype=np.float32)
    out = oracle.float_to_fr(fs, 2, 16)
    assert out.shape[0] == 32
    for j, x in enumerate(fs[0]):
        scaled = float(np.float32(x) * np.float32(65536.0))
        mag = int(np.floor(abs(scaled) + 0.5))             # half away from zero
        want = (R - mag) % R if (scaled < 0 or (scaled == 0 and np.signbit(x))) and mag else mag
        assert bm.limbs_to_int(out[j]) == want, (j, x)
    assert not out[11:].any()                               # zero padding (columns and the extra row)


def test_fc_forward_is_field_matmul(oracle, rng):
    B, n_in, n_out = 3, 4, 5
    X, W = small_signed_fr(oracle, rng, B * n_in, bits=20), small_signed_fr(oracle, rng, n_in * n_out)
    Z = ints(oracle.fc_forward(X, W, B, n_in, n_out))
    xi, wi = ints(X), ints(W)
    for b in range(B):
        for o in range(n_out):
            assert Z[b * n_out + o] == sum(xi[b * n_in + k] * wi[k * n_out + o] for k in range(n_in)) % R


def test_relu_forward_semantics(oracle, rng):
    vals = [0, 1, 32767, 32768, 32769, 65536, 98304, 98303, (1 << 47) - 1, -1, -32768, -32769, -65536, -(1 << 47),
            123456789, -987654321] + rng.integers(-(1 << 40), 1 << 40, size=16).tolist()
    X = to_mont([v % R for v in vals])
    Z, sign, mag, rem = oracle.relu_forward(X)
    zi, si = ints(Z), ints(sign)
    mb, rb = ints(mag), ints(rem)
    for g, v in enumerate(vals):
        assert all(b in (0, 1) for b in mb[32 * g:32 * g + 32] + rb[16 * g:16 * g + 16])
        q = (v + (1 << 15)) >> 16                     # round-to-nearest /2^16, ties up
        if v >= 0:
            assert si[g] == 1 and zi[g] == q
            m = sum(b << i for i, b in enumerate(mb[32 * g:32 * g + 32]))
            assert m == q
        else:
            assert si[g] == 0 and zi[g] == 0
        # recomposition: mag = 2^16 * mag_rescaled + rem  with rem the signed low-16 value
        mag_full = v if v >= 0 else (1 << 47) + v
        rem_mag = sum(b << i for i, b in enumerate(rb[16 * g:16 * g + 15]))
        rem_val = rem_mag - (1 << 15) * rb[16 * g + 15]
        m = sum(b << i for i, b in enumerate(mb[32 * g:32 * g + 32]))
        assert (m << 16) + rem_val == mag_full


def test_fc_and_relu_prove_transcripts(oracle, rng):
    """One FC + ReLU pair on a tiny layer: transcript pieces equal the pieces computed one by one."""
    B, n_in, n_out = 2, 4, 8
    W = small_signed_fr(oracle, rng, n_in * n_out)
    X = small_signed_fr(oracle, rng, B * n_in, bits=16)
    n_g = 1 << ((oracle.ceil_log2(n_in * n_out) + 1) // 2)        # demo.cu:81
    gens = rand_points(oracle, rng, n_g)
    com = oracle.commit(gens, W)
    Z = oracle.fc_forward(X, W, B, n_in, n_out)
    u_bs, u_in, u_out = rand_fr(rng, 1), rand_fr(rng, 2), rand_fr(rng, 3)
    frs, g1s = oracle.fc_prove(X, Z, W, n_in, n_out, gens, com, u_bs, u_in, u_out)
    a = oracle.fr_partial_me(X, u_bs, n_in)
    b = oracle.fr_partial_me(W, u_out, 1)
    ip = oracle.ip_sumcheck(a, b, u_in)
    assert np.array_equal(frs[:8], ip)
    assert np.array_equal(frs[8], oracle.fr_me(Z, np.concatenate([u_out, u_bs])))
    ret, com_eval, proof = oracle.open(gens, W, com, np.concatenate([u_out, u_in]))
    assert np.array_equal(frs[9], ret) and frs.shape[0] == 10
    assert np.array_equal(g1s[0], com_eval) and np.array_equal(g1s[1:], proof)
    # the FC claim: Z~(u_out||u_bs) equals <a, b>, which round 0 of the sumcheck opens
    pr = ints(ip)
    assert (2 * pr[0] + pr[1] + pr[2]) % R == bm.fr_from_mont(frs[8])

    A, sign, mag, rem = oracle.relu_forward(Z)
    N = B * n_out
    k = oracle.ceil_log2(N)
    ch = [rand_fr(rng, k + 5), rand_fr(rng, k + 5), rand_fr(rng, k + 4), rand_fr(rng, k + 4), rand_fr(rng, k),
          rand_fr(rng, k), rand_fr(rng, k)]
    tr = oracle.relu_prove(Z, A, sign, mag, rem, *ch)
    parts = [oracle.bin_sumcheck(mag, ch[0], ch[1]), oracle.fr_partial_me(mag, ch[4], 32),
             oracle.bin_sumcheck(rem, ch[2], ch[3]), oracle.fr_partial_me(rem, ch[4], 16),
             oracle.hp_sumcheck(Z, sign, ch[5], ch[6])]
    assert np.array_equal(tr, np.concatenate(parts))
