"""The LeakyReLU extension of the oracle (zo_leaky_forward / zo_leaky_prove; no reference counterpart: zkrelu.cu:11-41 is the reference's
only activation) against a Python-integer model of its definition, and its proof against the protocol invariants."""
import numpy as np
import pytest

import bigint_model as bm

R = bm.R_MOD


def signed_fr(vals):
    return np.array([bm.fr_to_mont(v % R) for v in vals], dtype=np.uint32).reshape(-1, 8)


def model(v, shift):
    """v: signed integer with |v| < 2^47 (scale 2^32) -> requantised output (scale 2^16) as a signed integer."""
    mag = v if v >= 0 else v + (1 << 47)                  # zkrelu.cu:16-24: the negative window is stored offset by 2^47
    rem_sign = (mag >> 15) & 1
    rem_mag = mag & 32767
    rem = rem_mag - (1 << 15) if rem_sign else rem_mag
    m = ((mag - rem) >> 16) & 0xffffffff
    # negative window: m = 2^31 + q, q = round(v / 2^16) < 0; floor(q / 2^shift) = (m >> shift) - 2^(31 - shift)
    return m if v >= 0 else (m >> shift) - (1 << (31 - shift))


@pytest.mark.parametrize("shift", [0, 3, 7, 31])
def test_leaky_forward_matches_its_definition(oracle, shift):
    rng = np.random.default_rng(17 + shift)
    vals = [int(x) for x in rng.integers(-(1 << 46), 1 << 46, size=300)] + [0, 1, -1, 32767, 32768, -32768, -32769, (1 << 47) - 1, -(1 << 47) + 1, 65536, -65536]
    X = signed_fr(vals)
    A, sign, mag, rem, low, nsign = oracle.leaky_forward(X, shift)
    A_relu, sign_r, mag_r, rem_r = oracle.relu_forward(X)
    assert np.array_equal(sign, sign_r) and np.array_equal(mag, mag_r) and np.array_equal(rem, rem_r)      # the same decomposition as zkReLU
    for i, v in enumerate(vals):
        got = bm.fr_from_mont(A[i])
        got = got - R if got > R // 2 else got
        assert got == model(v, shift), (v, shift)
        if v >= 0:
            assert np.array_equal(A[i], A_relu[i])
        assert bm.fr_from_mont(nsign[i]) == (0 if v >= 0 else 1)
        if v < 0:
            q = model(v, 0)
            assert q <= 0 and abs(q - v / 65536.0) <= 0.5 and model(v, shift) == q >> shift        # Python's >> floors, like the definition
    # low is the magnitude bits shift .. 31
    mb = np.array([[bm.fr_from_mont(mag[32 * i + j]) for j in range(32)] for i in range(len(vals))])
    for i in range(len(vals)):
        m = sum(int(mb[i, j]) << j for j in range(32))
        assert bm.fr_from_mont(low[i]) == m >> shift
    if shift == 0:                                           # identity: a signed requantisation, |A - v / 2^16| <= 1/2 (+1 for the negative window's offset rounding)
        for i, v in enumerate(vals):
            got = bm.fr_from_mont(A[i])
            got = got - R if got > R // 2 else got
            assert abs(got - v / 65536.0) <= 0.5, v


def test_leaky_proof_extends_the_relu_proof_by_one_hadamard_sumcheck(oracle):
    from util import rand_fr
    rng = np.random.default_rng(23)
    n, k = 24, 5
    vals = [int(x) for x in rng.integers(-(1 << 40), 1 << 40, size=n)]
    X = signed_fr(vals)
    A, sign, mag, rem, low, nsign = oracle.leaky_forward(X, 3)
    ch = [rand_fr(rng, m) for m in (k + 5, k + 5, k + 4, k + 4, k, k, k, k, k)]
    pr = oracle.leaky_prove(X, A, sign, mag, rem, low, nsign, *ch)
    base = oracle.relu_prove(X, A, sign, mag, rem, *ch[:7])
    assert np.array_equal(pr[:base.shape[0]], base)
    ext = [bm.fr_from_mont(x) for x in pr[base.shape[0]:]]
    assert len(ext) == 3 * k + 2
    # Hadamard recurrence (proof.cu:110-137) with the claim (low o nsign)~(u_lk) from the big-integer model
    u, v = [bm.fr_from_mont(x) for x in ch[7]], [bm.fr_from_mont(x) for x in ch[8]]
    prod = [bm.fr_from_mont(a) * bm.fr_from_mont(b) % R for a, b in zip(low, nsign)]
    claim = bm.mle(prod + [0] * ((1 << k) - n), u)
    for r in range(k):
        c0, c1, c2 = ext[3 * r:3 * r + 3]
        assert ((1 - u[r]) * c0 + u[r] * (c0 + c1 + c2)) % R == claim, r
        claim = (c0 + c1 * v[r] + c2 * v[r] * v[r]) % R
    assert ext[3 * k] * ext[3 * k + 1] % R == claim
    assert ext[3 * k] == bm.mle([bm.fr_from_mont(a) for a in low] + [0] * ((1 << k) - n), v)
    assert ext[3 * k + 1] == bm.mle([bm.fr_from_mont(a) for a in nsign] + [0] * ((1 << k) - n), v)
