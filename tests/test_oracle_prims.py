"""Pin the CPU oracle's field/curve arithmetic (oracle/zkdl_oracle.c):
  1. against tests/golden/prims.json (outputs of the reference's own bls12-381.cu, see
     oracle/gen_golden.py) -- runs everywhere;
  2. live against oracle/_ref on fresh random operands -- where the reference is present;
  3. against an independent big-integer model (tests/bigint_model.py).
"""
import json
import os

import numpy as np
import pytest

import bigint_model as bm
from util import rand_fp, rand_fr

GOLD = os.path.join(os.path.dirname(__file__), "golden", "prims.json")


def unhex(lst, w):
    return np.frombuffer(bytes.fromhex("".join(lst)), dtype=np.uint32).reshape(-1, w).copy()


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_constants_match_reference_and_public_parameters(oracle, gold):
    c = oracle.constants()
    for k, v in c.items():
        assert v.tobytes().hex() == gold["constants"][k], k
    assert bm.limbs_to_int(c["fr_p"]) == bm.R_MOD
    assert bm.limbs_to_int(c["fp_p"]) == bm.P_MOD
    assert bm.limbs_to_int(c["fr_one"]) == bm.RR % bm.R_MOD
    assert bm.limbs_to_int(c["fr_r2"]) == bm.RR * bm.RR % bm.R_MOD
    assert bm.limbs_to_int(c["fp_one"]) == bm.RP % bm.P_MOD
    assert bm.limbs_to_int(c["fp_r2"]) == bm.RP * bm.RP % bm.P_MOD
    # Montgomery constants: INV = -p^-1 mod 2^32 (bls12-381.cuh:119,221)
    assert (-pow(bm.R_MOD, -1, 1 << 32)) % (1 << 32) == 4294967295
    assert (-pow(bm.P_MOD, -1, 1 << 32)) % (1 << 32) == 4294770685
    g = oracle.generator()[0]
    assert bm.fp_from_mont(g[0:12]) == bm.GX and bm.fp_from_mont(g[12:24]) == bm.GY
    assert bm.on_curve((bm.GX, bm.GY))
    # hex quoted in g1-tensor.cuh:28,44 comments
    assert bm.limbs_to_int(g[0:12]) == 0x120177419E0BFB75EDCE6ECC21DBF440F0AE6ACDF3D0E747154F95C7143BA1C17817FC679976FFF55CB38790FD530C16


@pytest.mark.parametrize("kind,w", [("fr", 8), ("fp", 12)])
def test_field_ops_match_golden(oracle, gold, kind, w):
    sec = gold[kind]
    a, b, un = unhex(sec["a"], w), unhex(sec["b"], w), unhex(sec["un"], w)
    f = getattr(oracle, kind)
    for op in ("add", "sub", "mul"):
        assert np.array_equal(f(op, a, b), unhex(sec[op], w)), (kind, op)
    for op in ("sqr", "double", "mont", "unmont"):
        assert np.array_equal(f(op, un), unhex(sec[op], w)), (kind, op)


def test_g1_ops_match_golden(oracle, gold):
    s = gold["g1_add"]
    assert np.array_equal(oracle.g1_add1(unhex(s["a"], 36), unhex(s["b"], 36)), unhex(s["out"], 36))
    s = gold["g1_double"]
    assert np.array_equal(oracle.g1_double(unhex(s["a"], 36)), unhex(s["out"], 36))
    s = gold["g1_add_mixed"]
    assert np.array_equal(oracle.g1_add_mixed(unhex(s["a"], 36), unhex(s["b"], 24)), unhex(s["out"], 36))
    s = gold["g1_mul"]
    k = unhex(s["k"], 8)
    base = np.repeat(unhex([s["base"]], 36), k.shape[0], axis=0)
    assert np.array_equal(oracle.g1_mul(base, k), unhex(s["out"], 36))


def test_live_against_reference_arithmetic(oracle, refprims, rng):
    a, b = rand_fr(rng, 200), rand_fr(rng, 200)
    for op in ("add", "sub", "mul"):
        assert np.array_equal(oracle.fr(op, a, b), refprims.fr(op, a, b)), op
    for op in ("sqr", "double", "mont", "unmont"):
        assert np.array_equal(oracle.fr(op, a), refprims.fr(op, a)), op
    a, b = rand_fp(rng, 200), rand_fp(rng, 200)
    for op in ("add", "sub", "mul"):
        assert np.array_equal(oracle.fp(op, a, b), refprims.fp(op, a, b)), op
    for op in ("sqr", "double", "mont", "unmont"):
        assert np.array_equal(oracle.fp(op, a), refprims.fp(op, a)), op
    g = np.repeat(oracle.generator(), 24, axis=0)
    P = oracle.g1_mul(g, rand_fr(rng, 24))
    Q = np.roll(P, 5, axis=0)
    assert np.array_equal(oracle.g1_add1(P, Q), refprims.g1_add(P, Q))
    assert np.array_equal(oracle.g1_double(P), refprims.g1_double(P))
    aff, _ = oracle.g1_to_affine(Q)
    assert np.array_equal(oracle.g1_add_mixed(P, aff), refprims.g1_add_mixed(P, aff))


def test_field_ops_match_bigint_model(oracle, rng):
    a, b = rand_fr(rng, 64), rand_fr(rng, 64)
    ai = [bm.limbs_to_int(x) for x in a]
    bi = [bm.limbs_to_int(x) for x in b]
    rinv = pow(bm.RR, -1, bm.R_MOD)
    got = oracle.fr("mul", a, b)
    assert [bm.limbs_to_int(x) for x in got] == [x * y * rinv % bm.R_MOD for x, y in zip(ai, bi)]
    got = oracle.fr("add", a, b)
    assert [bm.limbs_to_int(x) for x in got] == [(x + y) % bm.R_MOD for x, y in zip(ai, bi)]
    got = oracle.fr("sub", a, b)
    assert [bm.limbs_to_int(x) for x in got] == [(x - y) % bm.R_MOD for x, y in zip(ai, bi)]
    a, b = rand_fp(rng, 64), rand_fp(rng, 64)
    ai = [bm.limbs_to_int(x) for x in a]
    bi = [bm.limbs_to_int(x) for x in b]
    pinv = pow(bm.RP, -1, bm.P_MOD)
    got = oracle.fp("mul", a, b)
    assert [bm.limbs_to_int(x) for x in got] == [x * y * pinv % bm.P_MOD for x, y in zip(ai, bi)]


def test_scalar_mul_matches_affine_group_law(oracle, rng):
    ks = [0, 1, 2, bm.R_MOD - 1, bm.R_MOD, 0xDEADBEEF] + [bm.limbs_to_int(x) for x in rand_fr(rng, 4)]
    k = np.stack([bm.int_to_limbs(x, 8) for x in ks])
    g = np.repeat(oracle.generator(), len(ks), axis=0)
    out = oracle.g1_mul(g, k)
    aff, inf = oracle.g1_to_affine(out)
    for i, kk in enumerate(ks):
        want = bm.g1_mul((bm.GX, bm.GY), kk)
        assert bm.jac_to_affine_int(out[i]) == want
        assert bm.affine_mont_to_int(aff[i], bool(inf[i])) == want
    assert inf[0] == 1 and inf[4] == 1          # 0*G and r*G are the point at infinity


def test_g1_special_cases(oracle, rng):
    G = oracle.generator()
    inf = np.zeros((1, 36), np.uint32)
    inf[0, 12:24] = oracle.constants()["fp_one"]
    negG = oracle.g1_neg(G)
    # inf + P = P, P + inf = P (limbs untouched); P + P = double(P); P + (-P) has Z == 0
    assert np.array_equal(oracle.g1_add1(inf, G), G)
    assert np.array_equal(oracle.g1_add1(G, inf), G)
    assert np.array_equal(oracle.g1_add1(G, G), oracle.g1_double(G))
    assert not oracle.g1_add1(G, negG)[0, 24:36].any()
    assert np.array_equal(oracle.g1_double(inf), inf)


def test_g1_sum_tree_and_me_agree_with_model(oracle, rng):
    n = 150                                   # crosses one 128-input block, ragged second block
    P = oracle.g1_mul(np.repeat(oracle.generator(), n, axis=0), rand_fr(rng, n))
    want = None
    for p in P:
        want = bm.g1_add(want, bm.jac_to_affine_int(p))
    assert bm.jac_to_affine_int(oracle.g1_sum(P)) == want
    # G1 multilinear evaluation == MSM with eq weights of the un-Montgomery'd challenges
    m, k = 6, 3
    u = rand_fr(rng, k)
    ui = [bm.fr_from_mont(x) for x in u]
    w = bm.eq_weights(ui)
    want = None
    for i in range(m):
        want = bm.g1_add(want, bm.g1_mul(bm.jac_to_affine_int(P[i]), w[i]))
    assert bm.jac_to_affine_int(oracle.g1_me(P[:m], u)) == want
