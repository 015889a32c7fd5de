"""HIP kernels against the reference-held vectors DIRECTLY: tests/golden/prims.json holds operands and outputs of
the reference's own compiled arithmetic (/root/reference/bls12-381.cu via oracle/_ref, frozen by
oracle/gen_golden.py).  The operands go through the C ABI (zk_fr_*, zk_g1_add / zk_g1_madd / zk_g1_mul) and the
results are compared limb for limb with the stored outputs -- no oracle in between, nothing read from
/root/reference at run time."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "prims.json")


def unhex(lst, w):
    return np.frombuffer(bytes.fromhex("".join(lst)), dtype=np.uint32).reshape(-1, w).copy()


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def zk():
    from zkdl_amd.capi import Zk
    z = Zk(0)
    yield z
    z.close()


def test_fr_kernels_reproduce_the_reference_vectors(zk, gold):
    """Scalar_add / sub / mul / sqr / double / mont / unmont (bls12-381.cu:289-300,499-556,585-593): random operands
    plus the edge set (0, 1, r-1, R mod r, carries across every limb)."""
    sec = gold["fr"]
    a, b, un = zk.upload(unhex(sec["a"], 8)), zk.upload(unhex(sec["b"], 8)), zk.upload(unhex(sec["un"], 8))
    for op, fn in (("add", zk.fr_add), ("sub", zk.fr_sub), ("mul", zk.fr_mul)):
        assert np.array_equal(fn(a, b).download(), unhex(sec[op], 8)), op
    assert np.array_equal(zk.fr_mul(un, un).download(), unhex(sec["sqr"], 8))
    assert np.array_equal(zk.fr_add(un, un).download(), unhex(sec["double"], 8))
    assert np.array_equal(zk.fr_mont(un).download(), unhex(sec["mont"], 8))
    assert np.array_equal(zk.fr_unmont(un).download(), unhex(sec["unmont"], 8))
    # the broadcast-scalar entry points run the same device functions: row i against b[0]
    b0 = unhex(sec["b"], 8)[:1]
    bb = zk.upload(np.repeat(b0, a.n, axis=0))
    for fn_s, fn_t in ((zk.fr_add_scalar, zk.fr_add), (zk.fr_sub_scalar, zk.fr_sub), (zk.fr_mul_scalar, zk.fr_mul)):
        assert np.array_equal(fn_s(a, b0).download(), fn_t(a, bb).download())


def test_g1_kernels_reproduce_the_reference_vectors(zk, gold):
    """G1Affine_add / _double / _add_mixed (bls12-381.cu:1332-1435), Jacobian limbs compared bit for bit, including
    the special cases the fixture holds (inf + P, P + inf, P + P, P + (-P), 2 * inf).  Every Fp operation of
    bls12-381.cu:652-963 is exercised underneath."""
    s = gold["g1_add"]
    got = zk.g1_add(zk.upload(unhex(s["a"], 36)), zk.upload(unhex(s["b"], 36))).download()
    assert np.array_equal(got, unhex(s["out"], 36))
    s = gold["g1_double"]                       # reached through add's equal-operands branch (bls12-381.cu:1413-1414)
    a = zk.upload(unhex(s["a"], 36))
    assert np.array_equal(zk.g1_add(a, a).download(), unhex(s["out"], 36))
    s = gold["g1_add_mixed"]
    got = zk.g1_madd(zk.upload(unhex(s["a"], 36)), zk.upload(unhex(s["b"], 24))).download()
    assert np.array_equal(got, unhex(s["out"], 36))


def test_g1_scalar_mul_reproduces_the_reference_vectors(zk, gold):
    """G1Jacobian_mul's op sequence (g1-tensor.cu:422-430) evaluated with the reference's compiled add / double:
    k * G for k in {0, 1, 2, r-1, r, r+1, 2^256-1, random ...}."""
    s = gold["g1_mul"]
    k = unhex(s["k"], 8)
    base = np.repeat(unhex([s["base"]], 36), k.shape[0], axis=0)
    got = zk.g1_mul(zk.upload(base), zk.upload(k)).download()
    assert np.array_equal(got, unhex(s["out"], 36))
    # broadcast form: one generator, all scalars
    got = zk.g1_mul_bcast(zk.upload(base[:1]), zk.upload(k)).download()
    assert np.array_equal(got, unhex(s["out"], 36))


def test_constants_are_the_reference_constants(zk, gold):
    """mont(1) = ONE = R mod r and unmont(ONE) = 1 with the reference's limb values (bls12-381.cu:3-5)."""
    one = np.zeros((1, 8), np.uint32)
    one[0, 0] = 1
    m = zk.fr_mont(zk.upload(one)).download()
    assert m.tobytes().hex() == gold["constants"]["fr_one"]
    assert np.array_equal(zk.fr_unmont(zk.upload(m)).download(), one)
    r2 = unhex([gold["constants"]["fr_r2"]], 8)
    assert np.array_equal(zk.fr_mul(zk.upload(one), zk.upload(r2)).download(), m)


def test_every_device_field_function_reproduces_the_reference_vectors(zk, gold):
    """zk_field_selftest runs the device functions the kernels are built from, one element per lane: Scalar_* AND Fp_* (bls12-381.cu:499-593, 652-963) as
    field.hpp implements them -- f_double and f_sqr directly, not through a + a / a * a -- and the same operations through the lazily reduced limb forms
    the hot loops use (fs29.hpp: 9 x 29 bits for the big sumcheck rounds, incl. the unreduced column accumulation; fq28.hpp: 14 x 28 bits for the MSM
    pipeline).  Expected values: the reference's own compiled arithmetic (tests/golden/prims.json)."""
    ADD, SUB, MUL, SQR, DOUBLE, MONT, UNMONT, NEG = range(8)
    for name, width, plain, lazy in (("fr", 8, 0, 2), ("fp", 12, 1, 3)):
        sec = gold[name]
        a, b, un = zk.upload(unhex(sec["a"], width)), zk.upload(unhex(sec["b"], width)), zk.upload(unhex(sec["un"], width))
        for which in (plain, lazy):
            for op, key in ((ADD, "add"), (SUB, "sub"), (MUL, "mul")):
                assert np.array_equal(zk.field_selftest(which, op, a, b).download(), unhex(sec[key], width)), (name, which, key)
            for op, key in ((SQR, "sqr"), (DOUBLE, "double")):
                assert np.array_equal(zk.field_selftest(which, op, un).download(), unhex(sec[key], width)), (name, which, key)
        for op, key in ((MONT, "mont"), (UNMONT, "unmont")):
            assert np.array_equal(zk.field_selftest(plain, op, un).download(), unhex(sec[key], width)), (name, key)
        # neg: a + (-a) = 0 through the reference-checked add
        neg = zk.field_selftest(plain, NEG, un)
        assert not zk.field_selftest(plain, ADD, un, neg).download().any()
    # 29-bit form, the unreduced product accumulated as columns and reduced once (what the round coefficients are made of) == the Montgomery product
    sec = gold["fr"]
    a, b = zk.upload(unhex(sec["a"], 8)), zk.upload(unhex(sec["b"], 8))
    assert np.array_equal(zk.field_selftest(2, MONT, a, b).download(), unhex(sec["mul"], 8))
    # 28-bit form: negation
    un = zk.upload(unhex(gold["fp"]["un"], 12))
    assert not zk.field_selftest(1, ADD, un, zk.field_selftest(3, NEG, un)).download().any()
    assert np.array_equal(zk.field_selftest(1, MUL, zk.upload(unhex([gold["constants"]["fp_one"]], 12)), zk.upload(unhex([gold["constants"]["fp_one"]], 12))).download(),
                          unhex([gold["constants"]["fp_one"]], 12))
