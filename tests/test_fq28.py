"""zkdl_amd/csrc/fq28.hpp + g1q.hpp (the 28-bit-limb base field the MSM bucket pipeline, the derived round commitments and the
canonicalisation compute in) against Python integers, on the CPU: the headers are plain C++ apart from their __host__ __device__
markers, so the same source the kernels use is compiled here."""
import os
import random
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R384 = (1 << 384) % P
RINV = pow(1 << 384, -1, P)
GX = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
GY = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    cxx = CLANG if os.path.exists(CLANG) else shutil.which("clang++")
    if not cxx:
        pytest.skip("no clang++ (field.hpp uses __builtin_addc / __builtin_subc)")
    out = str(tmp_path_factory.mktemp("fq28") / "fq28_host")
    subprocess.run([cxx, "-std=c++17", "-O1", "-o", out, os.path.join(ROOT, "tests", "native", "fq28_host.cpp")], check=True)
    return out


def run(exe, lines):
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
    return [int(x, 16) for x in out.stdout.split()]


def mont(x):
    return x * R384 % P


def vals(rng, n):
    edge = [0, 1, 2, P - 1, P - 2, (1 << 28) - 1, 1 << 28, (1 << 364) - 1, 1 << 364, 1 << 380, P >> 1]
    return [rng.choice(edge) if rng.random() < 0.25 else rng.randrange(P) for _ in range(n)]


def test_field_products_inverse_and_lazy_operands(exe):
    rng = random.Random(28)
    a, b = vals(rng, 300), vals(rng, 300)
    got = run(exe, ["mul %x %x" % (mont(x), mont(y)) for x, y in zip(a, b)])
    assert got == [mont(x * y) for x, y in zip(a, b)]
    got = run(exe, ["lazy %x %x" % (mont(x), mont(y)) for x, y in zip(a, b)])
    assert got == [mont((x + y) * (x - y)) for x, y in zip(a, b)]
    got = run(exe, ["sqr %x" % mont(x) for x in a])
    assert got == [mont(x * x) for x in a]
    nz = [x for x in a if x][:40]
    got = run(exe, ["inv %x" % mont(x) for x in nz])
    assert got == [mont(pow(x, -1, P)) for x in nz]


def test_binary_gcd_inverse_equals_the_fermat_chain_and_python(exe):
    """fpinv.hpp (optimised binary GCD: 25 groups of 31 halving steps on 64-bit approximations) against pow(x, -1, p) and against the Fermat
    chains it replaced, in both limb forms; 0 -> 0; operands that stress the approximations (tiny values, values just below p, powers of two,
    a and b agreeing in their top bits)."""
    rng = random.Random(31)
    edge = [0, 1, 2, 3, P - 1, P - 2, P - 3, (P - 1) // 2, (P + 1) // 2, (1 << 31) - 1, 1 << 31, (1 << 31) + 1, (1 << 62) - 1, 1 << 62, (1 << 64) - 1, 1 << 64,
            (1 << 64) + 1, (1 << 96) - 1, 1 << 380, (1 << 380) + 1, P - (1 << 32), P - (1 << 64) + 1, P >> 1, P >> 2, P >> 33, (P >> 33) + 1]
    xs = edge + [rng.randrange(P) for _ in range(700)] + [rng.randrange(1 << rng.randrange(1, 381)) for _ in range(300)]
    want = [pow(x, -1, P) if x else 0 for x in xs]
    assert run(exe, ["invint %x" % x for x in xs]) == want                                   # plain integers
    assert run(exe, ["inv32 %x" % mont(x) for x in xs]) == [mont(w) for w in want]              # Montgomery form in, Montgomery form out
    assert run(exe, ["inv %x" % mont(x) for x in xs]) == [mont(w) for w in want]                # through the 28-bit-limb form
    sub = xs[:60]
    assert run(exe, ["inv32f %x" % mont(x) for x in sub]) == [mont(w) for w in want[:60]]       # the chains it replaced agree
    assert run(exe, ["invf %x" % mont(x) for x in sub if x]) == [mont(pow(x, -1, P)) for x in sub if x]


# ---- affine reference arithmetic
def ec_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return x3, (lam * (x1 - x3) - y1) % P


def ec_mul(k, p):
    r = None
    while k:
        if k & 1:
            r = ec_add(r, p)
        p = ec_add(p, p)
        k >>= 1
    return r


def jac(rng, pt):
    """a random Jacobian representative, Montgomery form, as the three hex words the driver reads"""
    if pt is None:
        return "0 %x 0" % mont(1)
    z = rng.randrange(1, P)
    return "%x %x %x" % (mont(pt[0] * z * z), mont(pt[1] * z * z * z), mont(z))


def canon(pt):
    return [0, mont(1), 0] if pt is None else [mont(pt[0]), mont(pt[1]), mont(1)]


def test_group_law_and_canonical_form(exe):
    rng = random.Random(381)
    g = (GX, GY)
    pts = [ec_mul(rng.randrange(1, 1 << 64), g) for _ in range(6)]
    lines, want = [], []
    for p in pts + [None]:
        lines.append("dbl " + jac(rng, p))
        want += canon(ec_add(p, p))
    for p in pts[:4]:
        for q in pts[2:] + [None, p, (p[0], (-p[1]) % P)]:      # generic, identity, doubling and inverse cases
            lines.append("add " + jac(rng, p) + " " + jac(rng, q))
            want += canon(ec_add(p, q))
    for p, q in zip(pts, pts[1:]):
        lines.append("chain " + jac(rng, p) + " " + jac(rng, q))
        a = ec_add(ec_add(p, p), q)
        for _ in range(5):
            a = ec_add(a, a)
        want += canon(ec_add(ec_add(a, (q[0], (-q[1]) % P)), p))
    assert run(exe, lines) == want
