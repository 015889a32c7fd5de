"""zkdl_amd/csrc/fs29.hpp (the 29-bit-limb scalar field the big sumcheck rounds compute in) against Python integers, on the CPU:
the header is plain C++ apart from its __host__ __device__ markers, so the same source the kernels inline is compiled here."""
import os
import random
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R_MOD = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
RINV = pow(1 << 256, -1, R_MOD)
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    cxx = CLANG if os.path.exists(CLANG) else shutil.which("clang++")
    if not cxx:
        pytest.skip("no clang++ (field.hpp uses __builtin_addc / __builtin_subc)")
    out = str(tmp_path_factory.mktemp("fs29") / "fs29_host")
    subprocess.run([cxx, "-std=c++17", "-O1", "-o", out, os.path.join(ROOT, "tests", "native", "fs29_host.cpp")], check=True)
    return out


def run(exe, lines):
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
    return [int(x, 16) for x in out.stdout.split()]


def vals(rng, n):
    edge = [0, 1, 2, R_MOD - 1, R_MOD - 2, (1 << 29) - 1, 1 << 29, (1 << 232) - 1, 1 << 232, (1 << 254), R_MOD >> 1]
    return [rng.choice(edge) if rng.random() < 0.25 else rng.randrange(R_MOD) for _ in range(n)]


def test_products_and_lazy_operands(exe):
    rng = random.Random(29)
    a, b = vals(rng, 300), vals(rng, 300)
    got = run(exe, ["mul %x %x" % (x, y) for x, y in zip(a, b)])
    assert got == [x * y * RINV % R_MOD for x, y in zip(a, b)]
    got = run(exe, ["lazy %x %x" % (x, y) for x, y in zip(a, b)])
    assert got == [(x + y) * (x - y) * RINV % R_MOD for x, y in zip(a, b)]
    c = vals(rng, 300)
    got = run(exe, ["fold %x %x %x" % t for t in zip(a, b, c)])
    assert got == [(x + z * (y - x) * RINV) % R_MOD for x, y, z in zip(a, b, c)]
    got = run(exe, ["pack %x" % x for x in a])
    assert got == a


def test_weighted_pair_coefficients(exe):
    """The per-pair arithmetic of k_sc_round29 (binary and Hadamard rounds) on Montgomery-form operands."""
    rng = random.Random(5)
    RR = (1 << 256) % R_MOD
    for _ in range(50):
        a0, a1, b0, b1, wa, wb = vals(rng, 6)
        got = run(exe, ["pair %x %x %x %x %x %x" % (a0, a1, b0, b1, wa, wb)])
        w = wa * wb * RINV % R_MOD
        da, db = a1 - a0, b1 - b0
        m = lambda x, y: x * y * RINV % R_MOD          # Montgomery product
        want = [m(m(a0, w), a0 - RR), m(m(da, w), 2 * a0 - RR), m(m(da, w), da),
                m(m(a0, w), b0), m(m(a0, w) + m(da, w), b1), m(m(da, w), db)]
        assert got == [x % R_MOD for x in want]


@pytest.mark.parametrize("op", ["raw", "raw64"])        # 32-bit columns cut per product / 64-bit columns across products (what k_sc_round29 keeps per lane)
@pytest.mark.parametrize("n", [1, 6, 7, 8, 9, 64, 200, 2048])
def test_unreduced_accumulation(exe, n, op):
    rng = random.Random(n)
    a, b = vals(rng, n), vals(rng, n)
    if n == 2048:       # the largest total a workgroup accumulates: every operand at its bound
        a, b = [R_MOD - 1] * n, [1] * (n // 2) + [R_MOD - 1] * (n // 2)
    line = "%s %d " % (op, n) + " ".join("%x %x" % t for t in zip(a, b))
    got = run(exe, [line])
    assert got == [sum((x - y) * y for x, y in zip(a, b)) * RINV % R_MOD]
