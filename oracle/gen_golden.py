#!/usr/bin/env python3
"""Generate tests/golden/prims.json from the REFERENCE'S OWN compiled arithmetic (oracle/_ref).

Run in the build container only (needs /root/reference to build oracle/_ref):
    python oracle/gen_golden.py
The reference ships no known-answer tests; these vectors are outputs of its bls12-381.cu
(Fr/Fp add, sub, mul, sqr, double, mont, unmont; G1 double, add, add_mixed) on seeded random and
edge operands, plus k*G chains driven through its add/double in the LSB-first order of
g1-tensor.cu:422-430.  Fixtures are data only: inputs and expected outputs as little-endian hex.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.oracle_py import RefPrims, build  # noqa: E402

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def to_limbs(x, n):
    return np.frombuffer(int(x).to_bytes(4 * n, "little"), dtype=np.uint32).copy()


def hexrow(a):
    return np.ascontiguousarray(a, dtype=np.uint32).tobytes().hex()


def hexlist(arr):
    return [hexrow(r) for r in arr]


def edge_values(mod, n):
    vals = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (mod + 1) // 2, (1 << (32 * n)) % mod, pow(1 << (32 * n), 2, mod)]
    for k in range(1, n + 1):            # carries across every limb boundary
        vals.append(((1 << (32 * k)) - 1) % mod)
        vals.append((1 << (32 * k - 1)) % mod)
    vals.append(mod - ((1 << 47)))       # relu window edges live here for Fr
    return [v % mod for v in vals]


def field_section(ref, kind, mod, n, rng, top):
    edges = np.stack([to_limbs(v, n) for v in edge_values(mod, n)])
    rnd = rng.integers(0, 1 << 32, size=(48, n), dtype=np.uint64).astype(np.uint32)
    rnd[:, n - 1] = (rnd[:, n - 1].astype(np.uint64) % top).astype(np.uint32)
    ops = np.concatenate([edges, rnd])
    # edge pairs (each edge with itself, neighbours, 0, p-1, p-2) + random pairs
    E = len(edges)
    pairs = sorted({(i, j % E) for i in range(E) for j in (i, i + 1, i + 3, i + 7, 0, 3, 4)})
    ia = np.array([p[0] for p in pairs]); ib = np.array([p[1] for p in pairs])
    a = np.concatenate([edges[ia], rnd[:24]])
    b = np.concatenate([edges[ib], rnd[24:]])
    f = getattr(ref, kind)
    sec = {"a": hexlist(a), "b": hexlist(b), "un": hexlist(ops)}
    for op in ("add", "sub", "mul"):
        sec[op] = hexlist(f(op, a, b))
    for op in ("sqr", "double", "mont", "unmont"):
        sec[op] = hexlist(f(op, ops))
    return sec


def ref_scalar_mul(ref, P, k_limbs):
    """g1-tensor.cu:422-430 order, each step executed by the reference's add/double."""
    zero = np.zeros((1, 36), dtype=np.uint32)
    zero[0, 12:24] = ref.constants()["fp_one"]
    out, a = zero.copy(), P.reshape(1, 36).copy()
    for i in range(256):
        if (int(k_limbs[i // 32]) >> (i % 32)) & 1:
            out = ref.g1_add(out, a)
        a = ref.g1_double(a)
    return out[0]


def main():
    build()
    ref = RefPrims()
    rng = np.random.default_rng(381)
    c = ref.constants()
    gold = {"_meta": {"source": "oracle/_ref/libzkdl_ref.so = /root/reference/bls12-381.cu (OpenCL-C branch, x86-64)",
                      "generator": "oracle/gen_golden.py", "encoding": "little-endian hex of uint32 limbs"}}
    gold["constants"] = {k: hexrow(v) for k, v in c.items()}
    gold["fr"] = field_section(ref, "fr", R_MOD, 8, rng, 1944954707)
    gold["fp"] = field_section(ref, "fp", P_MOD, 12, rng, 436277738)

    # ---- G1
    gx = [4250078230, 1555269520, 2574712821, 2014837863, 339452353, 357537223, 4090554183, 4037962445, 568063040, 3989728972, 2651585397, 302085953]
    gy = [216474225, 3131872213, 2031680910, 2351063834, 1460086222, 3713621779, 1346392468, 1370249257, 2902481344, 236751935, 1342743146, 196886268]
    G = np.array(gx + gy + list(c["fp_one"]), dtype=np.uint32)   # g1-tensor.cuh:29-63
    Gaff = G[:24].copy()
    inf = np.zeros(36, dtype=np.uint32); inf[12:24] = c["fp_one"]            # bls12-381.cuh:417
    ks = [0, 1, 2, 3, 5, R_MOD - 1, R_MOD, R_MOD + 1, (1 << 255) + 12345, (1 << 256) - 1]
    ks += [int.from_bytes(rng.bytes(32), "little") % R_MOD for _ in range(6)]
    kl = np.stack([to_limbs(k, 8) for k in ks])
    kG = np.stack([ref_scalar_mul(ref, G, k) for k in kl])
    gold["g1_mul"] = {"base": hexrow(G), "k": hexlist(kl), "out": hexlist(kG)}
    negG = G.copy(); negG[12:24] = ref.fp("sub", np.zeros((1, 12), np.uint32), G[12:24].reshape(1, 12))[0]
    pts = np.stack([G, negG, inf] + [kG[i] for i in (2, 3, 4, 5, 10, 11, 12, 13, 14, 15)])
    ia, ib = np.meshgrid(np.arange(len(pts)), np.arange(len(pts)), indexing="ij")
    a, b = pts[ia.ravel()], pts[ib.ravel()]
    gold["g1_add"] = {"a": hexlist(a), "b": hexlist(b), "out": hexlist(ref.g1_add(a, b))}
    gold["g1_double"] = {"a": hexlist(pts), "out": hexlist(ref.g1_double(pts))}
    negGaff = negG[:24].copy()
    affs = np.stack([Gaff, negGaff])
    ia, ib = np.meshgrid(np.arange(len(pts)), np.arange(2), indexing="ij")
    a, b = pts[ia.ravel()], affs[ib.ravel()]
    gold["g1_add_mixed"] = {"a": hexlist(a), "b": hexlist(b), "out": hexlist(ref.g1_add_mixed(a, b))}

    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "prims.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(gold, f, indent=0, separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
