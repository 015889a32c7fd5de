"""Independent big-integer model of BLS12-381 Fr / Fp / G1 (Python ints, affine group law).

Shares no code and no algorithm with the reference or the oracle (no limbs, no Montgomery
reduction, no Jacobian formulas): it pins the oracle through public curve parameters only.
"""
import numpy as np

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
RR = 1 << 256   # Montgomery radix for Fr
RP = 1 << 384   # Montgomery radix for Fp


def limbs_to_int(a):
    a = np.asarray(a, dtype=np.uint32).reshape(-1)
    return int.from_bytes(a.tobytes(), "little")


def int_to_limbs(x, n):
    return np.frombuffer(int(x).to_bytes(4 * n, "little"), dtype=np.uint32).copy()


def fr_from_mont(a):
    return limbs_to_int(a) * pow(RR, -1, R_MOD) % R_MOD


def fr_to_mont(x):
    return int_to_limbs(x % R_MOD * RR % R_MOD, 8)


def fp_from_mont(a):
    return limbs_to_int(a) * pow(RP, -1, P_MOD) % P_MOD


def fp_to_mont(x):
    return int_to_limbs(x % P_MOD * RP % P_MOD, 12)


# ---- affine group law; None is the point at infinity
def g1_add(P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % P_MOD == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P_MOD) % P_MOD
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P_MOD) % P_MOD
    x3 = (lam * lam - x1 - x2) % P_MOD
    y3 = (lam * (x1 - x3) - y1) % P_MOD
    return (x3, y3)


def g1_neg(P):
    return None if P is None else (P[0], (-P[1]) % P_MOD)


def g1_mul(P, k):
    acc = None
    while k:
        if k & 1:
            acc = g1_add(acc, P)
        P = g1_add(P, P)
        k >>= 1
    return acc


def on_curve(P):
    return P is None or (P[1] * P[1] - P[0] ** 3 - 4) % P_MOD == 0


def jac_to_affine_int(j):
    """(36,) uint32 Montgomery Jacobian -> affine ints or None."""
    j = np.asarray(j, dtype=np.uint32).reshape(36)
    X, Y, Z = fp_from_mont(j[0:12]), fp_from_mont(j[12:24]), fp_from_mont(j[24:36])
    if Z == 0:
        return None
    zi = pow(Z, -1, P_MOD)
    return (X * zi * zi % P_MOD, Y * zi * zi * zi % P_MOD)


def affine_int_to_jac(P):
    out = np.zeros(36, dtype=np.uint32)
    if P is None:
        out[12:24] = fp_to_mont(1)
        return out
    out[0:12] = fp_to_mont(P[0])
    out[12:24] = fp_to_mont(P[1])
    out[24:36] = fp_to_mont(1)
    return out


def affine_mont_to_int(a, inf=False):
    if inf:
        return None
    a = np.asarray(a, dtype=np.uint32).reshape(24)
    return (fp_from_mont(a[0:12]), fp_from_mont(a[12:24]))


# ---- multilinear-extension helpers over Fr (plain ints mod r), LSB-first challenges
def eq_weights(u):
    w = [1]
    for ui in u:
        w = [x * (1 - ui) % R_MOD for x in w] + [x * ui % R_MOD for x in w]
    return w


def mle(t, u):
    n = 1 << len(u)
    t = list(t) + [0] * (n - len(t))
    w = eq_weights(u)
    return sum(a * b for a, b in zip(t, w)) % R_MOD
