"""Shared test helpers: seeded Fr/G1 inputs in the reference's distributions."""
import numpy as np

FR_TOP = 1944954707  # top limb of r; fr-tensor.cu:346 / proof.cu:9 draw the top limb modulo this


def rand_fr(rng, n):
    """Uniform-ish canonical Fr limbs, as random_vec / random_kernel draw them (proof.cu:3-11)."""
    a = rng.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32)
    a[:, 7] = (a[:, 7].astype(np.uint64) % FR_TOP).astype(np.uint32)
    return a


def rand_fp(rng, n):
    a = rng.integers(0, 1 << 32, size=(n, 12), dtype=np.uint64).astype(np.uint32)
    a[:, 11] = (a[:, 11].astype(np.uint64) % 436277738).astype(np.uint32)
    return a


def small_signed_fr(oracle, rng, n, bits=13, mont=True):
    """Quantised-weight-like values: signed ints of `bits` magnitude bits, negative -> r - v."""
    v = rng.integers(-(1 << bits) + 1, 1 << bits, size=n)
    a = np.zeros((n, 8), dtype=np.uint32)
    a[:, 0] = np.abs(v).astype(np.uint32)
    neg = v < 0
    if neg.any():
        a[neg] = oracle.fr_neg(a[neg])
    return oracle.fr_mont(a) if mont else a


def rand_points(oracle, rng, n):
    """n random multiples of the generator (Jacobian, non-trivial Z)."""
    g = np.repeat(oracle.generator(), n, axis=0)
    return oracle.g1_mul(g, rand_fr(rng, n))
