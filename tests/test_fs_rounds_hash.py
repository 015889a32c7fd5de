"""The round-by-round Fiat-Shamir chain (csrc/sha256.hpp, zk_fs_round_challenges: host code, no GPU) against hashlib."""
import hashlib
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def model(seed, layer, kind, which, coeffs):
    st = hashlib.sha256(b"zkdl-amd/fs-rounds/v1" + seed + struct.pack("<III", layer, kind, which)).digest()
    out = []
    for r in range(coeffs.shape[0] // 3):
        st = hashlib.sha256(st + coeffs[3 * r:3 * r + 3].astype("<u4").tobytes()).digest()
        limbs = np.frombuffer(st, dtype="<u4").astype(np.uint32).copy()
        limbs[7] = np.uint32(int(limbs[7]) % 1944954707)
        out.append(limbs)
    return np.array(out, dtype=np.uint32).reshape(-1, 8)


def test_round_challenges_follow_the_documented_hash_chain():
    from zkdl_amd.capi import fs_round_challenges
    rng = np.random.default_rng(3)
    seed = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    def canonical(rows):                                     # round messages are field elements: below r (top limb below r's top limb is enough)
        c = rng.integers(0, 1 << 32, size=(rows, 8), dtype=np.uint64).astype(np.uint32)
        c[:, 7] %= np.uint32(1944954707)
        return c
    for k in (0, 1, 2, 11, 24):
        coeffs = canonical(3 * k)
        got = fs_round_challenges(seed, 5, 1, 2, coeffs)
        assert np.array_equal(got, model(seed, 5, 1, 2, coeffs)), k
    coeffs = canonical(9)
    # c and c + r are the same field element but different hash inputs: an unreduced message is refused, not hashed
    from zkdl_amd.capi import ZkError
    bad = coeffs.copy()
    bad[2, :] = 0xffffffff
    with pytest.raises(ZkError):
        fs_round_challenges(seed, 0, 0, 0, bad)
    base = fs_round_challenges(seed, 0, 0, 0, coeffs)
    # a changed round message changes that round's challenge and every later one, and nothing before it
    c2 = coeffs.copy()
    c2[4, 3] ^= 1                                            # round 1's second coefficient
    other = fs_round_challenges(seed, 0, 0, 0, c2)
    assert np.array_equal(other[0], base[0]) and not np.array_equal(other[1], base[1]) and not np.array_equal(other[2], base[2])
    # domain separation: layer, kind, which, seed
    for args in ((1, 0, 0), (0, 1, 0), (0, 0, 1)):
        assert not np.array_equal(fs_round_challenges(seed, *args, coeffs)[0], base[0])
    assert not np.array_equal(fs_round_challenges(bytes(32), 0, 0, 0, coeffs)[0], base[0])
    assert all(int(x[7]) < 1944954707 for x in base)
