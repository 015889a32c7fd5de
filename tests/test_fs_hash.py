"""Host-only parts of the verifier's plumbing: the library's SHA-256 against hashlib, and the challenge derivation
format of zk_fs_derive recomputed independently (so a verifier written against the header's description derives the
same challenges).  No GPU work: both functions are pure host code."""
import hashlib
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zkdl_amd.capi import fs_derive, sha256  # noqa: E402
from zkdl_amd.proof import call_order, fs_challenges, layer_shapes, parse_transcript, split_calls  # noqa: E402

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def test_sha256_matches_hashlib():
    rng = np.random.default_rng(3)
    for n in (0, 1, 3, 55, 56, 57, 63, 64, 65, 119, 120, 1000, 100000):
        data = rng.bytes(n)
        assert sha256(data) == hashlib.sha256(data).digest(), n


def test_fs_derive_format_and_distribution():
    seed = hashlib.sha256(b"statement").digest()
    got = fs_derive(seed, 5, 1, 3, 40)
    for i in range(40):
        d = hashlib.sha256(b"zkdl-amd/fs/v1" + seed + struct.pack("<IIII", 5, 1, 3, i)).digest()
        limbs = list(struct.unpack("<8I", d))
        limbs[7] %= 1944954707                       # proof.cu:9: keeps the value below r
        assert got[i].tolist() == limbs
        assert int.from_bytes(got[i].tobytes(), "little") < R_MOD
    # different tags give unrelated vectors; the same tags reproduce
    assert not np.array_equal(fs_derive(seed, 5, 1, 2, 4), got[:4])
    assert np.array_equal(fs_derive(seed, 5, 1, 3, 4), got[:4])
    assert fs_derive(seed, 0, 0, 0, 0).shape == (0, 8)


def test_challenge_schedule_follows_the_reference_draw_order():
    dims, batch = [6, 5, 4], 3
    chal = fs_challenges(b"\x07" * 32, dims, batch)
    assert [(k, i) for k, i, _ in chal] == call_order(2) == [("fc", 1), ("relu", 0), ("fc", 0)]
    sh = layer_shapes(dims)
    assert [c.shape[0] for c in chal[0][2]] == [2, 3, 2]                  # u_bs, u_in, u_out (zkfc.cu:135-137)
    k = 2 + 3
    assert [c.shape[0] for c in chal[1][2]] == [k + 5, k + 5, k + 4, k + 4, k, k, k]   # zkrelu.cu:85-98
    assert sh[0]["n_gens"] == 8 and sh[1]["n_gens"] == 8 and sh[0]["m"] == 8 and sh[1]["m"] == 4   # demo.cu:81: 1 << ((ceilLog2(in*out) + 1) / 2)


def test_transcript_parser_rejects_malformed_blobs():
    import pytest
    good = b"ZKDLTR01" + struct.pack("<QQ", 2, 1) + bytes(64) + bytes(96) + b"\x01"
    fr, aff, inf = parse_transcript(good)
    assert fr.shape == (2, 8) and aff.shape == (1, 24) and inf.tolist() == [1]
    with pytest.raises(ValueError):
        parse_transcript(good[:-1])
    with pytest.raises(ValueError):
        parse_transcript(b"ZKDLTR02" + good[8:])
    with pytest.raises(ValueError):
        split_calls([6, 5, 4], 3, fr, aff, inf)
