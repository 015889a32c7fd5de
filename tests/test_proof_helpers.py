"""Host-side helpers of zkdl_amd/proof.py that need no GPU: transcript layout arithmetic and the seeds of the activation openings (SHA-256 through
the library's host code)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MLP18M = [784, 1000, 1773, 1773, 1773, 1773, 1773, 1124, 1000]


def test_activation_commitment_counts_and_claim_positions():
    from zkdl_amd.proof import act_claimed_values, act_commit_counts, act_open_lengths, call_order, ceil_log2, layer_shapes, split_calls
    assert act_commit_counts(MLP18M, 256) == [2 * (256 * 1024 // 1024)] + [2 * (256 * 2048 // 2048)] * 6      # 2 m_i, m_i = B_p out_p / n_gens (layer 0: 1024 generators)
    assert act_commit_counts([6, 5, 4], 3) == [2 * (4 * 8 // 8)]
    assert [a for a, _ in act_open_lengths(MLP18M, 256)] == [3 * (8 + 10 + 1) + 3] + [3 * (8 + 11 + 1) + 3] * 6
    # act_claimed_values reads the positions split_calls assigns: zkFC slice = [3 k_in coefficients][a(0)][b(0)][Z~][t(0)], zkReLU's last two = Z~(v_hp), sign~(v_hp)
    dims, batch = [12, 20, 9, 7], 5
    sh = layer_shapes(dims)
    kb = ceil_log2(8)
    n_fr = sum(3 * ceil_log2(s["in_p"]) + 4 for s in sh) + sum(3 * (kb + ceil_log2(s["out_p"]) + 5) + 1 + 32 + 3 * (kb + ceil_log2(s["out_p"]) + 4) + 1 + 16 +
                                                              3 * (kb + ceil_log2(s["out_p"])) + 2 for s in sh[:-1])
    n_g1 = sum(3 * ceil_log2(s["n_gens"]) + 2 for s in sh)
    fr = np.arange(n_fr * 8, dtype=np.uint32).reshape(n_fr, 8)
    calls = {(kind, i): f for kind, i, f, _, _ in split_calls(dims, batch, fr, np.zeros((n_g1, 24), np.uint32), np.zeros(n_g1, np.uint8))}
    assert list(calls) == call_order(len(sh))
    ys = act_claimed_values(dims, batch, fr)
    for i, (y1, y2, y3) in enumerate(ys):
        k_in, k_nx = ceil_log2(sh[i]["in_p"]), ceil_log2(sh[i + 1]["in_p"])
        assert np.array_equal(y1, calls[("fc", i)][3 * k_in + 2])              # Z_i~(u_out || u_bs)
        assert np.array_equal(y2, calls[("relu", i)][-2])                       # Z_i~(v_hp)
        assert np.array_equal(y3, calls[("fc", i + 1)][3 * k_nx])               # A_i~(u_in || u_bs): the next layer's a(0)


def test_activation_seeds_bind_commitments_values_and_points():
    from zkdl_amd.proof import act_commit_counts, act_layer_seed, act_open_challenges, act_seed_from_points
    dims, batch = [12, 20, 9, 7], 5
    n = sum(act_commit_counts(dims, batch))
    rng = np.random.default_rng(3)
    aff = rng.integers(0, 1 << 32, size=(n, 24), dtype=np.uint64).astype(np.uint32)
    inf = np.zeros(n, np.uint8)
    seed = bytes(range(32))
    s0 = act_seed_from_points(seed, dims, batch, aff, inf)
    assert s0 == act_seed_from_points(seed, dims, batch, aff.copy(), inf.copy()) and len(s0) == 32
    aff2 = aff.copy()
    aff2[-1, 5] ^= 1
    assert act_seed_from_points(seed, dims, batch, aff2, inf) != s0                   # any commitment point
    assert act_seed_from_points(bytes(32), dims, batch, aff, inf) != s0                # the statement
    with pytest.raises(ValueError):
        act_seed_from_points(seed, dims, batch, aff[:-1], inf[:-1])
    c0, c1 = act_open_challenges(s0, dims, batch), act_open_challenges(act_seed_from_points(seed, dims, batch, aff2, inf), dims, batch)
    assert len(c0) == 2 and not np.array_equal(c0[0][0], c1[0][0]) and not np.array_equal(c0[1][1], c1[1][1])
    ys = [rng.integers(0, 1 << 30, size=8).astype(np.uint32) for _ in range(3)]
    pts = [rng.integers(0, 1 << 30, size=(6, 8)).astype(np.uint32) for _ in range(3)]
    base = act_layer_seed(s0, 0, ys, pts)
    assert base != act_layer_seed(s0, 1, ys, pts)
    ys2 = [y.copy() for y in ys]
    ys2[1][0] ^= 1
    assert base != act_layer_seed(s0, 0, ys2, pts)
    pts2 = [p.copy() for p in pts]
    pts2[2][3, 1] ^= 1
    assert base != act_layer_seed(s0, 0, ys, pts2)
