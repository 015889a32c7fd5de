"""The bucket-free MSM path (zk_gens_prepare_full; msm.hip: k_fulltable_build / k_msmf_accum / k_msmf_reduce): every signed digit of every window
of a generator set precomputed, so a full-width MSM is mixed additions of table entries plus one sum of lane partials.  Same group elements as
the bucket method and as the reference's literal op sequence (commitment.cu:43-92 through the oracle)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


@pytest.fixture(scope="module")
def zk():
    from zkdl_amd.capi import MSM_PIPPENGER, Zk
    z = Zk(0)
    z.set_msm_mode(MSM_PIPPENGER)
    yield z
    z.close()


def fr_of(vals):
    return np.array([np.frombuffer(int(v % R_MOD).to_bytes(32, "little"), dtype=np.uint32) for v in vals], dtype=np.uint32).reshape(-1, 8)


@pytest.mark.parametrize("n,c", [(8, 4), (64, 6), (256, 9), (100, 7)])
def test_full_table_msm_equals_bucket_msm_and_oracle(zk, oracle, n, c):
    from util import rand_fr, rand_points
    rng = np.random.default_rng(7 * n + c)
    pts = rand_points(oracle, rng, n)
    pts[n // 3] = 0                                           # a generator at infinity (Z = 0)
    q = 5
    sc = rand_fr(rng, q * n)
    edge = [0, 1, 2, R_MOD - 1, R_MOD - 2, (1 << 254) - 1, 1 << 253, (R_MOD - 1) // 2, (R_MOD + 1) // 2]
    sc[:len(edge)] = fr_of(edge)
    sc[n:n + n // 2] = 0                                      # a sparse row
    gb, gf = zk.gens(zk.upload(pts)), zk.gens(zk.upload(pts))
    gf.prepare_full(c)
    sd = zk.upload(sc)
    a_b, i_b = zk.g1_to_affine(zk.msm_batch(gb, sd))
    a_f, i_f = zk.g1_to_affine(zk.msm_batch(gf, sd))
    assert eq(a_f, a_b) and eq(i_f, i_b)
    zk.set_flag("full_tables", 0)                             # the flag turns the path off without freeing the tables
    a_o, i_o = zk.g1_to_affine(zk.msm_batch(gf, sd))
    zk.set_flag("full_tables", 1)
    assert eq(a_o, a_b) and eq(i_o, i_b)
    if n <= 64:                                               # the oracle's literal scalar multiplications (g1-tensor.cu:422-445) and sum
        want = []
        for row in range(q):
            prod = oracle.g1_mul(pts, sc[row * n:(row + 1) * n])
            acc = prod[0:1]
            for j in range(1, n):
                acc = oracle.g1_add(acc, prod[j:j + 1])
            want.append(acc[0])
        a_w, i_w = oracle.g1_to_affine(np.array(want))
        assert eq(a_f, a_w) and eq(i_f, i_w)


@pytest.mark.parametrize("dims,batch,derive", [([6, 5, 4], 3, 0), ([16, 32, 24, 8], 4, 1), ([16, 32, 24, 8], 4, 0), ([40, 70, 30], 20, 1)])
def test_whole_proof_on_full_tables_equals_the_oracle(zk, oracle, dims, batch, derive):
    from mlp_util import prove_mlp_gpu, prove_mlp_oracle, synth_mlp
    model = synth_mlp(oracle, dims, batch, seed=61)
    gens = [zk.gens(zk.upload(g)) for g in model["gens"]]
    for g in gens:
        g.prepare_full(5)
    zk.set_flag("derive_rounds", derive)
    try:
        fr_g, aff_g, inf_g, _ = prove_mlp_gpu(zk, model, gens_dev=gens)
    finally:
        zk.set_flag("derive_rounds", 0)
    fr_o, g1_o = prove_mlp_oracle(oracle, model)
    aff_o, inf_o = oracle.g1_to_affine(g1_o)
    assert eq(fr_g, fr_o)
    assert eq(aff_g, aff_o) and eq(inf_g, inf_o)


def test_full_size_proofs_on_full_tables_equal_the_bucket_ones(zk):
    """18M-parameter MLP, batch 256, 12-bit digit tables (83 GB): pipelined, derived, batched G1 phases -- value for value the transcripts of the
    bucket method; and the activation openings as well."""
    import bench
    from zkdl_amd.capi import Transcript
    dims, batch = bench.WORKLOADS["mlp18m_b256"]
    wl = bench.Workload(zk, dims, batch, seed=31, inflight=4, full_tables=12)
    assert wl.info["full_table_bytes"] > 80e9
    want = []
    zk.set_flag("full_tables", 0)
    tr = Transcript(zk)
    for st in range(4):
        wl.prove(wl.challenges(st), tr)
        fr, _ = tr.read()
        aff, inf = tr.read_affine()
        want.append((fr, aff, inf))
    zk.set_flag("full_tables", 1)
    for derive, g1_batch in ((1, 2), (0, 1)):
        zk.set_flag("derive_rounds", derive)
        zk.set_flag("g1_batch", g1_batch)
        try:
            for st in range(4):
                wl.prove(wl.challenges(st), wl.trs[st])
                wl.trs[st].seal()
            for st in range(4):
                fr, _ = wl.trs[st].read()
                aff, inf = wl.trs[st].read_affine()
                assert eq(fr, want[st][0]) and eq(aff, want[st][1]) and eq(inf, want[st][2]), (derive, g1_batch, st)
        finally:
            zk.set_flag("derive_rounds", 0)
            zk.set_flag("g1_batch", 1)
    tr.close()
