"""Full-size cases of BASELINE.json (the 18M-parameter MLP at batch 256; the width-4096 deep MLP with its
2^24-pair commitments) through size-independent properties: the oracle cannot run at these sizes in
seconds, so the GPU transcript is checked by a mini-verifier (tests/bigint_model.py, Python integers)
against the protocol invariants of SURVEY.md App. E, by cross-checks between independent GPU code paths,
and by scheduling/window invariance.  Everything goes through the C ABI."""
import os
import sys

import numpy as np
import pytest

import bigint_model as bm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu
R = bm.R_MOD


def ints(a):
    return [bm.fr_from_mont(x) for x in np.asarray(a).reshape(-1, 8)]


def poly(c, x):
    return (c[0] + c[1] * x + c[2] * x * x) % R


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


@pytest.fixture(scope="module")
def zk():
    from zkdl_amd.capi import MSM_PIPPENGER, Zk
    z = Zk(0)
    z.set_msm_mode(MSM_PIPPENGER)
    yield z
    z.close()


def verify_fc(zk, fr, fc_x, fc_z, ch, k_in, z_is_unmont=False):
    """Inner-product sumcheck recurrence of one zkFC::prove (zkfc.cu:128-145): Z~(u_out||u_bs) = <a, b>, the
    final pair multiplies to the last claim, W~(u_out||u_in) equals the opening's scalar, and X~(u_in||u_bs)
    equals what an independent GPU path (fr_me) computes."""
    u_bs, u_in, u_out = ch
    pr = ints(fr)
    u_z = np.concatenate([u_out, u_bs])
    assert pr[3 * k_in + 2] == bm.fr_from_mont(zk.fr_me(fc_z, u_z))   # the Z~(u_Z) the transcript carries
    # the last layer is proved on the un-Montgomery'd output (demo.cu:119): the sumcheck then opens the MLE of
    # the Montgomery-form tensor, not the value the transcript carries (reference quirk)
    claim = bm.fr_from_mont(zk.fr_me(zk.fr_mont(fc_z), u_z)) if z_is_unmont else pr[3 * k_in + 2]
    ui = ints(u_in)
    for r in range(k_in):
        c = pr[3 * r:3 * r + 3]
        assert (poly(c, 0) + poly(c, 1)) % R == claim, ("ip round", r)
        claim = poly(c, ui[r])
    fa, fb = pr[3 * k_in], pr[3 * k_in + 1]
    assert fa * fb % R == claim
    assert fb == pr[3 * k_in + 3]                              # Commitment::open's return value
    assert fa == bm.fr_from_mont(zk.fr_me(fc_x, np.concatenate([u_in, u_bs])))


def verify_relu(zk, fr, relu, Z, ch, k):
    """zkReLU::prove (zkrelu.cu:79-100): both bit sumchecks start from claim 0 with a vanishing c0, end in
    fa^2 - fa; partial_me values recombine to the magnitude / remainder tensors; the Hadamard sumcheck opens
    (Z o sign)~(u_hp)."""
    pr = ints(fr)
    sign, mag_bin, rem_bin = relu.aux()
    off = 0
    pme = {}
    for nb, lg, (u, v) in ((32, 5, (ch[0], ch[1])), (16, 4, (ch[2], ch[3]))):
        kk = k + lg
        ui, vi = ints(u), ints(v)
        claim = 0
        for r in range(kk):
            c = pr[off + 3 * r:off + 3 * r + 3]
            if r == 0:
                assert c[0] == 0
            assert ((1 - ui[r]) * poly(c, 0) + ui[r] * poly(c, 1)) % R == claim, ("bin round", nb, r)
            claim = poly(c, vi[r])
        fa = pr[off + 3 * kk]
        assert (fa * fa - fa) % R == claim
        assert fa == bm.fr_from_mont(zk.fr_me(mag_bin if nb == 32 else rem_bin, v))
        off += 3 * kk + 1
        pme[nb] = pr[off:off + nb]
        off += nb
    # bit i of every element, evaluated at u_rec: an independent path is partial_me over the one-Fr-per-bit tensor
    for nb, t in ((32, mag_bin), (16, rem_bin)):
        want = ints(zk.fr_partial_me(t, ch[4], nb).download())
        assert pme[nb] == want
    ui, vi = ints(ch[5]), ints(ch[6])
    zs = zk.fr_mul(Z, sign)
    claim = bm.fr_from_mont(zk.fr_me(zs, ch[5]))
    for r in range(k):
        c = pr[off + 3 * r:off + 3 * r + 3]
        assert ((1 - ui[r]) * poly(c, 0) + ui[r] * poly(c, 1)) % R == claim, ("hp round", r)
        claim = poly(c, vi[r])
    fa, fb = pr[off + 3 * k], pr[off + 3 * k + 1]
    assert fa * fb % R == claim
    assert fa == bm.fr_from_mont(zk.fr_me(Z, ch[6])) and fb == bm.fr_from_mont(zk.fr_me(sign, ch[6]))
    assert off + 3 * k + 2 == len(pr)


def split_transcript(wl, fr):
    """Fr transcript -> per-call slices in call order."""
    import bench
    sh, L = wl.shapes, len(wl.shapes)
    kb = bench.ceil_log2(wl.B_p)
    out, off = [], 0

    def fc_len(s):
        return 3 * bench.ceil_log2(s["in_p"]) + 4

    def relu_len(s):
        k = bench.ceil_log2(wl.B_p * s["out_p"])
        return 3 * (k + 5) + 1 + 32 + 3 * (k + 4) + 1 + 16 + 3 * k + 2

    order = [("fc", L - 1)]
    for i in range(L - 2, -1, -1):
        order += [("relu", i), ("fc", i)]
    for kind, i in order:
        n = fc_len(sh[i]) if kind == "fc" else relu_len(sh[i])
        out.append((kind, i, fr[off:off + n]))
        off += n
    assert off == fr.shape[0]
    return out, kb


def check_workload(zk, wl, relu_layers, fc_layers):
    import bench
    from zkdl_amd.capi import Transcript
    from zkdl_amd.dist import lanes_to_fr
    chal = wl.challenges(0)
    tr = Transcript(zk)
    wl.prove(chal, tr)
    fr, _ = tr.read()
    aff, inf = tr.read_affine()
    digest = tr.digest()
    pieces, kb = split_transcript(wl, fr)
    L = len(wl.shapes)
    by_call = {(kind, i): ch for kind, i, ch in chal}
    for kind, i, sl in pieces:
        s = wl.shapes[i]
        if kind == "fc" and i in fc_layers:
            x_in = wl.A[i - 1] if i > 0 else wl.X
            z_out = wl.Yhat if i == L - 1 else wl.Z[i]
            verify_fc(zk, sl, x_in, z_out, by_call[(kind, i)], bench.ceil_log2(s["in_p"]), z_is_unmont=(i == L - 1))
        elif kind == "relu" and i in relu_layers:
            verify_relu(zk, sl, wl.relus[i], wl.Z[i], by_call[(kind, i)], bench.ceil_log2(wl.B_p * s["out_p"]))
    # scheduling invariance at full size: immediate execution, no worker streams, sealed/pipelined
    for batch, lanes in ((0, 0), (1, 0)):
        zk.set_flag("batch_layers", batch)
        zk.set_flag("lanes", lanes)
        tr2 = Transcript(zk)
        wl.prove(chal, tr2)
        assert eq(tr2.digest(), digest), (batch, lanes)
        tr2.close()
    zk.set_flag("batch_layers", 1)
    zk.set_flag("lanes", 1)
    lanes8 = zk.empty(2, 8)
    wl.prove(chal, wl.trs[0])
    wl.trs[0].seal(lanes8.ptr)
    wl.prove(wl.challenges(1), wl.trs[1])                      # the next proof starts while the first one's G1 phase runs
    wl.trs[1].seal()
    fr3, _ = wl.trs[0].read()
    assert eq(fr3, fr)
    aff3, inf3 = wl.trs[0].read_affine()
    assert eq(aff3, aff) and eq(inf3, inf)
    l64 = lanes8.download().reshape(-1).view(np.uint64)
    assert lanes_to_fr([int(x) for x in l64]) == int.from_bytes(np.ascontiguousarray(digest).tobytes(), "little")
    tr.close()
    return fr, aff, inf


def g1_ints(aff, inf):
    return [bm.affine_mont_to_int(a, bool(i)) for a, i in zip(aff, inf)]


def check_opening_points(zk, wl, layer, seed):
    """Full-size G1 values, checked three ways for one layer of the workload (2048 generators, c = 10, 26 windows,
    buckets stitched across many chunks -- the regime that exists only at this size):
      1. zk_open under ZK_MSM_LITERAL -- the reference's op sequence (commitment.cu:43-92), whose Jacobian limbs are
         bit-identical to the oracle at small sizes -- against zk_open under ZK_MSM_PIPPENGER (bucket MSMs over the
         flattened opening): all 3k+2 points must be the same group elements (affine-canonical form);
      2. the me_open round relation of SURVEY.md App. E evaluated by the big-integer model on those points:
         T_{r+1} = u(1-u) T_r + (1-u)^2 T0_r + u^2 T1_r  (u = unmont(challenge)), because T_r is the commitment of the
         round's scalars under the round's generators; and at the end  T_last' = proof.back() * t(0);
      3. binding to the layer commitment: T_0 = [R mod r] * com(u_hi), com(u_hi) the first transcript point."""
    from zkdl_amd.capi import MSM_LITERAL, MSM_PIPPENGER
    rng = np.random.default_rng(seed)
    s = wl.shapes[layer]
    gens, fc = wl.gens[layer], wl.fcs[layer]
    import bench
    k = bench.ceil_log2(s["in_p"]) + bench.ceil_log2(s["out_p"])
    k_lo = bench.ceil_log2(s["n_gens"])
    u = np.zeros((k, 8), np.uint32)
    u[:] = rng.integers(0, 1 << 32, size=(k, 8), dtype=np.uint64).astype(np.uint32)
    u[:, 7] = (u[:, 7].astype(np.uint64) % 1944954707).astype(np.uint32)
    res = {}
    for mode in (MSM_LITERAL, MSM_PIPPENGER):
        zk.set_msm_mode(mode)
        ret, ce, proof = zk.open(gens, fc.weights(), fc.com(), u)
        pts = zk.upload(np.concatenate([ce.reshape(1, 36), proof]))
        aff, inf = zk.g1_to_affine(pts)
        res[mode] = (ret, aff, inf)
    zk.set_msm_mode(MSM_PIPPENGER)
    assert eq(res[MSM_LITERAL][0], res[MSM_PIPPENGER][0])
    assert eq(res[MSM_LITERAL][1], res[MSM_PIPPENGER][1]) and eq(res[MSM_LITERAL][2], res[MSM_PIPPENGER][2])
    ret, aff, inf = res[MSM_PIPPENGER]
    assert aff.shape[0] == 3 * k_lo + 2
    P = g1_ints(aff, inf)
    com_eval, proof = P[0], P[1:]
    uh = ints(u[:k_lo])                                 # generator-side challenges, un-Montgomery'd
    for r in range(k_lo):
        T, T0, T1 = proof[3 * r:3 * r + 3]
        nxt = bm.g1_add(bm.g1_add(bm.g1_mul(T, uh[r] * (1 - uh[r]) % R), bm.g1_mul(T0, (1 - uh[r]) ** 2 % R)), bm.g1_mul(T1, uh[r] ** 2 % R))
        want = proof[3 * (r + 1)] if r + 1 < k_lo else bm.g1_mul(proof[3 * k_lo], bm.limbs_to_int(ret))   # raw (Montgomery-form) limbs of t(0)
        assert nxt == want, ("me_open round", r)
    assert proof[0] == bm.g1_mul(com_eval, bm.RR % R), "T_0 must commit to the row-folded weights"
    # com(u_hi) itself: the commitment points' multilinear evaluation at the row challenges, against the literal G1_me
    assert com_eval == bm.jac_to_affine_int(zk.g1_me(fc.com(), u[k_lo:]))


def test_mlp18m_batch256_verifies(zk):
    """BASELINE configs[2]: every layer of the 18M-parameter MLP at batch 256."""
    import bench
    dims, batch = bench.WORKLOADS["mlp18m_b256"]
    wl = bench.Workload(zk, dims, batch, seed=5)
    L = len(wl.shapes)
    check_workload(zk, wl, relu_layers=set(range(L - 1)), fc_layers=set(range(L)))
    check_opening_points(zk, wl, layer=3, seed=77)      # 2048 x 2048 weights, 2048 generators, 2048 commitment rows
    check_opening_points(zk, wl, layer=0, seed=78)      # 1024 x 1024, 1024 generators


def test_mlp18m_batch1_verifies(zk):
    """BASELINE configs[1]: the same model at batch 1 (k_bs = 0: `partial_me` with no challenges is the identity,
    fr-tensor.cu:370-374; every activation table is one row)."""
    import bench
    dims, batch = bench.WORKLOADS["mlp18m_b1"]
    wl = bench.Workload(zk, dims, batch, seed=6)
    L = len(wl.shapes)
    check_workload(zk, wl, relu_layers=set(range(L - 1)), fc_layers=set(range(L)))
    check_opening_points(zk, wl, layer=L - 1, seed=79)  # 2048 x 1024, 2048 generators, 1024 commitment rows


def test_deep4096_slice_verifies_and_2p24_commit_is_linear(zk):
    """BASELINE configs[4]: width 4096, batch 256 (a 4-layer slice of the depth-64 model: the layers are
    identical in shape, `bench.py --workload deep4096x64_b256` runs all 64), plus the 2^24-pair commitment:
    linear in the tensor and independent of the Pippenger window width."""
    import bench
    wl = bench.Workload(zk, [4096] * 5, 256, seed=9)
    L = len(wl.shapes)
    check_workload(zk, wl, relu_layers={0, L - 2}, fc_layers={0, L - 1})
    gens, fc = wl.gens[1], wl.fcs[1]
    w = fc.weights()
    assert w.n == 1 << 24 and gens.n == 4096
    other = zk.fr_mont(zk.fr_random_int(w.n, 14))
    ca, cb = zk.commit(gens, w), zk.commit(gens, other)
    csum = zk.commit(gens, zk.fr_add(w, other))
    a1, i1 = zk.g1_to_affine(zk.g1_add(ca, cb))
    a2, i2 = zk.g1_to_affine(csum)
    assert eq(a1, a2) and eq(i1, i2)
    # the commitment made at layer construction (quantised weights: 7-bit windows) is the same row MSM
    a0, i0 = zk.g1_to_affine(fc.com())
    a3, i3 = zk.g1_to_affine(ca)
    assert eq(a0, a3) and eq(i0, i3)
    for window in (8, 10, 13):
        zk.set_msm_window(window)
        aw, iw = zk.g1_to_affine(zk.commit(gens, w))
        assert eq(aw, a3) and eq(iw, i3), window
    zk.set_msm_window(0)


def test_mlp18m_batch1_whole_proof_matches_oracle(zk, oracle):
    """BASELINE configs[0] == configs[1]: the WHOLE default model (model.py's 784-1000-1773x5-1124-1000, 18.2M parameters)
    at batch 1, proved by the CPU oracle (the restatement of the reference algorithm: its scalar path) and by the HIP path on
    the same quantised weights, generators, commitments and challenges: inference output, every Fr transcript value bit for bit,
    every G1 value in affine-canonical form.  (The commitments are inputs here -- the literal 26M-scalar-mul commit is hours on
    a CPU; commit parity is covered at small sizes and by linearity / window invariance at 2^24 pairs.)"""
    import bench
    from zkdl_amd.capi import Transcript
    dims, batch = bench.WORKLOADS["mlp18m_b1"]
    wl = bench.Workload(zk, dims, batch, seed=8)
    sh, L = wl.shapes, len(wl.shapes)
    chal = wl.challenges(0)
    tr = Transcript(zk)
    wl.prove(chal, tr)
    fr_g, _ = tr.read()
    aff_g, inf_g = tr.read_affine()
    # the oracle's run on the downloaded inputs
    W = [fc.weights().download() for fc in wl.fcs]
    gens = [g.points().download() for g in wl.gens]
    coms = [fc.com().download() for fc in wl.fcs]
    X = wl.X.download()
    Z, A, aux = [], [], []
    cur = X
    for i in range(L - 1):
        z = oracle.fc_forward(cur, W[i], 1, sh[i]["in_p"], sh[i]["out_p"])
        a, sign, mag, rem = oracle.relu_forward(z)
        assert eq(z, wl.Z[i].download()) and eq(a, wl.A[i].download())
        Z.append(z); A.append(a); aux.append((sign, mag, rem))
        cur = a
    Yhat = oracle.fr_unmont(oracle.fc_forward(cur, W[L - 1], 1, sh[L - 1]["in_p"], sh[L - 1]["out_p"]))
    assert eq(Yhat, wl.Yhat.download())
    frs, g1s = [], []
    for kind, i, ch in chal:
        if kind == "fc":
            x_in = A[i - 1] if i > 0 else X
            z_out = Yhat if i == L - 1 else Z[i]
            f, g = oracle.fc_prove(x_in, z_out, W[i], sh[i]["in_p"], sh[i]["out_p"], gens[i], coms[i], *ch)
            frs.append(f); g1s.append(g)
        else:
            frs.append(oracle.relu_prove(Z[i], A[i], *aux[i], *ch))
    fr_o, g1_o = np.concatenate(frs), np.concatenate(g1s)
    assert eq(fr_g, fr_o)
    aff_o, inf_o = oracle.g1_to_affine(g1_o)
    assert eq(aff_g, aff_o) and eq(inf_g, inf_o)


def test_deep4096x64_at_full_depth_with_oracle_parity_at_width_4096(zk):
    """BASELINE configs[4] at its stated size: width 4096, depth 64, batch 256 -- all 64 zkFC::prove and 63 zkReLU::prove calls in one
    transcript (4096 generators per layer, 12 me_open rounds: demo.cu:81, zkfc.cu:92-93).  Every layer's Fr recurrences are verified
    (check_workload: sumcheck round consistency and final claims of all 64 zkFC slices, boolean vanishing and recombination of three zkReLU slices,
    scheduling invariance and the sealed / pipelined form of the whole transcript), and
    the CPU oracle -- the restatement of the reference algorithm -- reproduces the last layer's zkFC (2^24 weights, 4096 generators) and the
    last hidden layer's zkReLU + zkFC at batch 256 bit for bit (Fr) / in affine-canonical form (G1), from the GPU's own tensors and challenges.
    The per-proof time of the same loop is what `bench.py --workload deep4096x64_b256` reports (profiles/r04_deep4096x64.json)."""
    import json
    import time
    import bench
    from zkdl_amd.capi import Transcript
    dims, batch = bench.WORKLOADS["deep4096x64_b256"]
    t0 = time.perf_counter()
    wl = bench.Workload(zk, dims, batch, seed=12)
    zk.sync()
    t_setup = time.perf_counter() - t0
    L = len(wl.shapes)
    assert L == 64 and all(s["n_gens"] == 4096 and s["m"] == 4096 for s in wl.shapes)
    # every zkFC slice; the zkReLU slices of the first, a middle and the last hidden layer (each materialises that layer's 1.5 GB of
    # one-field-element-per-bit tensors for the check; the other 60 are covered by the whole-transcript invariance checks below)
    fr, aff, inf = check_workload(zk, wl, relu_layers={0, L // 2, L - 2}, fc_layers=set(range(L)))
    assert fr.shape[0] == sum(3 * 12 + 4 for _ in range(L)) + (L - 1) * (3 * 25 + 1 + 32 + 3 * 24 + 1 + 16 + 3 * 20 + 2) and aff.shape[0] == L * (3 * 12 + 2)
    # timing of the synchronous loop (the reference's own definition of its metric, demo.cu:124-140)
    tr = Transcript(zk)
    wl.prove(wl.challenges(0), tr)
    tr.flush()
    zk.sync()
    t0 = time.perf_counter()
    for st in range(3):
        wl.prove(wl.challenges(st), tr)
        tr.flush()
    zk.sync()
    ms = (time.perf_counter() - t0) * 1e3 / 3
    assert ms / batch < 100.0
    tr.close()
    # oracle parity at width 4096, batch 256: zkFC 63 (opens Y_hat), zkReLU 62, zkFC 62
    calls = [("fc", L - 1), ("relu", L - 2), ("fc", L - 2)]
    t0 = time.perf_counter()
    _, checked = bench.cpu_baseline_check_only(zk, wl, calls, (wl.challenges(0), fr, aff, inf))
    t_oracle = time.perf_counter() - t0
    assert checked["ok"], checked
    assert checked["calls"] == ["zkFC63", "zkFC62", "zkReLU62"]
    assert checked["g1"] == 2 * (3 * 12 + 2) and checked["fr"] == 2 * (3 * 12 + 4) + (3 * 25 + 1 + 32 + 3 * 24 + 1 + 16 + 3 * 20 + 2)
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "deep4096x64_test.json"), "w") as f:
            json.dump({"workload": "deep4096x64_b256", "dims": "4096 x 65", "batch": batch, "setup_s": t_setup, "synchronous_ms_per_proof": ms,
                       "ms_per_example": ms / batch, "transcript": [int(fr.shape[0]), int(aff.shape[0])], "layers_verified": {"zkFC": L, "zkReLU": 3},
                       "parity_checked": checked, "oracle_s": t_oracle}, f, indent=1)
