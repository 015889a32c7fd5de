"""Proof objects above the C ABI: the statement digest, hash-derived challenges for every prove call of the reference's
timed loop (demo.cu:124-138), the serialised transcript split back into its calls, and the verifier loop.

The verifier (`verify_mlp`) reads only public data -- generators, the layers' commitments, the network's input and
output tensors, the challenges, the serialised proof -- and calls zk_verify_fc / zk_verify_relu (csrc/verify.hip) per
call.  What the reference's protocol binds, it checks; what it leaves unbound (the ReLU layers' final evaluations and
the activation-side claims between layers: "commitment of activations" is the reference's stated future work) comes
back in `open_claims`.
"""
import struct

import numpy as np

from .capi import fs_derive, fs_round_challenges, sha256

KIND_FC, KIND_RELU = 0, 1


def ceil_log2(x):
    return 0 if x <= 1 else (x - 1).bit_length()


def layer_shapes(dims):
    out = []
    for n_in, n_out in zip(dims[:-1], dims[1:]):
        in_p, out_p = 1 << ceil_log2(n_in), 1 << ceil_log2(n_out)
        n_g = 1 << ((ceil_log2(n_in * n_out) + 1) // 2)             # demo.cu:81 (unpadded in*out)
        out.append(dict(n_in=n_in, n_out=n_out, in_p=in_p, out_p=out_p, n_gens=n_g, m=in_p * out_p // n_g))
    return out


def call_order(n_layers):
    """(kind, layer) in the order demo.cu:128-137 proves them."""
    out = [("fc", n_layers - 1)]
    for i in range(n_layers - 2, -1, -1):
        out += [("relu", i), ("fc", i)]
    return out


def statement_digest(zk, dims, batch, gens, coms, x_fr, y_fr):
    """SHA-256 over everything public: shape, generators and commitments (affine-canonical bytes + infinity flags),
    the quantised input X and the claimed output Y_hat (Fr limbs as stored)."""
    parts = [b"zkdl-amd/statement/v1", struct.pack("<I", len(dims)), struct.pack("<%dI" % len(dims), *dims), struct.pack("<I", batch)]
    for g, c in zip(gens, coms):
        for pts in (g.points(), c):
            aff, inf = zk.g1_to_affine(pts)
            parts += [sha256(aff.tobytes()), sha256(inf.tobytes())]
    parts += [sha256(x_fr.download().tobytes()), sha256(y_fr.download().tobytes())]
    return sha256(b"".join(parts))


def fs_challenges(seed, dims, batch):
    """Challenge vectors of every prove call, in call order, each derived from (seed, layer, kind, position in the
    reference's draw order: zkfc.cu:135-137 u_bs,u_in,u_out / zkrelu.cu:85-98 u_z,v_z,u_r,v_r,u_rec,u_hp,v_hp)."""
    sh = layer_shapes(dims)
    kb = ceil_log2(1 << ceil_log2(batch))
    out = []
    for kind, i in call_order(len(sh)):
        s = sh[i]
        if kind == "fc":
            lens = (kb, ceil_log2(s["in_p"]), ceil_log2(s["out_p"]))
            out.append(("fc", i, [fs_derive(seed, i, KIND_FC, v, n) for v, n in enumerate(lens)]))
        else:
            k = kb + ceil_log2(s["out_p"])
            lens = (k + 5, k + 5, k + 4, k + 4, k, k, k)
            out.append(("relu", i, [fs_derive(seed, i, KIND_RELU, v, n) for v, n in enumerate(lens)]))
    return out


# ------------------------------------------------------------------ round-by-round Fiat-Shamir ("sound mode": zk_fc_prove_fs / zk_relu_prove_fs)
def prove_mlp_fs(fcs, relus, gens, X, Z, A, Yhat, seed, tr):
    """The reference's timed loop (demo.cu:124-138) with every challenge derived by the library: evaluation points from the statement digest
    `seed`, every sumcheck round challenge from a hash of the rounds before it (csrc/fs_rounds.hip)."""
    L = len(fcs)
    tr.reset()
    fcs[L - 1].prove_fs(A[L - 2] if L > 1 else X, Yhat, gens[L - 1], seed, L - 1, tr)
    for i in range(L - 2, -1, -1):
        relus[i].prove_fs(Z[i], A[i], seed, i, tr)
        fcs[i].prove_fs(A[i - 1] if i > 0 else X, Z[i], gens[i], seed, i, tr)


def fs_round_mode_challenges(seed, dims, batch, fr):
    """Every challenge vector of a proof made by prove_mlp_fs, re-derived from the statement digest and the proof's own Fr values `fr`
    (the verifier's derivation): the list has fs_challenges()'s shape, so verify_mlp -- and the oracle -- take it as they take any other."""
    sh = layer_shapes(dims)
    kb = ceil_log2(1 << ceil_log2(batch))
    out, off = [], 0
    for kind, i in call_order(len(sh)):
        s = sh[i]
        if kind == "fc":
            k_in, k_out = ceil_log2(s["in_p"]), ceil_log2(s["out_p"])
            u_in = fs_round_challenges(seed, i, KIND_FC, 0, fr[off:off + 3 * k_in])
            out.append(("fc", i, [fs_derive(seed, i, KIND_FC, 0, kb), u_in, fs_derive(seed, i, KIND_FC, 2, k_out)]))
            off += 3 * k_in + 4
        else:
            k = kb + ceil_log2(s["out_p"])
            o_mag, o_rem = off, off + 3 * (k + 5) + 1 + 32
            o_hp = o_rem + 3 * (k + 4) + 1 + 16
            v_z = fs_round_challenges(seed, i, KIND_RELU, 1, fr[o_mag:o_mag + 3 * (k + 5)])
            v_r = fs_round_challenges(seed, i, KIND_RELU, 2, fr[o_rem:o_rem + 3 * (k + 4)])
            v_hp = fs_round_challenges(seed, i, KIND_RELU, 3, fr[o_hp:o_hp + 3 * k])
            out.append(("relu", i, [fs_derive(seed, i, KIND_RELU, 0, k + 5), v_z, fs_derive(seed, i, KIND_RELU, 2, k + 4), v_r,
                                    fs_derive(seed, i, KIND_RELU, 4, k), fs_derive(seed, i, KIND_RELU, 5, k), v_hp]))
            off = o_hp + 3 * k + 2
    if off != fr.shape[0]:
        raise ValueError("transcript does not have the shape of this model's proof")
    return out


def verify_mlp_fs(zk, dims, batch, gens, coms, x_fr, y_fr, blob):
    """Verify a proof made by prove_mlp_fs: the challenges are re-derived from the statement AND the proof's round messages, then every
    check of verify_mlp runs.  A changed round message changes every later challenge of its sumcheck, so the prover cannot have chosen the
    message knowing them.  Returns (ok, message, open_claims)."""
    from .capi import ZkError
    try:
        fr, _, _ = parse_transcript(blob)
        chal = fs_round_mode_challenges(statement_digest(zk, dims, batch, gens, coms, x_fr, y_fr), dims, batch, fr)
    except ZkError:                      # zk_fs_round_challenges: a round message that is not a canonical field element (>= r)
        return False, "a round message is not a canonical field element", []
    except ValueError as e:
        return False, str(e), []
    return verify_mlp(zk, dims, batch, gens, coms, x_fr, y_fr, blob, chal=chal)


def parse_transcript(blob):
    """zk_transcript_serialize's format -> (Fr (n, 8), affine G1 (n, 24), infinity flags (n,))."""
    if blob[:8] != b"ZKDLTR01":
        raise ValueError("not a zkdl_amd transcript")
    n_fr, n_g1 = struct.unpack("<QQ", blob[8:24])
    need = 24 + 32 * n_fr + 97 * n_g1
    if len(blob) != need:
        raise ValueError("transcript length %d does not match its header (%d)" % (len(blob), need))
    off = 24
    fr = np.frombuffer(blob, dtype=np.uint32, count=8 * n_fr, offset=off).reshape(-1, 8)
    off += 32 * n_fr
    aff = np.frombuffer(blob, dtype=np.uint32, count=24 * n_g1, offset=off).reshape(-1, 24)
    off += 96 * n_g1
    inf = np.frombuffer(blob, dtype=np.uint8, count=n_g1, offset=off)
    return fr, aff, inf


def split_calls(dims, batch, fr, aff, inf):
    """Per-call slices of a whole-MLP transcript (SURVEY.md App. A.6 order)."""
    sh = layer_shapes(dims)
    kb = ceil_log2(1 << ceil_log2(batch))
    out, of, og = [], 0, 0
    for kind, i in call_order(len(sh)):
        s = sh[i]
        if kind == "fc":
            nf, ng = 3 * ceil_log2(s["in_p"]) + 4, 3 * ceil_log2(s["n_gens"]) + 2
        else:
            k = kb + ceil_log2(s["out_p"])
            nf, ng = 3 * (k + 5) + 1 + 32 + 3 * (k + 4) + 1 + 16 + 3 * k + 2, 0
        out.append((kind, i, fr[of:of + nf], aff[og:og + ng], inf[og:og + ng]))
        of += nf
        og += ng
    if of != fr.shape[0] or og != aff.shape[0]:
        raise ValueError("transcript does not have the shape of this model's proof")
    return out


def verify_mlp(zk, dims, batch, gens, coms, x_fr, y_fr, blob, chal=None):
    """Verify a serialised whole-MLP proof.  gens[i]: the layer's Gens handle; coms[i]: DevArray of its commitment
    points; x_fr / y_fr: the public quantised input (Montgomery form, padded) and output Y_hat (un-Montgomery'd, as
    demo.out holds it).  chal: the challenges (default: derived from the statement digest).  Returns
    (ok, message, open_claims) -- open_claims lists the evaluations the reference's protocol leaves unbound."""
    sh = layer_shapes(dims)
    L = len(sh)
    B_p = 1 << ceil_log2(batch)
    if chal is None:
        chal = fs_challenges(statement_digest(zk, dims, batch, gens, coms, x_fr, y_fr), dims, batch)
    try:
        fr, aff, inf = parse_transcript(blob)
        calls = split_calls(dims, batch, fr, aff, inf)
    except ValueError as e:
        return False, str(e), []
    open_claims = []
    for (kind, i, f, a, nf), (ckind, ci, ch) in zip(calls, chal):
        assert (kind, i) == (ckind, ci)
        s = sh[i]
        if kind == "fc":
            # the first call opens the public output, the last one the public input
            expect_z = zk.fr_me(y_fr, np.concatenate([ch[2], ch[0]])) if i == L - 1 else None
            expect_x = zk.fr_me(x_fr, np.concatenate([ch[1], ch[0]])) if i == 0 else None
            ok, msg, cz, cx = zk.verify_fc(gens[i], coms[i], B_p, s["in_p"], s["out_p"], ch, f, a, nf, z_is_raw=(i == L - 1), expect_z=expect_z, expect_x=expect_x)
            if not ok:
                return False, "zkFC layer %d: %s" % (i, msg), open_claims
            if i != L - 1:
                open_claims.append(("Z_%d~(u_out||u_bs)" % i, cz))
            if i != 0:
                open_claims.append(("A_%d~(u_in||u_bs)" % (i - 1), cx))
        else:
            ok, msg, claims = zk.verify_relu(B_p * s["out_p"], ch, f)
            if not ok:
                return False, "zkReLU layer %d: %s" % (i, msg), open_claims
            open_claims += [("relu_%d:%s" % (i, nm), c) for nm, c in zip(("mag_bin~(v_z)", "rem_bin~(v_r)", "(Z.sign)~(u_hp)", "Z~(v_hp)", "sign~(v_hp)"), claims)]
    return True, "", open_claims


# ------------------------------------------------------------------ activation commitments (the reference's stated future work)
# The layer proofs leave the activation-side evaluations as open claims (verify_mlp returns them): nothing in the reference's protocol
# ties Z_i~(u), A_i~(u) to the tensors.  With the activations COMMITTED (the same row-wise Pedersen commitment the weights use:
# zk_commit over the layer's generators) every such claim is closed by an opening (zk_open) that zk_verify_open checks against the
# commitment -- and because FC_i's output claim, ReLU_i's input claim and FC_{i+1}'s input claim open the SAME commitments, the layers
# are chained.  What stays open: the ReLU's bit / sign tensors (the reference's ReLU relation between Z, sign, the bits and A is itself
# incomplete).  This is optional protocol work on top of the reference's proof: ~2^19-pair commitments and three extra openings per layer.
def activation_claims(dims, batch, chal):
    """[(tensor name, point u)] for every claim on a committed activation: FC_i opens Z_i at u_out||u_bs and A_{i-1} at u_in||u_bs,
    ReLU_i's Hadamard sumcheck ends in Z_i at v_hp."""
    L = len(dims) - 1
    out = []
    for kind, i, ch in chal:
        if kind == "fc":
            if i != L - 1:
                out.append(("Z%d" % i, np.concatenate([ch[2], ch[0]])))
            if i != 0:
                out.append(("A%d" % (i - 1), np.concatenate([ch[1], ch[0]])))
        else:
            out.append(("Z%d" % i, ch[6]))
    return out


def commit_activations(zk, gens, Z, A):
    """Commit every pre-activation Z_i and activation A_i (i < L-1) over layer i's generators.  Returns {name: (gens, tensor, commitment)}."""
    out = {}
    for i, (z, a) in enumerate(zip(Z, A)):
        g = gens[i]
        out["Z%d" % i] = (g, z, zk.commit(g, z))
        out["A%d" % i] = (g, a, zk.commit(g, a))
    return out


def open_activations(zk, committed, claims):
    """One zk_open per claim: [(name, value, affine points, infinity flags)]."""
    out = []
    for name, u in claims:
        g, t, com = committed[name]
        ret, com_eval, proof = zk.open(g, t, com, u)
        aff, inf = zk.g1_to_affine(zk.upload(np.concatenate([com_eval.reshape(1, 36), proof])))
        out.append((name, ret, aff, inf))
    return out


def verify_activation_openings(zk, commitments, claims, openings, claimed_values):
    """commitments: {name: (gens, commitment points)} -- public.  claimed_values[j]: the evaluation the layer proofs claim for claims[j]
    (from verify_mlp's open_claims).  Every opening must verify against its commitment and open exactly the claimed value."""
    if len(openings) != len(claims) or len(claimed_values) != len(claims):
        return False, "wrong number of openings"
    for (name, u), (oname, value, aff, inf), want in zip(claims, openings, claimed_values):
        if oname != name:
            return False, "opening %s where %s was expected" % (oname, name)
        if not np.array_equal(np.asarray(value).reshape(8), np.asarray(want).reshape(8)):
            return False, "%s: the opened value differs from the layer proof's claim" % name
        g, com = commitments[name]
        ok, msg = zk.verify_open(g, com, u, value, aff, inf)
        if not ok:
            return False, "%s: %s" % (name, msg)
    return True, ""


def claimed_activation_values(dims, open_claims):
    """verify_mlp's open_claims -> the values in activation_claims() order."""
    L = len(dims) - 1
    by = {k: v for k, v in open_claims}
    out = []
    for kind, i in call_order(L):
        if kind == "fc":
            if i != L - 1:
                out.append(by["Z_%d~(u_out||u_bs)" % i])
            if i != 0:
                out.append(by["A_%d~(u_in||u_bs)" % (i - 1)])
        else:
            out.append(by["relu_%d:Z~(v_hp)" % i])
    return out


# ------------------------------------------------------------------ activation openings INSIDE the proof, one per hidden layer (csrc/act_open.hip)
KIND_ACT = 2


def act_statement_seed(zk, seed, act_coms):
    """The seed every challenge of the activation openings comes from: the statement digest AND the activation commitments [com(Z_i) ; com(A_i)]
    (affine-canonical bytes + infinity flags).  A prover therefore has to fix com(Z_i), com(A_i) before it learns alpha or the opening point
    (derived from the statement alone, the challenges were known before the commitments were chosen)."""
    parts = [b"zkdl-amd/act/v1", bytes(seed)]
    for c in act_coms:
        aff, inf = zk.g1_to_affine(c)
        parts += [sha256(aff.tobytes()), sha256(inf.tobytes())]
    return sha256(b"".join(parts))


def act_layer_seed(aseed, layer, ys, pts):
    """Sound mode, per hidden layer: the activation seed, the layer, the three claimed evaluations as the layer proofs state them and the three
    points they are claimed at -- everything alpha may depend on (zk_act_open_fs)."""
    parts = [b"zkdl-amd/act-layer/v1", bytes(aseed), struct.pack("<I", layer)]
    parts += [np.ascontiguousarray(np.asarray(y, dtype=np.uint32).reshape(8)).tobytes() for y in ys]
    parts += [np.ascontiguousarray(np.asarray(p, dtype=np.uint32).reshape(-1, 8)).tobytes() for p in pts]
    return sha256(b"".join(parts))


def act_open_challenges(aseed, dims, batch):
    """Per hidden layer i: (alpha, fold) -- the combination challenge of its three claims and the k + 1 round challenges of the claim-reduction
    sumcheck (= the opening point), derived from act_statement_seed() like every other challenge of the throughput mode: a public-coin schedule
    fixed by the statement and the commitments, NOT a function of the prover's round messages (the sound variant is prove_activation_openings_fs)."""
    sh = layer_shapes(dims)
    kb = ceil_log2(1 << ceil_log2(batch))
    out = []
    for i in range(len(sh) - 1):
        k = kb + ceil_log2(sh[i]["out_p"])
        out.append((fs_derive(aseed, i, KIND_ACT, 0, 1), fs_derive(aseed, i, KIND_ACT, 1, k + 1)))
    return out


def act_claim_points(dims, chal):
    """Per hidden layer i: (p1, p2, p3) -- Z_i is claimed at p1 = u_out || u_bs by zkFC_i and at p2 = v_hp by zkReLU_i, A_i at p3 = u_in || u_bs by zkFC_{i+1}."""
    L = len(dims) - 1
    by = {(kind, i): ch for kind, i, ch in chal}
    out = []
    for i in range(L - 1):
        fc, relu, nxt = by[("fc", i)], by[("relu", i)], by[("fc", i + 1)]
        out.append((np.concatenate([fc[2], fc[0]]), relu[6], np.concatenate([nxt[1], nxt[0]])))
    return out


def act_claimed_values(dims, batch, fr):
    """Per hidden layer i: (y1, y2, y3) = Z_i~(p1), Z_i~(p2), A_i~(p3) read from the layer proofs' Fr values (SURVEY App. A.6 layout): zkFC's slice
    ends [.. a(0) = X~, b(0) = W~][Z~(u_out||u_bs)][t(0)], the ReLU's Hadamard sumcheck ends [.. (Z)~(v_hp), sign~(v_hp)]."""
    sh = layer_shapes(dims)
    kb = ceil_log2(1 << ceil_log2(batch))
    L = len(sh)
    z1, z2, a3 = {}, {}, {}
    off = 0
    for kind, i in call_order(L):
        s = sh[i]
        if kind == "fc":
            k_in = ceil_log2(s["in_p"])
            z1[i] = fr[off + 3 * k_in + 2]
            if i > 0:
                a3[i - 1] = fr[off + 3 * k_in]
            off += 3 * k_in + 4
        else:
            k = kb + ceil_log2(s["out_p"])
            nf = 3 * (k + 5) + 1 + 32 + 3 * (k + 4) + 1 + 16 + 3 * k + 2
            z2[i] = fr[off + nf - 2]
            off += nf
    return [(z1[i], z2[i], a3[i]) for i in range(L - 1)]


def prove_activation_openings(zk, dims, batch, gens, Z, A, chal, aseed, tr):
    """Behind the layer proofs already recorded on `tr`: one claim-reduction sumcheck and ONE opening of [Z_i ; A_i] per hidden layer
    (zk_act_open); their MSMs join the transcript's deferred batch.  aseed = act_statement_seed(zk, statement digest, act_coms)."""
    pts = act_claim_points(dims, chal)
    for i, (alpha, fold) in enumerate(act_open_challenges(aseed, dims, batch)):
        zk.act_open(gens[i], Z[i], A[i], pts[i], alpha, fold, tr)


def prove_activation_openings_fs(zk, dims, batch, gens, Z, A, seed, aseed, tr):
    """The sound variant, behind prove_mlp_fs(.., tr): the layer proofs are finished and read back (their hashed round challenges are the claim
    points p2, p3, and alpha has to depend on the claimed values), then every layer's reduction runs with alpha and its round challenges hashed
    from act_layer_seed() and its own round messages on the device (zk_act_open_fs).  One host read per proof: an extension, not the headline loop."""
    tr.flush()
    fr, _ = tr.read()
    chal = fs_round_mode_challenges(seed, dims, batch, fr)
    pts = act_claim_points(dims, chal)
    ys = act_claimed_values(dims, batch, fr)
    for i in range(len(dims) - 2):
        zk.act_open_fs(gens[i], Z[i], A[i], pts[i], act_layer_seed(aseed, i, ys[i], pts[i]), i, tr)


def act_open_lengths(dims, batch):
    """(n_fr, n_g1) of every layer's zk_act_open slice"""
    sh = layer_shapes(dims)
    kb = ceil_log2(1 << ceil_log2(batch))
    return [(3 * (kb + ceil_log2(s["out_p"]) + 1) + 3, 3 * ceil_log2(s["n_gens"]) + 2) for s in sh[:-1]]


def verify_mlp_with_activation_openings(zk, dims, batch, gens, coms, act_coms, x_fr, y_fr, blob, chal=None, sound=False):
    """Verify the layer proofs AND the activation openings behind them.  act_coms[i] = DevArray of the 2m points [com(Z_i) ; com(A_i)] (public, like
    the weight commitments).  What verify_mlp returns as open claims on Z_i / A_i is now bound to those commitments; what stays open is what the
    reference's ReLU leaves unbound (bit tensors, sign).  Returns (ok, message, remaining open claims).
    sound=False: every challenge derived from the statement (and, for the openings, the activation commitments) -- a public-coin schedule, as in
    the reference, whose coins are drawn before any round runs; a consistency check, not a sound non-interactive argument.
    sound=True (a proof made by prove_mlp_fs + prove_activation_openings_fs): the layer proofs' round challenges, alpha and the reductions' round
    challenges are all re-derived from the proof's own messages."""
    from .capi import ZkError
    sh = layer_shapes(dims)
    L = len(sh)
    seed = statement_digest(zk, dims, batch, gens, coms, x_fr, y_fr)
    aseed = act_statement_seed(zk, seed, act_coms)
    try:
        fr, aff, inf = parse_transcript(blob)
    except ValueError as e:
        return False, str(e), []
    lens = act_open_lengths(dims, batch)
    nf_act, ng_act = sum(a for a, _ in lens), sum(b for _, b in lens)
    if fr.shape[0] < nf_act or aff.shape[0] < ng_act:
        return False, "transcript does not have the shape of this model's proof", []
    nf0, ng0 = fr.shape[0] - nf_act, aff.shape[0] - ng_act
    try:
        if sound:
            chal = fs_round_mode_challenges(seed, dims, batch, fr[:nf0])
        elif chal is None:
            chal = fs_challenges(seed, dims, batch)
    except ZkError:
        return False, "a round message is not a canonical field element", []
    except ValueError as e:
        return False, str(e), []
    head = b"ZKDLTR01" + struct.pack("<QQ", nf0, ng0) + fr[:nf0].tobytes() + aff[:ng0].tobytes() + inf[:ng0].tobytes()
    ok, msg, open_claims = verify_mlp(zk, dims, batch, gens, coms, x_fr, y_fr, head, chal=chal)
    if not ok:
        return False, msg, open_claims
    by = {k: v for k, v in open_claims}
    pts = act_claim_points(dims, chal)
    kb = ceil_log2(1 << ceil_log2(batch))
    of, og = nf0, ng0
    closed = set()
    static = None if sound else act_open_challenges(aseed, dims, batch)
    for i in range(L - 1):
        nf, ng = lens[i]
        ys = [by["Z_%d~(u_out||u_bs)" % i], by["relu_%d:Z~(v_hp)" % i], by["A_%d~(u_in||u_bs)" % i]]
        k = kb + ceil_log2(sh[i]["out_p"])
        if sound:
            lseed = act_layer_seed(aseed, i, ys, pts[i])
            alpha = fs_derive(lseed, i, KIND_ACT, 0, 1)
            try:
                fold = fs_round_challenges(lseed, i, KIND_ACT, 0, fr[of:of + 3 * (k + 1)])
            except ZkError:
                return False, "activation opening, layer %d: a round message is not a canonical field element" % i, open_claims
        else:
            alpha, fold = static[i]
        ok, msg = zk.verify_act_open(gens[i], act_coms[i], k, pts[i], alpha, fold, ys, fr[of:of + nf], aff[og:og + ng], inf[og:og + ng])
        if not ok:
            return False, "activation opening, layer %d: %s" % (i, msg), open_claims
        closed |= {"Z_%d~(u_out||u_bs)" % i, "relu_%d:Z~(v_hp)" % i, "A_%d~(u_in||u_bs)" % i}
        of += nf
        og += ng
    return True, "", [(k, v) for k, v in open_claims if k not in closed]


# ------------------------------------------------------------------ activation commitments made INSIDE the proving pipeline
def act_commit_counts(dims, batch):
    """Points per hidden layer of [com(Z_i) ; com(A_i)]: 2 m_i, m_i = B_p out_p / n_gens"""
    sh = layer_shapes(dims)
    B_p = 1 << ceil_log2(batch)
    return [2 * (B_p * s["out_p"] // s["n_gens"]) for s in sh[:-1]]


def act_seed_from_points(seed, dims, batch, aff, inf):
    """act_statement_seed() from the affine-canonical commitment points as a transcript holds them (all hidden layers, [com(Z_i) ; com(A_i)] in order)"""
    parts = [b"zkdl-amd/act/v1", bytes(seed)]
    og = 0
    for n in act_commit_counts(dims, batch):
        parts += [sha256(np.ascontiguousarray(aff[og:og + n]).tobytes()), sha256(np.ascontiguousarray(inf[og:og + n]).tobytes())]
        og += n
    if og != aff.shape[0]:
        raise ValueError("commitment transcript does not have the shape of this model")
    return sha256(b"".join(parts))


def prove_with_activation_commitments_stream(zk, dims, batch, gens, fcs, relus, X, Z, A, Yhat, seed, count, depth=2, max_bits=48):
    """`count` proofs of one statement, each WITH its activation commitments, software-pipelined in two sealed stages per proof:
        A: com(Z_i), com(A_i) of every hidden layer as deferred row MSMs (zk_act_commit) on their own transcript -- read back when they are done, because
           the openings' challenges must depend on them (act_statement_seed);
        B: the layer proofs (statement-derived challenges) and the activation openings (challenges from the seed that absorbed stage A's points).
    While proof k's stage A is read, proof k + 1's stage A and proof k - 1's stage B are on the device: the commitments cost their MSM work
    (7 M short-scalar products per 18M / batch-256 proof) but no idle time.  max_bits: the bound passed to zk_act_commit (zkReLU's inputs are < 2^47 in
    magnitude -- zk_relu_forward refuses anything else -- and its outputs < 2^32; 0 = full width).  Yields (commitment points affine, their infinity flags, serialised stage-B proof)."""
    from .capi import Transcript
    chal = fs_challenges(seed, dims, batch)
    L = len(dims) - 1
    inflight = 2 * depth + 1          # `depth` proofs between a stage's submission and its read
    trsA = [Transcript(zk) for _ in range(inflight)]
    trsB = [Transcript(zk) for _ in range(inflight)]
    com = {}
    try:
        for k in range(count + 2 * depth):
            if k < count:
                t = trsA[k % inflight]
                t.reset()
                for i in range(L - 1):
                    zk.act_commit(gens[i], Z[i], t, max_bits)
                    zk.act_commit(gens[i], A[i], t, max_bits)
                t.seal()
            j = k - depth
            if 0 <= j < count:
                _, aff, inf = trsA[j % inflight].read_sealed()
                com[j] = (aff, inf)
                aseed = act_seed_from_points(seed, dims, batch, aff, inf)
                t = trsB[j % inflight]
                t.reset()
                fcs[L - 1].prove(A[L - 2] if L > 1 else X, Yhat, gens[L - 1], *[c for c in chal[0][2]], t)
                for (kind, i, ch) in chal[1:]:
                    if kind == "relu":
                        relus[i].prove(Z[i], A[i], ch, t)
                    else:
                        fcs[i].prove(A[i - 1] if i > 0 else X, Z[i], gens[i], ch[0], ch[1], ch[2], t)
                prove_activation_openings(zk, dims, batch, gens, Z, A, chal, aseed, t)
                t.seal()
            j = k - 2 * depth
            if 0 <= j < count:
                aff, inf = com.pop(j)
                yield aff, inf, trsB[j % inflight].serialize_sealed()
    finally:
        for t in trsA + trsB:
            t.close()


def act_coms_from_points(zk, dims, batch, aff, inf):
    """Device arrays [com(Z_i) ; com(A_i)] (Jacobian, Z = 1) per hidden layer from the affine points of a stage-A transcript: what
    verify_mlp_with_activation_openings takes as act_coms."""
    p_mod = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    G1_ONE_MONT = np.frombuffer(((1 << 384) % p_mod).to_bytes(48, "little"), dtype=np.uint32)     # 1 in the library's Montgomery form
    out, og = [], 0
    for n in act_commit_counts(dims, batch):
        a, f = aff[og:og + n], inf[og:og + n]
        jac = np.zeros((n, 36), dtype=np.uint32)
        jac[:, :24] = a
        jac[:, 24:] = np.asarray(G1_ONE_MONT, dtype=np.uint32)
        jac[f != 0] = 0
        jac[f != 0, 12:24] = np.asarray(G1_ONE_MONT, dtype=np.uint32)      # infinity as {0, ONE, 0}
        out.append(zk.upload(jac))
        og += n
    return out
