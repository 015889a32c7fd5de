"""Synthetic quantised MLPs and the demo.cu proving loop, run once through the oracle and once
through the HIP library (C ABI), on the same seeded weights / inputs / generators / challenges.

Shapes and order follow the reference driver: padded power-of-two dims (zkfc.cu:92-93,108-109),
n_gens from the UNPADDED in*out (demo.cu:81), inference then reverse-order proving with Y_hat
un-Montgomery'd in place (demo.cu:119-137), challenge draw order of zkfc.cu:135-137 /
zkrelu.cu:85-98."""
import numpy as np

from util import rand_fr


def ceil_log2(x):
    return 0 if x <= 1 else (x - 1).bit_length()


def layer_shapes(dims):
    out = []
    for n_in, n_out in zip(dims[:-1], dims[1:]):
        in_p, out_p = 1 << ceil_log2(n_in), 1 << ceil_log2(n_out)
        n_g = 1 << ((ceil_log2(n_in * n_out) + 1) // 2)
        out.append(dict(n_in=n_in, n_out=n_out, in_p=in_p, out_p=out_p, n_gens=n_g, m=in_p * out_p // n_g))
    return out


def synth_mlp(orc, dims, batch, seed=0, gens=None):
    """Random-init bias-free MLP in model.py's distributions: W ~ U(-1/sqrt(in), 1/sqrt(in)) stored
    [in, out] (= weight.t(), demo.cu:72), X ~ N(0, 1)."""
    rng = np.random.default_rng(seed)
    shapes = layer_shapes(dims)
    W = [rng.uniform(-1.0 / np.sqrt(s["n_in"]), 1.0 / np.sqrt(s["n_in"]), size=(s["n_in"], s["n_out"])).astype(np.float32) for s in shapes]
    X = rng.standard_normal((batch, dims[0])).astype(np.float32)
    if gens is None:
        # generators = G * random scalars (demo.cu:81-82), scalar-mul on raw limbs
        gens = []
        for s in shapes:
            g = np.repeat(orc.generator(), s["n_gens"], axis=0)
            gens.append(orc.g1_mul(g, rand_fr(rng, s["n_gens"])))
    B_p = 1 << ceil_log2(batch)
    chal = []
    L = len(shapes)

    def fc_ch(s):
        return [rand_fr(rng, ceil_log2(B_p)), rand_fr(rng, ceil_log2(s["in_p"])), rand_fr(rng, ceil_log2(s["out_p"]))]

    def relu_ch(s):
        k = ceil_log2(B_p * s["out_p"])
        return [rand_fr(rng, k + 5), rand_fr(rng, k + 5), rand_fr(rng, k + 4), rand_fr(rng, k + 4), rand_fr(rng, k), rand_fr(rng, k), rand_fr(rng, k)]

    def draw():
        out = [("fc", L - 1, fc_ch(shapes[L - 1]))]
        for i in range(L - 2, -1, -1):
            out.append(("relu", i, relu_ch(shapes[i])))
            out.append(("fc", i, fc_ch(shapes[i])))
        return out

    chal = draw()
    return dict(dims=dims, batch=batch, B_p=B_p, shapes=shapes, W=W, X=X, gens=gens, chal=chal, draw_chal=draw)


def prove_mlp_oracle(orc, model):
    sh, L = model["shapes"], len(model["shapes"])
    B_p = model["B_p"]
    Wq = [orc.fr_mont(orc.float_to_fr(w, s["in_p"], s["out_p"])) for w, s in zip(model["W"], sh)]
    coms = [orc.commit(g, w) for g, w in zip(model["gens"], Wq)]
    X = orc.fr_mont(orc.float_to_fr(model["X"], B_p, sh[0]["in_p"]))
    Z, A, aux = [], [], []
    cur = X
    for i in range(L - 1):
        z = orc.fc_forward(cur, Wq[i], B_p, sh[i]["in_p"], sh[i]["out_p"])
        a, sign, mag, rem = orc.relu_forward(z)
        Z.append(z); A.append(a); aux.append((sign, mag, rem))
        cur = a
    Yhat = orc.fr_unmont(orc.fc_forward(cur, Wq[L - 1], B_p, sh[L - 1]["in_p"], sh[L - 1]["out_p"]))
    frs, g1s = [], []
    for kind, i, ch in model["chal"]:
        if kind == "fc":
            x_in = (A[i - 1] if i > 0 else X)
            z_out = Yhat if i == L - 1 else Z[i]
            f, g = orc.fc_prove(x_in, z_out, Wq[i], sh[i]["in_p"], sh[i]["out_p"], model["gens"][i], coms[i], *ch)
            frs.append(f); g1s.append(g)
        else:
            frs.append(orc.relu_prove(Z[i], A[i], *aux[i], *ch))
    model["_oracle"] = dict(Yhat=Yhat, coms=coms, Wq=Wq)
    return np.concatenate(frs), np.concatenate(g1s)


def setup_mlp_gpu(zk, model, gens_dev=None):
    """Layers, forward pass and commitments on the device; returns the state enqueue_proof() needs."""
    from zkdl_amd.capi import FC, ReLU
    sh, L = model["shapes"], len(model["shapes"])
    gens = gens_dev or [zk.gens(zk.upload(g)) for g in model["gens"]]
    fcs = []
    for i, s in enumerate(sh):
        wf = zk.upload_f32(model["W"][i])
        fcs.append(FC.from_float(zk, s["n_in"], s["n_out"], wf, gens[i]))
    xf = zk.upload_f32(model["X"])
    X = zk.fr_mont(zk.fc_load_input(model["batch"], model["dims"][0], xf))
    relus = [ReLU(zk) for _ in range(L - 1)]
    Z, A = [], []
    cur = X
    for i in range(L - 1):
        z = fcs[i].forward(cur)
        a = relus[i].forward(z)
        Z.append(z); A.append(a)
        cur = a
    Yhat = zk.fr_unmont(fcs[L - 1].forward(cur))
    return dict(L=L, gens=gens, fcs=fcs, relus=relus, X=X, Z=Z, A=A, Yhat=Yhat)


def enqueue_proof(st, chal, tr):
    """The prove calls of one proof, in the reference's order (last layer first); nothing is read back."""
    L = st["L"]
    for kind, i, ch in chal:
        if kind == "fc":
            x_in = (st["A"][i - 1] if i > 0 else st["X"])
            z_out = st["Yhat"] if i == L - 1 else st["Z"][i]
            st["fcs"][i].prove(x_in, z_out, st["gens"][i], ch[0], ch[1], ch[2], tr)
        else:
            st["relus"][i].prove(st["Z"][i], st["A"][i], ch, tr)


def prove_mlp_gpu(zk, model, gens_dev=None, timed=False):
    """Returns (Fr transcript, affine G1 transcript, inf flags, info)."""
    from zkdl_amd.capi import Transcript
    st = setup_mlp_gpu(zk, model, gens_dev)
    tr = Transcript(zk)
    zk.sync()
    if timed:
        zk.timer_start()
    enqueue_proof(st, model["chal"], tr)
    ms = zk.timer_stop_ms() if timed else None
    fr, _ = tr.read()
    aff, inf = tr.read_affine()
    info = dict(ms=ms, Yhat=st["Yhat"].download(), coms=[fc.com().download() for fc in st["fcs"]], fcs=st["fcs"], relus=st["relus"], tr=tr, state=st)
    return fr, aff, inf, info
