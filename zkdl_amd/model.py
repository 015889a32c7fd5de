#!/usr/bin/env python3
"""Counterpart of the reference's model.py: emit traced_model.pt / sample_input.pt /
sample_output.pt for `zkdl_amd/host/demo` (same file formats: a traced nn.Sequential of bias-free
Linear + ReLU whose children are named "0","1",...; a tensor stored as parameter "0" of a scripted
empty module, model.py:4-8).  Unlike the reference it takes a seed and a --dims/--batch override, so
tiny fixtures and the full-size synthetic configs come from the same code.

    python -m zkdl_amd.model --out-dir /tmp/m                      # the 18M-parameter MLP, batch 256
    python -m zkdl_amd.model --dims 6,5,4 --batch 3 --seed 7 --out-dir /tmp/m
"""
import argparse
import os

import torch
import torch.nn as nn

DEFAULT_DIMS = [784, 1000, 1773, 1773, 1773, 1773, 1773, 1124, 1000]   # model.py:14-30


def save_tensor(t, fn):
    m = nn.Module()
    m.register_parameter("0", nn.Parameter(t))
    torch.jit.script(m).save(fn)


def activation(spec):
    """relu (the reference's model.py) | leaky:<s> (nn.LeakyReLU with negative slope 2^-s) | identity (nn.Identity) | none (no module at all)"""
    if spec == "relu":
        return nn.ReLU()
    if spec.startswith("leaky:"):
        return nn.LeakyReLU(2.0 ** -int(spec.split(":")[1]))
    if spec == "identity":
        return nn.Identity()
    if spec == "none":
        return None
    raise ValueError("unknown activation " + spec)


def build(dims, seed, bias=False, acts=("relu",)):
    """acts: one spec for every hidden layer, or a single one for all of them"""
    torch.manual_seed(seed)
    layers = []
    n_hidden = len(dims) - 2
    acts = list(acts) * n_hidden if len(acts) == 1 else list(acts)
    if len(acts) != n_hidden:
        raise ValueError("need %d activation specs" % n_hidden)
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        layers.append(nn.Linear(a, b, bias=bias))
        if i + 2 < len(dims):
            m = activation(acts[i])
            if m is not None:
                layers.append(m)
    return nn.Sequential(*layers).eval()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", default=",".join(map(str, DEFAULT_DIMS)))
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--bias", action="store_true", help="Linear layers with bias (the reference's model.py has none; the demo's loader folds them into the matrices)")
    ap.add_argument("--act", default="relu", help="hidden activations, one spec or a comma-separated list (one per hidden layer): relu | leaky:<s> (negative slope "
                    "2^-s) | identity | none (consecutive Linear layers); the reference's model.py has ReLU only")
    ap.add_argument("--like-bench", type=int, default=None, metavar="RANK", help="weights and input exactly as bench.py's Workload draws them for that rank (numpy generator "
                    "1000 + RANK: uniform(+-1/sqrt(in)) matrices, N(0,1) input): `demo --seed 0xC0FFEE --pipeline K` then proves the statements bench.py times")
    ap.add_argument("--out-dir", default=".")
    args = ap.parse_args()
    dims = [int(x) for x in args.dims.split(",")]
    os.makedirs(args.out_dir, exist_ok=True)
    model = build(dims, args.seed, args.bias, tuple(args.act.split(",")))
    x = torch.randn(args.batch, dims[0])
    if args.like_bench is not None:                      # bench.py: Workload.__init__ (same draws in the same order)
        import numpy as np
        rng = np.random.default_rng(1000 + args.like_bench)
        linears = [m for m in model if isinstance(m, nn.Linear)]
        with torch.no_grad():
            for m, (n_in, n_out) in zip(linears, zip(dims[:-1], dims[1:])):
                w = rng.uniform(-1.0 / np.sqrt(n_in), 1.0 / np.sqrt(n_in), size=(n_in, n_out)).astype(np.float32)
                m.weight.copy_(torch.from_numpy(np.ascontiguousarray(w.T)))
        x = torch.from_numpy(rng.standard_normal((args.batch, dims[0])).astype(np.float32))
    with torch.no_grad():
        y = model(x)
    save_tensor(x, os.path.join(args.out_dir, "sample_input.pt"))
    save_tensor(y, os.path.join(args.out_dir, "sample_output.pt"))
    torch.jit.trace(model, x[:1]).save(os.path.join(args.out_dir, "traced_model.pt"))
    print("wrote traced_model.pt, sample_input.pt, sample_output.pt to", args.out_dir)


if __name__ == "__main__":
    main()
