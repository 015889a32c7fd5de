#!/usr/bin/env python3
"""Kernel micro-benchmarks through the C ABI (HIP-event timed on the library's stream)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkdl_amd.capi import Zk  # noqa: E402


def timeit(zk, fn, reps=5):
    fn()
    zk.sync()
    zk.timer_start()
    for _ in range(reps):
        fn()
    return zk.timer_stop_ms() / reps


def main():
    zk = Zk(0)
    zk.seed(1)
    out = {}
    n = 1 << 24
    a, b = zk.fr_random(n), zk.fr_random(n)
    for name, fn, nbytes in [("fr_add", lambda: zk.fr_add(a, b), 96 * n), ("fr_mul", lambda: zk.fr_mul(a, b), 96 * n),
                             ("fr_mont", lambda: zk.fr_mont(a), 64 * n), ("fr_sum", lambda: zk.fr_sum(a), 32 * n)]:
        ms = timeit(zk, fn)
        out[name] = dict(ms=ms, GBps=nbytes / ms / 1e6, Gelem_s=n / ms / 1e6)
    u = zk.random_vec(24)
    ms = timeit(zk, lambda: zk.fr_me(a, u), 3)
    out["fr_me_2^24"] = dict(ms=ms, GBps=32 * n / ms / 1e6)
    n2 = 1 << 22
    a2 = zk.fr_random(n2)
    b2 = zk.fr_random(n2)
    u22, v22 = zk.random_vec(22), zk.random_vec(22)
    ms = timeit(zk, lambda: zk.sumcheck_binary(a2, u22, v22), 3)
    out["sumcheck_bin_2^22"] = dict(ms=ms, alg_GBps=96 * n2 / ms / 1e6)
    ms = timeit(zk, lambda: zk.sumcheck_hadamard(a2, b2, u22, v22), 3)
    out["sumcheck_hp_2^22"] = dict(ms=ms, alg_GBps=192 * n2 / ms / 1e6)
    # G1
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); from oracle_free_gen import generator_jac
    ng = 1 << 16
    G = zk.g1_fill(ng, generator_jac())
    s = zk.fr_random(ng)
    t0 = time.time()
    P = zk.g1_mul(G, s)
    zk.sync()
    out["g1_mul_naive_65536"] = dict(ms=(time.time() - t0) * 1e3, Mop_s=ng / (time.time() - t0) / 1e6)
    Q = zk.g1_add(P, G)
    ms = timeit(zk, lambda: zk.g1_add(P, Q), 5)
    out["g1_add_65536"] = dict(ms=ms, Madd_s=ng / ms / 1e3)
    ms = timeit(zk, lambda: zk.g1_sum(P), 3)
    out["g1_sum_65536"] = dict(ms=ms)
    print(json.dumps(out, indent=1))
    zk.close()


if __name__ == "__main__":
    main()
