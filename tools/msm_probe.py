#!/usr/bin/env python3
"""Latency / throughput probe of the MSM pipeline and of single-wave G1 ops (through the C ABI)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle_free_gen import generator_jac  # noqa: E402
from zkdl_amd.capi import Zk  # noqa: E402


def timeit(zk, fn, reps=3):
    fn()
    zk.sync()
    zk.timer_start()
    for _ in range(reps):
        fn()
    return zk.timer_stop_ms() / reps


def main():
    zk = Zk(0)
    zk.seed(3)
    for n in (64, 2048, 65536):
        G = zk.g1_fill(n, generator_jac())
        P = zk.g1_mul(G, zk.fr_random(n))
        Q = zk.g1_add(P, G)
        aff = zk.upload(zk.g1_to_affine(Q)[0])
        print("n=%6d  g1_add %.3f ms   g1_madd %.3f ms" % (n, timeit(zk, lambda: zk.g1_add(P, Q)), timeit(zk, lambda: zk.g1_madd(P, aff))))
    n = 2048
    G = zk.g1_fill(n, generator_jac())
    gens = zk.gens(zk.g1_mul(G, zk.fr_random(n)))
    for q in (1, 4, 34):
        s = zk.fr_random(q * n)
        zk.msm_batch(gens, s)          # builds the table
        zk.sync()
        ms = timeit(zk, lambda: zk.msm_batch(gens, s), 2)
        zk.prof_reset(); zk.prof_enable(True)
        zk.msm_batch(gens, s)
        zk.sync()
        pr = zk.prof_read(); zk.prof_enable(False)
        print("msm n=%d q=%d: %.3f ms  (%.2f Mpairs/s)  " % (n, q, ms, q * n / ms / 1e3), {k: round(v["ms"], 3) for k, v in pr.items()})
    for w in (6, 8, 10, 13):
        zk.set_msm_window(w)
        s = zk.fr_random(34 * n)
        zk.msm_batch(gens, s)
        zk.sync()
        zk.prof_reset(); zk.prof_enable(True)
        zk.msm_batch(gens, s)
        zk.sync()
        pr = zk.prof_read(); zk.prof_enable(False)
        print("window %d q=34:" % w, {k: round(v["ms"], 3) for k, v in pr.items()})
    zk.close()


if __name__ == "__main__":
    main()
