"""Debug aid: which entries of a Hadamard / binary sumcheck transcript differ from the oracle (run on the GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zkdl_amd.capi import Zk          # noqa: E402
from oracle.oracle_py import Oracle   # noqa: E402

zk, oracle = Zk(), Oracle()
rng = np.random.default_rng(1)


def rand_fr(n):
    return oracle.fr_mont(rng.integers(0, 2 ** 32, size=(n, 8), dtype=np.uint64).astype(np.uint32) & np.array([0xffffffff] * 7 + [0x3fffffff], dtype=np.uint32))


for n, k in ((4096, 12), (8192, 13)):
    a, b, u, v = rand_fr(n), rand_fr(n), rand_fr(k), rand_fr(k)
    for name, got, want in (("hp", zk.sumcheck_hadamard(zk.upload(a), zk.upload(b), u, v), oracle.hp_sumcheck(a, b, u, v)),
                            ("bin", zk.sumcheck_binary(zk.upload(a), u, v), oracle.bin_sumcheck(a, u, v)),
                            ("ip", zk.sumcheck_ip(zk.upload(a), zk.upload(b), u), oracle.ip_sumcheck(a, b, u))):
        got, want = np.asarray(got), np.asarray(want)
        bad = [i for i in range(len(want)) if not np.array_equal(got[i], want[i])]
        print(name, n, k, "mismatching transcript entries:", bad[:24], "of", len(want), flush=True)

# weights eq(0, h) = [h == 0]: the first big round's c0 of a Hadamard sumcheck is a[0] * b[0]
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bigint_model as bm  # noqa: E402
n, k = 2048, 11
a, b, v = rand_fr(n), rand_fr(n), rand_fr(k)
u = np.zeros((k, 8), dtype=np.uint32)
got = np.asarray(zk.sumcheck_hadamard(zk.upload(a), zk.upload(b), u, v))
want = np.asarray(oracle.hp_sumcheck(a, b, u, v))
R = bm.R_MOD
ai = [bm.limbs_to_int(x) for x in oracle.fr_unmont(a)]
bi = [bm.limbs_to_int(x) for x in oracle.fr_unmont(b)]
g0 = bm.limbs_to_int(oracle.fr_unmont(got[:1])[0])
w0 = bm.limbs_to_int(oracle.fr_unmont(want[:1])[0])
print("c0 got", hex(g0)[:18], "want", hex(w0)[:18], "a0b0", hex(ai[0] * bi[0] % R)[:18], "sum a_even b_even", hex(sum(ai[2 * h] * bi[2 * h] for h in range(n // 2)) % R)[:18])
for name, val in (("a0b0/R", ai[0] * bi[0] * pow(1 << 256, -1, R) % R), ("a0b0*R", ai[0] * bi[0] * (1 << 256) % R), ("0", 0)):
    print(name, hex(val)[:18])
