import sys, time
sys.path.insert(0, ".")
from zkdl_amd.capi import Zk
zk = Zk(0)
for k in (21, 19):
    t = zk.fr_random(1 << k); u = zk.random_vec(k); v = zk.random_vec(k)
    zk.sumcheck_binary(t, u, v); zk.sync()
    zk.prof_reset(); zk.prof_enable(True)
    for _ in range(5): zk.sumcheck_binary(t, u, v)
    zk.sync(); p = zk.prof_read(); zk.prof_enable(False)
    print(k, {n: round(x["ms"] / 5, 3) for n, x in p.items()})
