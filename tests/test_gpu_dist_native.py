"""The native collective (zk_dist_*: RCCL over the context's stream, no torch in the process) with the one rank a 1-GPU box has: digest rows of
several sealed proofs are summed on the device, all-reduced (world 1) and reduced mod r -- equal to the sum of the proofs' host-side digests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
assert "torch" not in sys.modules
import bench
from zkdl_amd.capi import MSM_PIPPENGER, Transcript, Zk, _hp, _vp
zk = Zk(0); zk.set_msm_mode(MSM_PIPPENGER)
wl = bench.Workload(zk, [64, 96, 48, 32], 16, seed=1, inflight=3)
lib = zk.lib
uid = (ctypes.c_ubyte * 128)()
assert lib.zk_dist_unique_id(uid) == 0
d = ctypes.c_void_p()
zk._chk(lib.zk_dist_init(zk.ctx, 0, 1, uid, ctypes.byref(d)))
assert lib.zk_dist_world_size(d) == 1
K = 5
rows = zk.empty(2 * K, 8)                                   # K rows of 8 x uint64
want = 0
R = bench.R_MOD
for i in range(K):
    tr = wl.trs[i %% 3]
    wl.prove(wl.challenges(i), tr)
    tr.seal(rows.ptr + 64 * i)
agg = zk.empty(2, 8)
zk._chk(lib.zk_dist_sum_rows(zk.ctx, _vp(rows.ptr), ctypes.c_size_t(K), _vp(agg.ptr)))
zk._chk(lib.zk_dist_allreduce_u64(d, _vp(agg.ptr), ctypes.c_size_t(8)))
lanes = agg.download().reshape(-1).view(np.uint64)[:8].copy()
out = np.zeros(8, np.uint32)
assert lib.zk_dist_lanes_to_fr(_hp(lanes), _hp(out)) == 0
got = int.from_bytes(out.tobytes(), "little")
chk = Transcript(zk)
for i in range(K):
    wl.prove(wl.challenges(i), chk)
    want = (want + int.from_bytes(np.ascontiguousarray(chk.digest()).tobytes(), "little")) %% R
assert got == want, (hex(got), hex(want))
assert lib.zk_dist_destroy(d) == 0
zk.close()
print("native aggregate ok", hex(got))
'''


def test_native_rccl_aggregate_without_torch():
    code = SCRIPT % (ROOT, os.path.join(ROOT, "tools"))
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "native aggregate ok" in run.stdout
