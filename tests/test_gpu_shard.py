"""One proof sharded by batch over N (simulated) ranks == the single-GPU proof, bit for bit (zkdl_amd/shard.py)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def zk():
    from zkdl_amd.capi import MSM_PIPPENGER, Zk
    z = Zk(0)
    z.set_msm_mode(MSM_PIPPENGER)
    yield z
    z.close()


@pytest.mark.parametrize("dims,batch,world", [([64, 96, 48, 32], 16, 2), ([64, 96, 48, 32], 16, 4), ([64, 96, 48, 32], 16, 16), ([40, 24, 12], 8, 1)])
def test_sharded_proof_equals_single_gpu_proof(zk, dims, batch, world):
    import bench
    from zkdl_amd.capi import Transcript
    from zkdl_amd.proof import fs_challenges, statement_digest, verify_mlp
    from zkdl_amd.shard import Shard, prove_sharded
    whole = bench.Workload(zk, dims, batch, seed=4)
    coms = [fc.com() for fc in whole.fcs]
    chal = fs_challenges(statement_digest(zk, dims, batch, whole.gens, coms, whole.X, whole.Yhat), dims, batch)
    tr = Transcript(zk)
    whole.prove(chal, tr)
    fr, _ = tr.read()
    aff, inf = tr.read_affine()
    per = batch // world
    shards = [Shard(bench.Workload(zk, dims, batch, seed=4, rows=(r * per, (r + 1) * per)), r, world) for r in range(world)]
    fr_s, aff_s, inf_s = prove_sharded(shards, chal)
    assert np.array_equal(fr_s, fr)
    assert np.array_equal(aff_s, aff) and np.array_equal(inf_s, inf)
    # and the verifier accepts the sharded proof as it accepts the whole one
    blob = b"ZKDLTR01" + int(fr_s.shape[0]).to_bytes(8, "little") + int(aff_s.shape[0]).to_bytes(8, "little") + fr_s.tobytes() + aff_s.tobytes() + inf_s.tobytes()
    assert blob == tr.serialize()
    ok, msg, _ = verify_mlp(zk, dims, batch, whole.gens, coms, whole.X, whole.Yhat, blob)
    assert ok, msg


def test_mlp18m_batch256_sharded_over_8(zk):
    """BASELINE configs[2] as ONE proof over 8 ranks' shards of 32 rows each."""
    import bench
    from zkdl_amd.capi import Transcript
    from zkdl_amd.shard import Shard, prove_sharded
    dims, batch = bench.WORKLOADS["mlp18m_b256"]
    whole = bench.Workload(zk, dims, batch, seed=2)
    chal = whole.challenges(0)
    tr = Transcript(zk)
    whole.prove(chal, tr)
    fr, _ = tr.read()
    aff, inf = tr.read_affine()
    del whole
    shards = [Shard(bench.Workload(zk, dims, batch, seed=2, rows=(32 * r, 32 * r + 32)), r, 8) for r in range(8)]
    fr_s, aff_s, inf_s = prove_sharded(shards, chal)
    assert np.array_equal(fr_s, fr) and np.array_equal(aff_s, aff) and np.array_equal(inf_s, inf)


@pytest.mark.parametrize("dims,batch", [([40, 24, 12], 8), ([64, 96, 48, 32], 16)])
def test_pipelined_sharded_stream_equals_the_sequential_one(zk, dims, batch):
    """prove_sharded_stream keeps five sharded proofs in flight (every phase a sealed transcript, one read per phase waits for that transcript
    only): each proof it yields is the single-GPU transcript of its own challenges, bit for bit, in order."""
    import bench
    from zkdl_amd.capi import Transcript
    from zkdl_amd.shard import Shard, prove_sharded, prove_sharded_stream
    whole = bench.Workload(zk, dims, batch, seed=11)
    chals = [whole.challenges(st) for st in range(7)]
    want = []
    tr = Transcript(zk)
    for ch in chals:
        whole.prove(ch, tr)
        want.append((tr.read()[0], tr.read_affine()))
    tr.close()
    shard = Shard(whole, 0, 1)
    got = list(prove_sharded_stream(shard, chals))
    assert len(got) == len(chals)
    for (fr, aff, inf), (fr_w, (aff_w, inf_w)) in zip(got, want):
        assert np.array_equal(fr, fr_w) and np.array_equal(aff, aff_w) and np.array_equal(inf, inf_w)
    # and the unpipelined form still agrees
    fr, aff, inf = prove_sharded([shard], chals[3])
    assert np.array_equal(fr, want[3][0]) and np.array_equal(aff, want[3][1][0])
