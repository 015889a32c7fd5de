"""N > 1 path on CPU: world_size-2 gloo all-reduce of the per-proof aggregate (8 x int64 lanes),
then local carry-propagation + reduction mod r.  Mirrors what bench.py does over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zkdl_amd.dist import R_MOD, allreduce_digest, fr_to_lanes, lanes_to_fr  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, digests, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lanes = torch.tensor(fr_to_lanes(digests[rank]), dtype=torch.int64)
    agg = allreduce_digest(lanes)
    out[rank] = agg
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_allreduce_of_fr_digest(world):
    rng = np.random.default_rng(9)
    # worst-case carries: r - 1 on every rank, plus random values
    digests = [R_MOD - 1] + [int.from_bytes(rng.bytes(32), "little") % R_MOD for _ in range(world - 1)]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), digests, out), nprocs=world, join=True)
    want = sum(digests) % R_MOD
    assert all(out[r] == want for r in range(world))


def test_lane_roundtrip_and_carry():
    for v in (0, 1, R_MOD - 1, (1 << 255) % R_MOD, 0xFFFFFFFF, 0xFFFFFFFF << 32):
        assert lanes_to_fr(fr_to_lanes(v)) == v % R_MOD
    # lanes that each overflow 32 bits after summation still reduce correctly
    lanes = [0xFFFFFFFF * 7] * 8
    assert lanes_to_fr(lanes) == sum((0xFFFFFFFF * 7) << (32 * j) for j in range(8)) % R_MOD


def _run_bench(*argv):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=env, timeout=300)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_bench_launches_its_own_ranks(world):
    """`python bench.py --gpus N` as the driver invokes it (no WORLD_SIZE): the parent spawns N ranks, rank 0's
    JSON line comes back on stdout, and the ranks see each other (gloo here; the collective path on CPU)."""
    import json
    out = _run_bench("--gpus", str(world), "--launcher-selftest", "-1")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    want = sum(pow(7, r + 1, R_MOD) * (R_MOD - 1) % R_MOD for r in range(world)) % R_MOD
    assert rec["n_gpus"] == world and rec["collective_world_size"] == world and int(rec["aggregate_fr"], 16) == want
    assert rec["ranks"] == list(range(world))               # every rank reported in (world 8: the driver's scaling run's process layout)


def test_bench_launcher_propagates_a_dead_rank():
    out = _run_bench("--gpus", "2", "--launcher-selftest", "1")
    assert out.returncode == 3


def _gather_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zkdl_amd.shard import dist_allgather
    payload = [dict(a=np.full((3, 8), rank, dtype=np.uint32), z=np.arange(8, dtype=np.uint32).reshape(1, 8) + rank), dict(fr=np.full((5, 8), 10 * rank, dtype=np.uint32))]
    got = dist_allgather(payload)
    out[rank] = all(int(got[r][0]["a"][0, 0]) == r and int(got[r][1]["fr"][4, 7]) == 10 * r and int(got[r][0]["z"][0, 1]) == 1 + r for r in range(world)) and len(got) == world
    dist.barrier()
    dist.destroy_process_group()


def test_shard_payload_allgather_over_gloo():
    """The sharded prover's only exchange (zkdl_amd/shard.py: two all-gathers of per-call payloads), world size 2 on CPU."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gather_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] and out[1]


def test_shard_rank_weights_are_the_eq_polynomial():
    from zkdl_amd.shard import _to_int, _to_mont, eq_weight
    rng = np.random.default_rng(2)
    u = rng.integers(0, 1 << 32, size=(3, 8), dtype=np.uint64).astype(np.uint32)
    u[:, 7] %= 1944954707
    ws = [_to_int(eq_weight(u, r)) for r in range(8)]
    assert sum(ws) % R_MOD == 1                                   # the eq weights of all ranks sum to one
    uv = [_to_int(x) for x in u]
    assert ws[5] == uv[0] * (1 - uv[1]) * uv[2] % R_MOD           # rank 5 = bits (1, 0, 1), LSB first
    assert _to_int(eq_weight(u[:0], 0)) == 1 and _to_int(_to_mont(12345)) == 12345


def test_native_lanes_to_fr_matches_the_python_reduction():
    """zk_dist_lanes_to_fr (host code of the native RCCL path: carry-propagate 8 summed 64-bit lanes, reduce mod r) against zkdl_amd.dist.lanes_to_fr."""
    import ctypes
    import numpy as np
    from zkdl_amd.capi import load_library
    from zkdl_amd.dist import lanes_to_fr
    lib = load_library()
    rng = np.random.default_rng(12)
    cases = [rng.integers(0, 1 << 63, size=8, dtype=np.uint64) for _ in range(50)]
    cases += [np.zeros(8, np.uint64), np.full(8, (1 << 64) - 1, dtype=np.uint64), np.array([1, 0, 0, 0, 0, 0, 0, 0], np.uint64)]
    for lanes in cases:
        out = np.zeros(8, np.uint32)
        assert lib.zk_dist_lanes_to_fr(lanes.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
        assert int.from_bytes(out.tobytes(), "little") == lanes_to_fr([int(x) for x in lanes])
