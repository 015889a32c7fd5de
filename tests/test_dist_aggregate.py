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


@pytest.mark.parametrize("world", [2, 3])
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


def test_bench_launcher_propagates_a_dead_rank():
    out = _run_bench("--gpus", "2", "--launcher-selftest", "1")
    assert out.returncode == 3
