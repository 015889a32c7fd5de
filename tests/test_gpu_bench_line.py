"""The bench contract the driver depends on: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line whose headline is
BASELINE.json's metric on BASELINE.json's workload, with the `roofline` and `cpu_baseline` objects beside it, and whose timed
configuration was compared with the CPU oracle in the same run (`parity_checked`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_bench_line_keeps_the_contract(tmp_path):
    detail = str(tmp_path / "bench_detail.json")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--cpu-baseline", "sample", "--steady-steps", "0",
                          "--latency-reps", "1", "--ext-steps", "0", "--no-msm-probe", "--no-prewarm", "--detail-out", detail], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                                            # ONE line on stdout
    assert len(lines[-1]) < 4096                                      # the driver parses the LAST line out of an 8 KB tail (BENCH_r03: 27.6 KB, unparsed)
    d = json.loads(lines[-1])                                         # ... and that line alone has to satisfy the contract
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "parity_checked"):
        assert key in d, key
    full = json.load(open(detail))                                    # everything else lives in the detail file
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"].split(";")[0] and d["unit"] == "ms/example"
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is False and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["config"]["workload"] == "mlp18m_b256" and d["config"]["batch_per_gpu"] == 256
    assert d["dtype"].startswith("u32x8")                              # the reference's limb widths, nothing narrower
    assert abs(d["value"] - d["ms_per_step"] / 256) < 1e-12 and 0 < d["ms_per_step"] < 100
    r = d["roofline"]
    assert set(("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms")) <= set(r)
    assert r["bound"] in ("hbm", "mfma", "valu") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["hbm"]["peak"] == 8000.0 and r["hbm"]["unit"] == "GB/s"    # the HBM view of the same kernel
    assert "roofline_all" not in d and "kernels" not in d
    assert any(x["kernel"] == r["kernel"] for x in full["roofline_all"]) and len(full["roofline_all"]) >= 6 and full["kernels"]
    c = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    p = d["parity_checked"]
    assert p["ok"] is True and p["mismatch"] == [] and p["fr"] > 0 and p["g1"] > 0 and p["calls"] > 0
    assert d["latency_ms_per_proof"] > 0 and d["collective_world_size"] == 1


def test_launcher_path_with_rccl_on_one_rank(tmp_path):
    """The multi-GPU path as far as a one-GPU box can take it: `--spawn` starts the rank as a child process (the parent never touches the GPU, relays
    rank 0's line, propagates its status) and `--collective --backend nccl` creates the RCCL process group and runs the aggregate's all-reduce on the
    single rank.  Same run aggregate as the in-process, collective-free run of the same steps.  (N > 1 needs N devices; the driver's 8-GPU run is the
    first time more than one rank meets.)"""
    common = ["--workload", "tiny", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--steady-steps", "0", "--latency-reps", "0", "--ext-steps", "0", "--no-msm-probe",
              "--no-prewarm"]
    runs = []
    for extra in (["--spawn", "--collective", "--backend", "nccl"], []):
        run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--detail-out", str(tmp_path / "d.json")] + common + extra,
                             capture_output=True, text=True, cwd=ROOT, timeout=600)
        assert run.returncode == 0, run.stderr[-2000:]
        lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1 and len(lines[0]) < 4096
        runs.append(json.loads(lines[0]))
    spawned, plain = runs
    assert spawned["collective_backend"] == "nccl" and spawned["collective_world_size"] == 1 and spawned["n_gpus"] == 1
    assert plain.get("collective_backend") is None
    assert spawned["aggregate_fr"] == plain["aggregate_fr"]
