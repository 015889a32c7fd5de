"""bench.py's ONE stdout line, checked without a GPU: the committed full record of the last run (profiles/r04_bench_detail.json) goes through the same
`compact_line` the bench prints with, and the result must keep the driver's contract -- every required key, the two extra objects, under 4 KB
(BENCH_r03 was 27.6 KB and the driver could not parse it)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_the_committed_full_record_compacts_to_a_contract_line_under_4_kb():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_detail.json")))
    line = bench.compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.LINE_LIMIT == 4096
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"] == base["metric"].split(";")[0] and line["higher_is_better"] is False and line["vs_baseline"] is None
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]          # the headline keeps its full precision
    assert abs(line["value"] - line["ms_per_step"] / line["config"]["batch_per_gpu"]) < 1e-12
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma", "valu") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert line["config"]["workload"] == "mlp18m_b256"
    # the line that was committed beside it is this very compaction
    committed = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")))
    for key in ("value", "ms_per_step", "steps", "warmup"):
        assert committed[key] == line[key], key
