#!/bin/bash
out=gpurun_out/ab5.txt
: > $out
run() {
  label=$1; shift
  python3 bench.py --cpu-baseline off --steady-steps 0 --ext-steps 0 --latency-reps 5 --prof-steps 2 --steps 60 --warmup 6 "$@" > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_err.log
  python3 - "$label" <<'PY' >> gpurun_out/ab5.txt
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
lat = d.get("latency") or {}
k = d["kernels"]
print("%-30s ms/proof %.3f enq %.2f lat %s mops %s | %s" % (sys.argv[1], d["ms_per_step"], d["host_enqueue_ms_per_step"],
      {n[:10]: round(v["median_ms"], 2) for n, v in lat.items() if isinstance(v, dict)}, round(d.get("g1_msm_mops") or 0),
      " ".join("%s=%.2f" % (n, k[n]["ms_per_step"]) for n in k if n.startswith("msm_") or n == "g1_canonicalise")))
PY
  tail -1 $out
}
run "ft12 b2 i6"
run "ft12 b3 i8" --g1-batch 3 --inflight 8
run "ft12 b4 i8" --g1-batch 4 --inflight 8
run "ft12 b4 i10" --g1-batch 4 --inflight 10
run "ft12 b3 i6" --g1-batch 3 --inflight 6
run "bucket b3 i8" --full-tables 0 --g1-batch 3 --inflight 8
run "bucket b2 i6" --full-tables 0
run "ft12 b3 i8 noderive" --g1-batch 3 --inflight 8 --no-derive
run "ft12 b3 i8 K=20" --g1-batch 3 --inflight 8 --steps 20 --warmup 5
run "ft12 b2 i6 K=20" --steps 20 --warmup 5
run "ft13 b3 i8" --g1-batch 3 --inflight 8 --full-tables 13
