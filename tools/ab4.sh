#!/bin/bash
out=gpurun_out/ab4.txt
: > $out
run() {
  label=$1; shift
  python3 bench.py --cpu-baseline off --steady-steps 0 --ext-steps 0 --latency-reps 5 --prof-steps 2 --steps 60 --warmup 6 "$@" > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_err.log
  python3 - "$label" <<'PY' >> gpurun_out/ab4.txt
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
lat = d.get("latency") or {}
k = d["kernels"]
print("%-34s ms/proof %.3f enq %.2f lat %s setup_ft %.2fs mops %s | %s" % (sys.argv[1], d["ms_per_step"], d["host_enqueue_ms_per_step"],
      {n: round(v["median_ms"], 2) for n, v in lat.items() if isinstance(v, dict)}, d["setup"].get("setup_full_tables_s", 0), round(d.get("g1_msm_mops") or 0),
      " ".join("%s=%.2f" % (n, k[n]["ms_per_step"]) for n in k if n.startswith("msm_") or n == "g1_canonicalise")))
PY
  tail -1 $out
}
run "full tables c=12"
run "bucket method" --full-tables 0
run "full tables c=11" --full-tables 11
run "full tables c=12 K=20" --steps 20 --warmup 5
run "bucket K=20" --full-tables 0 --steps 20 --warmup 5
run "full tables c=12 b1 s2 i4" --g1-batch 1 --inflight 4
run "full tables c=12 noderive" --no-derive
run "full tables c=12 b3 i8" --g1-batch 3 --inflight 8
run "full tables c=12 again"
