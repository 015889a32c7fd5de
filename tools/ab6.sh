#!/bin/bash
out=gpurun_out/ab6.txt
: > $out
run() {
  label=$1; shift
  python3 bench.py --cpu-baseline off --no-msm-probe --steady-steps 0 --ext-steps 0 --latency-reps 3 --prof-steps 2 --steps 60 --warmup 6 "$@" > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_err.log
  python3 - "$label" <<'PY' >> gpurun_out/ab6.txt
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
lat = d.get("latency") or {}
k = d["kernels"]
print("%-30s ms/proof %.3f enq %.2f lat %s | %s" % (sys.argv[1], d["ms_per_step"], d["host_enqueue_ms_per_step"],
      {n[:10]: round(v["median_ms"], 2) for n, v in lat.items() if isinstance(v, dict)},
      " ".join("%s=%.2f" % (n, k[n]["ms_per_step"]) for n in k if n.startswith("msm_") or n == "g1_canonicalise")))
PY
  tail -1 $out
}
run "v0 b3 i8" --g1-batch 3 --inflight 8
ZKDL_FT_VARIANT=1 run "v1 (2 ahead) b3 i8" --g1-batch 3 --inflight 8
ZKDL_FT_VARIANT=2 run "v2 (calls, 3 waves) b3 i8" --g1-batch 3 --inflight 8
ZKDL_FT_VARIANT=3 run "v3 (calls, 4 waves) b3 i8" --g1-batch 3 --inflight 8
run "v0 b4 i8 K=20" --g1-batch 4 --inflight 8 --steps 20 --warmup 5
run "v0 b3 i8 K=20" --g1-batch 3 --inflight 8 --steps 20 --warmup 5
run "v0 b2 i6 K=20" --steps 20 --warmup 5
run "v0 b2 i6 K=20 drain1" --steps 20 --warmup 5 --drain 1
run "v0 b3 i8 K=20 drain1" --g1-batch 3 --inflight 8 --steps 20 --warmup 5 --drain 1
run "v0 b4 i8" --g1-batch 4 --inflight 8
