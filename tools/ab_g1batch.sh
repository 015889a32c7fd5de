#!/bin/bash
# same-box A/B of the G1-phase scheduling knobs: g1_batch x inflight x early_g1 (K = 60 proofs each)
out=gpurun_out/ab_g1batch.txt
: > $out
run() {  # label, extra flags..., env
  label=$1; shift
  python3 bench.py --cpu-baseline off --no-msm-probe --steady-steps 0 --ext-steps 0 --latency-reps 3 --prof-steps 1 --steps 60 --warmup 6 "$@" > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_err.log
  python3 - "$label" <<'PY' >> gpurun_out/ab_g1batch.txt
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
lat = d.get("latency") or {}
print("%-40s ms/proof %.3f  enqueue %.2f  latency %s  agg %s" % (sys.argv[1], d["ms_per_step"], d["host_enqueue_ms_per_step"],
      {k: round(v["median_ms"], 2) for k, v in lat.items() if isinstance(v, dict)}, d["aggregate_fr"][:12]))
PY
  tail -1 $out
}
run "base g1_batch=1 inflight=4"
ZKDL_BENCH_FLAGS=early_g1=0 run "early_g1=0 g1_batch=1 inflight=4"
run "g1_batch=2 inflight=4" --g1-batch 2
run "g1_batch=2 inflight=6" --g1-batch 2 --inflight 6
run "g1_batch=2 inflight=8" --g1-batch 2 --inflight 8
run "g1_batch=3 inflight=6" --g1-batch 3 --inflight 6
run "g1_batch=4 inflight=8" --g1-batch 4 --inflight 8
ZKDL_G1_STREAMS=1 run "g1_batch=2 inflight=6 1 G1 stream" --g1-batch 2 --inflight 6
ZKDL_G1_STREAMS=1 run "g1_batch=4 inflight=8 1 G1 stream" --g1-batch 4 --inflight 8
ZKDL_G1_STREAMS=3 run "g1_batch=2 inflight=8 3 G1 streams" --g1-batch 2 --inflight 8
run "base again" 
