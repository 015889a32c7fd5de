#!/bin/bash
out=gpurun_out/ab3.txt
: > $out
run() {
  label=$1; shift
  python3 bench.py --cpu-baseline off --no-msm-probe --steady-steps 0 --ext-steps 0 --latency-reps 5 --prof-steps 1 --steps 60 --warmup 6 "$@" > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_err.log
  python3 - "$label" <<'PY' >> gpurun_out/ab3.txt
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
lat = d.get("latency") or {}
print("%-44s ms/proof %.3f  enqueue %.2f  latency %s agg %s" % (sys.argv[1], d["ms_per_step"], d["host_enqueue_ms_per_step"], {k: (round(v["median_ms"], 2), round(v["min_ms"], 2)) for k, v in lat.items() if isinstance(v, dict)}, d["aggregate_fr"][:12]))
PY
  tail -1 $out
}
run "default (s1 b2 i6 e1)"
run "pair 2" --pair 2
run "pair 2 b4 i8" --pair 2 --g1-batch 4 --inflight 8
run "b2 i5" --inflight 5
run "b2 i8" --inflight 8
run "old (s2 b1 i4)" --g1-batch 1 --inflight 4
run "K=20 default" --steps 20 --warmup 5
run "K=20 old" --steps 20 --warmup 5 --g1-batch 1 --inflight 4
run "K=20 drain 2" --steps 20 --warmup 5 --drain 2
run "K=20 drain 0" --steps 20 --warmup 5 --drain 0
