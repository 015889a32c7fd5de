#!/bin/bash
out=gpurun_out/ab7.txt
: > $out
run() {
  label=$1; shift
  python3 bench.py --cpu-baseline off --no-msm-probe --steady-steps 0 --ext-steps 0 --latency-reps 3 --prof-steps 1 --steps 60 --warmup 6 "$@" > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_err.log
  python3 - "$label" <<'PY' >> gpurun_out/ab7.txt
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print("%-34s ms/proof %.3f enq %.2f lat %.2f" % (sys.argv[1], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["latency_ms_per_proof"]))
PY
  tail -1 $out
}
run "default"
ZKDL_LANE_PRIO=1 run "lanes high priority"
ZKDL_LANE_PRIO=-1 run "lanes low priority"
ZKDL_LANES=3 run "3 lanes (groups 1-3 exactly)"
ZKDL_LANES=6 run "6 lanes"
ZKDL_G1_CU_PERCENT=33 run "G1 on a third of the CUs"
ZKDL_G1_CU_PERCENT=66 run "G1 on two thirds"
ZKDL_HP_GROUP=0 run "hadamard chains with group 0"
ZKDL_HP_GROUP=2 run "hadamard chains with group 2"
ZKDL_HP_GROUP=3 run "hadamard chains with group 3"
run "default again"
