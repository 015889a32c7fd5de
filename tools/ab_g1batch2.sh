#!/bin/bash
# second same-box sweep: one G1 stream, batched G1 phases; early start on/off; CU mask share of the G1 stream
out=gpurun_out/ab_g1batch2.txt
: > $out
run() {
  label=$1; shift
  python3 bench.py --cpu-baseline off --no-msm-probe --steady-steps 0 --ext-steps 0 --latency-reps 0 --prof-steps 1 --steps 60 --warmup 6 "$@" > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_err.log
  python3 - "$label" <<'PY' >> gpurun_out/ab_g1batch2.txt
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print("%-44s ms/proof %.3f  enqueue %.2f  agg %s" % (sys.argv[1], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["aggregate_fr"][:12]))
PY
  tail -1 $out
}
export ZKDL_G1_STREAMS=1
run "s1 b2 i6 e1" --g1-batch 2 --inflight 6
ZKDL_BENCH_FLAGS=early_g1=0 run "s1 b2 i6 e0" --g1-batch 2 --inflight 6
ZKDL_BENCH_FLAGS=early_g1=0 run "s1 b2 i4 e0" --g1-batch 2 --inflight 4
ZKDL_BENCH_FLAGS=early_g1=0 run "s1 b3 i6 e0" --g1-batch 3 --inflight 6
ZKDL_BENCH_FLAGS=early_g1=0 run "s1 b4 i8 e0" --g1-batch 4 --inflight 8
ZKDL_G1_CU_PERCENT=100 ZKDL_BENCH_FLAGS=early_g1=0 run "s1 b2 i6 e0 cu100" --g1-batch 2 --inflight 6
ZKDL_G1_CU_PERCENT=75 ZKDL_BENCH_FLAGS=early_g1=0 run "s1 b2 i6 e0 cu75" --g1-batch 2 --inflight 6
ZKDL_G1_CU_PERCENT=25 ZKDL_BENCH_FLAGS=early_g1=0 run "s1 b2 i6 e0 cu25" --g1-batch 2 --inflight 6
ZKDL_BENCH_FLAGS=early_g1=0 run "s1 b1 i4 e0" --inflight 4
ZKDL_G1_STREAMS=2 ZKDL_BENCH_FLAGS=early_g1=0 run "s2 b1 i4 e0 (round-2 default)" --inflight 4
ZKDL_G1_STREAMS=2 run "s2 b1 i4 e1" --inflight 4
run "s1 b2 i6 e1 again" --g1-batch 2 --inflight 6
ZKDL_BENCH_FLAGS=early_g1=0 run "s1 b2 i6 e0 K=20" --g1-batch 2 --inflight 6 --steps 20 --warmup 5
run "s1 b2 i6 e1 K=20" --g1-batch 2 --inflight 6 --steps 20 --warmup 5
ZKDL_G1_STREAMS=2 ZKDL_BENCH_FLAGS=early_g1=0 run "s2 b1 i4 e0 K=20" --inflight 4 --steps 20 --warmup 5
