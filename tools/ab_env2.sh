#!/bin/bash
# like ab_env.sh with extra bench flags in $FLAGS and a short run
for v in "$@"; do
  env $v timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe --steps ${STEPS:-20} --warmup 3 $FLAGS 2>gpurun_out/ab.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$v] [$FLAGS]', round(d['ms_per_step'],3), d['aggregate_fr'][:18])" || { echo "$v failed"; tail -3 gpurun_out/ab.err; }
done
