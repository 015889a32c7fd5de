#!/bin/bash
# A/B of bench flag sets (one per argument) on the default workload, K=100
for f in "$@"; do
  timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe --steps ${STEPS:-100} --warmup 5 $f 2>gpurun_out/ab.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$f]', round(d['ms_per_step'],3), d['aggregate_fr'][:12])" || { echo "$f failed"; tail -3 gpurun_out/ab.err; }
done
