#!/bin/bash
# same-box A/B: tools/ab_libs_env.sh "<lib> <ENV=..>" ...
for spec in "$@"; do
  set -- $spec; v=$1; shift
  cp zkdl_amd/lib/variants/lib_$v.so zkdl_amd/lib/libzkdl_amd.so
  env "$@" A=1 timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe --steps 100 --warmup 5 2>gpurun_out/ab.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$spec', round(d['ms_per_step'],3))" || { echo "$spec failed"; tail -3 gpurun_out/ab.err; }
done
