#!/bin/bash
# A/B of library variants built into zkdl_amd/lib/variants/ (run on the GPU box: the tree there is a scratch copy)
for v in "$@"; do
  cp zkdl_amd/lib/variants/lib_$v.so zkdl_amd/lib/libzkdl_amd.so
  for k in 100; do
    timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe --steps $k --warmup 5 2>gpurun_out/ab.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$v K=$k', round(d['ms_per_step'],3), 'accum', round(k['msm_accum']['ms_per_step'],3), 'conc', round(k['msm_accum']['concurrent_ms_per_step'],3), 'combine', round(k['msm_combine']['ms_per_step'],3), 'reduce', round(k['msm_bucket_reduce']['ms_per_step'],3), 'horner', round(k['msm_horner']['ms_per_step'],3))" || { echo "$v failed"; tail -3 gpurun_out/ab.err; }
  done
done
