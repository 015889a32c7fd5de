#!/bin/bash
# same-box A/B of library builds in zkdl_amd/lib/variants/lib_<name>.so (the tree on the GPU box is a scratch copy)
for v in "$@"; do
  cp zkdl_amd/lib/variants/lib_$v.so zkdl_amd/lib/libzkdl_amd.so
  timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe --steps ${STEPS:-100} --warmup 5 $FLAGS 2>gpurun_out/ab.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$v', 'K=${STEPS:-100}', round(d['ms_per_step'],3), d['aggregate_fr'][:12], 'binp_words', round(k['binp_words']['ms_per_step'],3), 'conc', round(k['binp_words']['concurrent_ms_per_step'],3))" || { echo "$v failed"; tail -3 gpurun_out/ab.err; }
done
