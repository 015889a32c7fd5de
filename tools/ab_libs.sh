#!/bin/bash
# same-box A/B of library builds: zkdl_amd/lib/variants/lib_<name>.so (make -C zkdl_amd/csrc variant NAME=<name> DEFS="-D..."), selected through
# ZKDL_AMD_LIB (the shipping library is never touched); "ship" = the shipping library.  usage: tools/ab_libs.sh ship subs2 ship subs2   [STEPS=60] [FLAGS=..]
set -e
for v in "$@"; do
  if [ "$v" = ship ]; then lib=""; else lib=$PWD/zkdl_amd/lib/variants/lib_$v.so; [ -f "$lib" ] || { echo "missing $lib"; exit 1; }; fi
  env ZKDL_AMD_LIB=$lib timeout -k 10 250 python bench.py --cpu-baseline off --no-msm-probe --ext-steps 0 --steps ${STEPS:-60} --warmup 5 --detail-out gpurun_out/ab_detail.json $FLAGS 2>gpurun_out/ab.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=json.load(open('gpurun_out/ab_detail.json'))['kernels']
top=sorted(k.items(), key=lambda kv: -kv[1]['ms_per_step'])[:6]
print('$v', 'K=${STEPS:-60}', round(d['ms_per_step'],3), 'steady', round(d.get('steady_state_ms_per_proof') or 0,3), 'lat', round(d.get('latency_ms_per_proof') or 0,3), d['aggregate_fr'][:12], ' '.join('%s=%.3f' % (n, v['ms_per_step']) for n, v in top))" || { echo "$v failed"; tail -3 gpurun_out/ab.err; }
done
