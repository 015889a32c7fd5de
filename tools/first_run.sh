#!/bin/bash
# is the first bench.py process on a box slower than the next ones?  (K = 20, the driver's run)  usage: tools/first_run.sh [n] [extra bench flags]
n=${1:-4}; shift
for i in $(seq 1 $n); do
  timeout -k 10 300 python bench.py --cpu-baseline off --no-msm-probe --steps 20 --warmup 5 "$@" 2>gpurun_out/ab.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('run $i', round(d['ms_per_step'],3), 'host enqueue', round(d['host_enqueue_ms_per_step'],3), 'setup s', {a: round(b,2) for a,b in d['setup'].items() if isinstance(b,float)} , 'conc accum', round(k['msm_accum']['concurrent_ms_per_step'],3), 'conc binp', round(k['binp_words']['concurrent_ms_per_step'],3))"
done
