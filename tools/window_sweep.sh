#!/bin/bash
# Pippenger window-width sweep on the bench workload (BASELINE configs[4] asks for it): ms/proof and the MSM kernels
# usage: tools/window_sweep.sh [workload]
wl=${1:-mlp18m_b256}
for c in 7 8 9 10 11 12 13; do
  timeout -k 10 300 python3 bench.py --workload $wl --steps 12 --warmup 3 --no-cpu-baseline --msm-window $c > gpurun_out/ws.log 2>&1 || { echo "c=$c failed"; continue; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ws.log').read().strip().split('\n')[-1])
k=d['kernels']
print('c=%2d  %.2f ms/proof  ' % (int(sys.argv[1]), d['ms_per_step']) + '  '.join('%s %.2f' % (n[4:], k[n]['ms_per_step']) for n in ('msm_sort','msm_accum','msm_combine','msm_bucket_reduce','msm_horner')))
" $c
done
