#!/bin/bash
# A/B of environment switches on the default bench: tools/ab_env.sh "VAR=1" "" "VAR=1" ...   (empty string = defaults)
for v in "$@"; do
  env $v timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe --steps 100 --warmup 5 2>gpurun_out/ab.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
def g(n): return round(k[n]['ms_per_step'],3) if n in k else None
print('[$v]', round(d['ms_per_step'],3), 'bin', g('sumcheck_bin_round'), 'hp', g('sumcheck_hp_round'), 'ip', g('sumcheck_ip_round'), 'prep', g('sumcheck_prep'), 'accum', g('msm_accum'), d['aggregate_fr'][:14])" || { echo "$v failed"; tail -3 gpurun_out/ab.err; }
done
