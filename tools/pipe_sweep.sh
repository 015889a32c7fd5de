run() { timeout -k 10 300 python bench.py --steps $3 --warmup 3 --no-cpu-baseline $2 > gpurun_out/sw.log 2>&1 || { echo "$1 FAILED/timeout"; tail -3 gpurun_out/sw.log; return 0; }
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/sw.log').read().strip().split('\n')[-1])
k=d['kernels']
print(sys.argv[1], round(d['ms_per_step'],2), d['aggregate_fr'][:14], {n[4:]:round(k[n]['ms_per_step'],2) for n in k if n.startswith('msm_') or n.startswith('g1_c')})
" "$1"; }
for c in 10 11 12 9 8 7; do run "c=$c" "--msm-window $c" 20; done
