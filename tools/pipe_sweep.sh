run() { timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $2 > gpurun_out/sw.log 2>&1 || { echo "$1 FAILED/timeout"; tail -3 gpurun_out/sw.log; return 0; }
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/sw.log').read().strip().split('\n')[-1])
k=d['kernels']
print(sys.argv[1], round(d['ms_per_step'],2), d['aggregate_fr'][:14], {n:round(k[n]['ms_per_step'],3) for n in k if 'round' in n})
" "$1"; }
run "fact" ""
ZKDL_SC_NOFACT=1 run "nofact" ""
run "fact" ""
ZKDL_SC_NOFACT=1 run "nofact" ""
