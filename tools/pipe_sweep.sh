run() { timeout -k 10 200 python bench.py --steps 16 --warmup 3 --no-cpu-baseline $2 > gpurun_out/sw.log 2>&1 || { tail -5 gpurun_out/sw.log; exit 1; }
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/sw.log').read().strip().split('\n')[-1])
print(sys.argv[1], round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],2), d['aggregate_fr'][:18])
" "$1"; }
run "pipelined" ""
run "no-pipeline" "--no-pipeline"
ZKDL_LANES=12 run "pipelined lanes=12" ""
ZKDL_LANES=16 run "pipelined lanes=16" ""
