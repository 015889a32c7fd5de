run() { timeout -k 10 120 python bench.py --steps 16 --warmup 3 --no-cpu-baseline $2 > gpurun_out/sw.log 2>&1 || { echo "$1 FAILED/timeout"; tail -3 gpurun_out/sw.log; return 0; }
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/sw.log').read().strip().split('\n')[-1])
k=d['kernels']
print(sys.argv[1], round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],2), d['aggregate_fr'][:14], 'launches', sum(v['launches_per_step'] for v in k.values()))
" "$1"; }
run "groups pipelined" ""
run "groups nopipe" "--no-pipeline"
GPU_MAX_HW_QUEUES=8 run "groups pipelined hwq8" ""
ITERS=4 timeout -k 10 200 python tools/host_probe.py 2>&1 | grep "iter [12]" | cut -c1-220
ZKDL_PROF_TRACE=gpurun_out/trace_g.csv timeout -k 10 200 python tools/trace_probe.py
