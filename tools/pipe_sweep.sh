#!/bin/bash
# A/B helper used while tuning: run bench.py under a few environment settings and print ms/proof.
# usage (on the GPU box): tools/pipe_sweep.sh            # edit the `run` lines below for the knobs to compare
run() { timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $2 > gpurun_out/sw.log 2>&1 || { echo "$1 FAILED/timeout"; tail -3 gpurun_out/sw.log; return 0; }
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/sw.log').read().strip().split('\n')[-1])
print(sys.argv[1], round(d['ms_per_step'],2), 'ms/proof  digest', d['aggregate_fr'][:14])
" "$1"; }
run "default" ""
ZKDL_G1_CU_PERCENT=100 run "G1 streams on all CUs" ""
ZKDL_G1_STREAMS=1 run "one G1 stream" ""
ZKDL_LANES=0 ZKDL_G1_STREAMS=0 run "no worker streams" ""
run "no pipelining" "--no-pipeline"
ZKDL_SC_NOFACT=1 run "sumcheck weights unfactored" ""
