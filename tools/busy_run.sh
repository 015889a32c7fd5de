#!/bin/bash
# kernel trace of the default (concurrent) bench + per-queue busy table
export TMPDIR=/tmp
out=$PWD/gpurun_out/busy; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -o kt -- python3 bench.py --cpu-baseline off --no-msm-probe --no-prewarm --steady-steps 0 --ext-steps 0 --latency-reps 0 --steps 40 --warmup 5 > $out/log 2>&1
python3 tools/stream_busy.py $out/kt_kernel_trace.csv | head -60
