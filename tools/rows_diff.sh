#!/bin/bash
# run the default bench N times, dump the per-proof digests and report which proofs differ between runs
N=${1:-4}; shift
for i in $(seq 1 $N); do
  env "$@" ZKDL_BENCH_DUMP_ROWS=gpurun_out/rows_$i.npy timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe --steps 200 --warmup 3 > gpurun_out/rows_$i.json 2>gpurun_out/rows.err || tail -3 gpurun_out/rows.err
done
python3 - <<PY
import numpy as np
rows=[np.load("gpurun_out/rows_%d.npy"%i) for i in range(1,$N+1)]
for i in range(1,$N):
    d=np.where((rows[i]!=rows[0]).any(axis=1))[0]
    print("run", i+1, "vs run 1: proofs that differ:", d.tolist())
PY
