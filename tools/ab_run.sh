#!/bin/bash
# A/B on ONE box (boxes differ by several per cent): tools/ab_run.sh "ENV=.. ENV=.. ----bench flags" ...   ("" = defaults), K = 100, unpaired
# e.g.  gpurun -- 'bash tools/ab_run.sh "" "ZKDL_G1_STREAMS=3 ----inflight 5" ""'
for spec in "$@"; do
  envs="${spec%%--*}"; flags="${spec#*--}"; [ "$flags" = "$spec" ] && flags=""
  env $envs A=1 timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe --steps 100 --warmup 5 --pair 1 $flags 2>gpurun_out/ab.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$spec]', round(d['ms_per_step'],3))" || { echo "$spec failed"; tail -3 gpurun_out/ab.err; }
done
