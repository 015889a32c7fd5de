#!/bin/bash
# Fr-only / both-phases ms per proof.  Needs the timing-experiment build of the library first (here, before gpurun): make -C zkdl_amd/csrc variant_exp
cp zkdl_amd/lib/libzkdl_amd.so /tmp/lib_ship.so
cp zkdl_amd/lib/variants/lib_exp.so zkdl_amd/lib/libzkdl_amd.so
run() { echo -n "$1: "; shift; env "$@" timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe --steady-steps 0 --latency-reps 0 --ext-steps 0 --prof-steps 1 --steps 60 --warmup 6 2>>gpurun_out/ab.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
run both X=1
run fr_only ZKDL_EXPERIMENT_SKIP_G1=1
run g1_only ZKDL_EXPERIMENT_SKIP_FR=1
run fr_only_again ZKDL_EXPERIMENT_SKIP_G1=1
cp /tmp/lib_ship.so zkdl_amd/lib/libzkdl_amd.so
