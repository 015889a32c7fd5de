#!/bin/bash
# Fr-only / both-phases ms per proof.  Needs the timing-experiment build of the library first (here, before gpurun): make -C zkdl_amd/csrc variant_exp
# The shipping library is never touched: the loader is pointed at the experiment build through ZKDL_AMD_LIB (zkdl_amd/capi.py).
set -e
EXP=$PWD/zkdl_amd/lib/variants/lib_exp.so
[ -f "$EXP" ] || { echo "missing $EXP: run  make -C zkdl_amd/csrc variant_exp  first"; exit 1; }
run() { echo -n "$1: "; shift; env ZKDL_AMD_LIB=$EXP "$@" timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe --steady-steps 0 --latency-reps 0 --ext-steps 0 --prof-steps 1 --steps 60 --warmup 6 --detail-out /dev/null 2>>gpurun_out/ab.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
run both X=1
run fr_only ZKDL_EXPERIMENT_SKIP_G1=1
run g1_only ZKDL_EXPERIMENT_SKIP_FR=1
run fr_only_again ZKDL_EXPERIMENT_SKIP_G1=1
