#!/bin/bash
# quick counter passes (VALU issue + LDS) of the serial bench; usage: tools/pmc_quick.sh <tag>
tag=${1:-q}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --cpu-baseline off --no-msm-probe --no-prewarm --steady-steps 0 --ext-steps 0 --latency-reps 0 --spinup-ms 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_serial -o kt -- $B --serial --steps 5 --warmup 2 > $out/kt_serial.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_valu -o pmc -- $B --serial --steps 2 --warmup 1 > $out/pmc_valu.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc_lds -o pmc -- $B --serial --steps 2 --warmup 1 > $out/pmc_lds.log 2>&1
echo done
