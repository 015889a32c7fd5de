#!/bin/bash
# Run on the GPU box from the repo root: the bench line + the rocprofv3 summaries that DESIGN.md / bench.py cite.
# usage: tools/collect_profiles.sh <tag>   (outputs under gpurun_out/prof_<tag>/; tools/pmc_summary.py condenses the
# counter CSVs into profiles/pmc_<round>.json, the kernel-stats CSVs are copied into profiles/ by hand)
# Every rocprofv3 pass runs the program directly after `--` (no env / bash -c hop) and uses --pmc ALONE.
set -e
tag=${1:-r04}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# profiler passes: no helper processes (they would inherit the profiler's preload and be traced into the same directory), no side measurements
B="python3 bench.py --detail-out /dev/null --no-cpu-baseline --no-msm-probe --no-prewarm --steady-steps 0 --ext-steps 0 --latency-reps 0 --spinup-ms 0"
timeout -k 10 900 python3 bench.py --detail-out $out/bench_detail.json > $out/bench.json 2> $out/bench.err
echo "bench done"
# every proof of this pass in the TIMED form (derived round commitments: --warmup 1 --drain 0 leave out the msm_rounds-form proofs of the warm-up and of the
# drain), so that the kernel-stats average of a kernel is the average bench.py's roofline entry reports for it
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_serial -o kt -- $B --serial --steps 5 --warmup 1 --drain 0 > $out/kt_serial.log 2>&1
echo "kernel trace (serial) done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_conc -o kt -- $B --steps 5 --warmup 2 > $out/kt_conc.log 2>&1
echo "kernel trace (concurrent) done"
pass() {  # name, counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -o pmc -- $B --serial --steps 2 --warmup 1 > $out/pmc_$name.log 2>&1 || echo "pmc pass $name FAILED (see $out/pmc_$name.log)"
  echo "pmc $name done"
}
pass rd FETCH_SIZE
pass wr WRITE_SIZE
# instruction-issue view (SQ block has 8 slots, GRBM 2): how busy the vector ALUs are while the kernel runs
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
find $out -name "*.csv" | head -40
