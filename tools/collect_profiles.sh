#!/bin/bash
# Run on the GPU box from the repo root: bench line + rocprofv3 summaries that DESIGN.md / bench.py cite.
# usage: tools/collect_profiles.sh <tag>      (outputs under gpurun_out/prof_<tag>/, copied into profiles/ by hand)
set -e
tag=${1:-r01}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_serial -o kt -- python3 bench.py --serial --steps 5 --warmup 2 --no-cpu-baseline > $out/kt_serial.log 2>&1
echo "kernel trace (serial) done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_conc -o kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/kt_conc.log 2>&1
echo "kernel trace (concurrent) done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_rd -o pmc -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_rd.log 2>&1
echo "pmc fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_wr -o pmc -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_wr.log 2>&1
echo "pmc write done"
find $out -name "*.csv" | head -20
