#!/bin/bash
# VALU instruction counts per kernel family for env variants: tools/pmc_valu.sh "" "ZKDL_SC_OLD=1"
export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1)); out=$PWD/gpurun_out/pmcv_$i; rm -rf $out; mkdir -p $out
  ( [ -n "$v" ] && export $v; timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out -o pmc -- python3 bench.py --cpu-baseline off --no-msm-probe --no-prewarm --steady-steps 0 --ext-steps 0 --latency-reps 0 --spinup-ms 0 --serial --steps 2 --warmup 1 > $out/log 2>&1 )
  python3 - "$out" "$v" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
tot = collections.Counter(); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] != "SQ_INSTS_VALU": continue
        k = r["Kernel_Name"]
        for key in ("k_sc_round29ILi0", "k_sc_round29ILi1", "k_sc_round29ILi2", "k_sc_roundILi0", "k_sc_roundILi1", "k_sc_roundILi2", "k_sc_round_hp0", "k_sc_prep", "k_binb_words", "k_msm_accum", "k_sc_tail"):
            if key in k:
                tot[key] += float(r["Counter_Value"]); n[key] += 1
print("[%s]" % tag, {k: (round(v / 1e6, 1), n[k]) for k, v in sorted(tot.items())}, "all (M):", round(sum(tot.values()) / 1e6, 1))
PY
done
