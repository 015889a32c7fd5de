#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs (one row per dispatch and counter) into one JSON:
per kernel, the number of dispatches seen and the AVERAGE value per launch of every counter, plus the derived
figures bench.py and DESIGN.md quote.  usage: pmc_summary.py <dir with pmc_*/ subdirs> [out_json] [commit the library was built from]

Derived (gfx950: SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles, GRBM_GUI_ACTIVE is summed over
the 8 XCDs -- MI355X_MICROARCH.md, rocprofv3 PMC slots):
  valu_share_of_wave   = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES  (share of resident-wave time spent issuing vector ALU work)
  valu_insts_per_wave  = SQ_INSTS_VALU / SQ_WAVES
  fetch_kb_per_launch / write_kb_per_launch (FETCH_SIZE / WRITE_SIZE are KB; gfx950: double FETCH_SIZE for 16 B/lane streams)
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    out_path = sys.argv[2] if len(sys.argv) > 2 else "profiles/pmc_r04.json"
    commit = sys.argv[3] if len(sys.argv) > 3 else "unknown"
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            c = acc[k][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
    kernels = {}
    for k, cs in acc.items():
        rec = {"launches": max(v[0] for v in cs.values())}
        for name, (n, tot) in cs.items():
            rec[name] = tot / max(n, 1)
        if "FETCH_SIZE" in rec:
            rec["fetch_kb_per_launch"] = rec.pop("FETCH_SIZE")
        if "WRITE_SIZE" in rec:
            rec["write_kb_per_launch"] = rec.pop("WRITE_SIZE")
        if rec.get("SQ_WAVE_CYCLES"):
            rec["valu_share_of_wave"] = rec.get("SQ_ACTIVE_INST_VALU", 0.0) / rec["SQ_WAVE_CYCLES"]
            rec["wait_share_of_wave"] = rec.get("SQ_WAIT_ANY", 0.0) / rec["SQ_WAVE_CYCLES"]
            rec["issue_stall_share_of_wave"] = rec.get("SQ_WAIT_INST_ANY", 0.0) / rec["SQ_WAVE_CYCLES"]
        if rec.get("SQ_WAVES"):
            rec["valu_insts_per_wave"] = rec.get("SQ_INSTS_VALU", 0.0) / rec["SQ_WAVES"]
        if rec.get("SQ_BUSY_CYCLES"):
            rec["valu_active_per_busy_cycle"] = rec.get("SQ_ACTIVE_INST_VALU", 0.0) / rec["SQ_BUSY_CYCLES"]
        kernels[k] = rec
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from src_fingerprint import source_fingerprint
    json.dump({"src_sha256": source_fingerprint(), "source": "rocprofv3 --pmc passes of `bench.py --serial --steps 2 --warmup 1` (tools/collect_profiles.sh), averaged per launch",
               "commit": commit, "kernels": kernels}, open(out_path, "w"), indent=1, sort_keys=True)
    print("wrote", out_path, len(kernels), "kernels")


if __name__ == "__main__":
    main()
