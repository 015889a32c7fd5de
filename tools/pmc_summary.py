#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs into profiles/pmc_traffic.json
(per-kernel average KB per launch).  usage: pmc_summary.py <fetch_csv> <write_csv> [out_json]"""
import collections, csv, json, sys


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


rd, wr = agg(sys.argv[1]), agg(sys.argv[2])
out = {k: {"launches": rd[k][0], "fetch_kb_per_launch": rd[k][1] / rd[k][0], "write_kb_per_launch": wr[k][1] / max(wr[k][0], 1)} for k in rd}
path = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic.json"
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print("wrote", path, len(out), "kernels")
