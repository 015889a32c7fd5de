#!/usr/bin/env python3
"""Per-stream busy time over the steady-state part of a rocprofv3 --kernel-trace CSV of bench.py (concurrent schedule).
usage: stream_busy.py <kernel_trace.csv> [skip_fraction=0.4]   -- looks at the last (1 - skip) of the timeline"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
dig = [r for r in rows if "k_digest" in r["Kernel_Name"]]
n = len(dig)
lo = dig[int(n * skip)]["e"]; hi = dig[-3]["e"]
steps = sum(1 for d in dig if lo < d["e"] <= hi)
win = [r for r in rows if r["s"] >= lo and r["e"] <= hi]
span = (hi - lo) / 1e6
print(f"window {span:.2f} ms, {steps} proofs -> {span / steps:.3f} ms/proof, {len(win)} dispatches")
short = lambda nm: re.sub(r"\(.*", "", nm.replace("void ", "").replace("zk::", ""))
def fam(nm):
    m = re.search(r"k_[a-z0-9_]+", nm)
    return m.group(0) if m else nm[:30]
by = collections.defaultdict(list)
for r in win: by[r["Stream_Id"]].append(r)
print("stream   n   busy_ms/proof  busy%   top kernels (ms/proof)")
for sid, rs in sorted(by.items(), key=lambda kv: -sum(r["e"] - r["s"] for r in kv[1])):
    busy = sum(r["e"] - r["s"] for r in rs) / 1e6
    top = collections.Counter()
    for r in rs: top[fam(r["Kernel_Name"])] += (r["e"] - r["s"]) / 1e6 / steps
    print(f"{sid:>5} {len(rs):5d} {busy / steps:10.3f} {100 * busy / span:7.1f}   " + ", ".join(f"{k}={v:.2f}" for k, v in top.most_common(5)))
# union busy (any kernel running) and average concurrency
ev = []
for r in win: ev += [(r["s"], 1), (r["e"], -1)]
ev.sort()
cur = 0; last = lo; anyb = 0; conc = 0
for t, d in ev:
    if cur > 0: anyb += t - last
    conc += cur * (t - last); last = t; cur += d
print(f"chip has >= 1 kernel running {100 * anyb / (hi - lo):.1f} % of the time; average kernels in flight {conc / (hi - lo):.2f}")
