#!/usr/bin/env python3
"""Per-stream timeline of ONE proof step out of a rocprofv3 --kernel-trace CSV.

usage: trace_timeline.py <kernel_trace.csv> [step_index_from_end=1]
The step window is delimited by k_digest dispatches (one per bench step)."""
import csv, sys, re, collections

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
dig = [i for i, r in enumerate(rows) if "k_digest" in r["Kernel_Name"]]
hi = dig[-back]; lo = dig[-back - 1] + 1
win = rows[lo:hi + 1]
t0 = min(r["s"] for r in win); t1 = max(r["e"] for r in win)
print(f"step window {(t1 - t0) / 1e6:.3f} ms, {len(win)} dispatches")
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("zk::", ""))[:40]
by = collections.defaultdict(list)
for r in win: by[r["Stream_Id"]].append(r)
print("stream  n   first_ms  last_ms  busy_ms  top kernels")
for sid, rs in sorted(by.items(), key=lambda kv: int(kv[0])):
    busy = sum(r["e"] - r["s"] for r in rs) / 1e6
    top = collections.Counter()
    for r in rs: top[short(r["Kernel_Name"])] += (r["e"] - r["s"]) / 1e6
    tops = ", ".join(f"{k}={v:.2f}" for k, v in top.most_common(4))
    print(f"{sid:>5} {len(rs):4d} {(min(r['s'] for r in rs) - t0) / 1e6:8.3f} {(max(r['e'] for r in rs) - t0) / 1e6:8.3f} {busy:8.3f}  {tops}")
# concurrency profile in 0.5 ms bins
ev = []
for r in win: ev += [(r["s"], 1), (r["e"], -1)]
ev.sort()
binw = 500_000; nb = (t1 - t0) // binw + 1
acc = [0.0] * nb; cur = 0; last = t0
for t, d in ev:
    while last < t:
        b = (last - t0) // binw; nxt = min(t, t0 + (b + 1) * binw)
        acc[b] += cur * (nxt - last); last = nxt
    cur += d
print("avg kernels in flight per 0.5 ms bin:")
print(" ".join(f"{a / binw:.1f}" for a in acc))
if len(sys.argv) > 3:
    sid = sys.argv[3]
    prev = None
    for r in by[sid]:
        gap = (r["s"] - prev) / 1e3 if prev else 0
        print(f"  {(r['s'] - t0) / 1e6:8.3f} +{(r['e'] - r['s']) / 1e3:8.1f}us gap {gap:7.1f}us grid {r['Grid_Size_X']:>9} {short(r['Kernel_Name'])}")
        prev = r["e"]
