#!/usr/bin/env python3
"""Summarise a ZKDL_PROF_TRACE file (name,stream,start_us,dur_us,bytes): per-stream span/busy time,
kernels in flight per 0.25 ms bin, and (optionally) the full listing of one stream.
usage: prof_timeline.py trace.csv [stream_index]"""
import sys, collections
rows = []
for ln in open(sys.argv[1]):
    n, s, t, d, b = ln.rstrip().rsplit(",", 4)
    rows.append((n, int(s), float(t), float(d), int(b)))
t0 = min(r[2] for r in rows); t1 = max(r[2] + r[3] for r in rows)
print(f"window {(t1 - t0) / 1e3:.3f} ms, {len(rows)} launches")
by = collections.defaultdict(list)
for r in rows: by[r[1]].append(r)
print("stream   n  first_ms  last_ms  busy_ms  top kernels")
for sid in sorted(by):
    rs = by[sid]
    top = collections.Counter()
    for r in rs: top[r[0]] += r[3] / 1e3
    print(f"{sid:>5} {len(rs):4d} {(min(r[2] for r in rs) - t0) / 1e3:8.3f} {(max(r[2] + r[3] for r in rs) - t0) / 1e3:8.3f} {sum(r[3] for r in rs) / 1e3:8.3f}  "
          + ", ".join(f"{k}={v:.2f}" for k, v in top.most_common(4)))
binw = 250.0; nb = int((t1 - t0) // binw) + 1
acc = [0.0] * nb
for r in rows:
    a, e = r[2] - t0, r[2] - t0 + r[3]
    b = int(a // binw)
    while a < e:
        nxt = min(e, (b + 1) * binw); acc[b] += nxt - a; a = nxt; b += 1
print("kernels in flight per 0.25 ms bin:"); print(" ".join(f"{x / binw:.1f}" for x in acc))
if len(sys.argv) > 2:
    prev = None
    for r in sorted(by[int(sys.argv[2])], key=lambda r: r[2]):
        gap = r[2] - prev if prev is not None else 0.0
        print(f"  {(r[2] - t0) / 1e3:8.3f} ms +{r[3]:8.1f} us gap {gap:7.1f} us  {r[4]:>10} B  {r[0]}")
        prev = r[2] + r[3]
