#!/usr/bin/env python
"""Idle gaps of a rocprofv3 kernel trace, by (kernel before, kernel after): python tools/diag/gaps.py <kernel_trace.csv> [min launches]
(where the GPU waits for the host: read-backs, host-side decisions, enqueue latency)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]) for r in rows)
segs, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur[-50:]) > 1_000_000:
        segs.append(cur); cur = []
    cur.append(e)
segs.append(cur)
for s in segs:
    if len(s) < (int(sys.argv[2]) if len(sys.argv) > 2 else 1000):
        continue
    gaps, ce, prev = [], s[0][1], s[0][2]
    for a, b, n in s[1:]:
        if a > ce:
            gaps.append(((a - ce) / 1e3, prev, n))
        if b > ce:
            ce, prev = b, n
    wall = (ce - s[0][0]) / 1e6
    print(f"segment of {len(s)} launches: wall {wall:.2f} ms, idle {sum(g[0] for g in gaps) / 1e3:.2f} ms in {len(gaps)} gaps")
    c, k = collections.Counter(), collections.Counter()
    for g, p, n in gaps:
        c[(p, n)] += g; k[(p, n)] += 1
    for key, v in c.most_common(10):
        print(f"   {v / 1e3:6.2f} ms  {k[key]:4d} x {v / k[key]:6.1f} us   {key[0]} -> {key[1]}")
