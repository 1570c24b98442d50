#!/usr/bin/env python
"""kernel-trace CSV -> the repeating launch sequence of an update: per position in the period, median duration and median gap to
the previous launch's end.  usage: trace_timeline.py <dir> <period (launches per update)> [anchor kernel substring]

The period is anchored at the LAST `period` x k launches of the trace (the timed region of the tools/bench_*.py scripts), or at the
launches that follow each occurrence of the anchor kernel."""
import csv
import glob
import statistics
import sys


def main():
    d, period = sys.argv[1], int(sys.argv[2])
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0].replace("void ", "")
            g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
            w = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, g // max(w, 1)))
    rows.sort()
    k = min(200, len(rows) // period - 2)
    tail = rows[len(rows) - k * period:]
    # align so that position 0 is the most common kernel name at a period boundary
    best, off = -1, 0
    for o in range(period):
        names = [tail[o + i * period][2] for i in range(k - 1)]
        same = max(names.count(x) for x in set(names))
        key = (same, -o)
        if same > best:
            best, off = same, o
    if len(sys.argv) > 3:
        for o in range(period):
            if sys.argv[3] in tail[o][2]:
                off = o
                break
    dur = [[] for _ in range(period)]
    gap = [[] for _ in range(period)]
    name = [None] * period
    for i in range(k - 1):
        for p in range(period):
            j = off + i * period + p
            s, e, n, g = tail[j]
            name[p] = (n, g)
            dur[p].append((e - s) / 1e3)
            gap[p].append((s - tail[j - 1][1]) / 1e3 if j > 0 else 0.0)
    tot = 0.0
    for p in range(period):
        md, mg = statistics.median(dur[p]), statistics.median(gap[p])
        tot += md + mg
        print(f"{p:2d} {name[p][0][:46]:46s} blocks {name[p][1]:5d}  dur {md:7.2f} us  gap before {mg:6.2f} us")
    print(f"period total {tot:.1f} us over {k - 1} periods")


if __name__ == "__main__":
    main()
