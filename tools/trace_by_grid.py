#!/usr/bin/env python
"""kernel-trace CSV -> mean duration per (kernel, grid size): usage trace_by_grid.py <dir> [name-filter ...]"""
import collections, csv, glob, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if len(sys.argv) > 2 and not any(k in n for k in sys.argv[2:]):
            continue
        g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0); w = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        acc[(n, g // max(w, 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in sorted(acc.items()):
    v = sorted(v)
    print(f"{n[:44]:44s} blocks {g:5d}  n {len(v):4d}  median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
