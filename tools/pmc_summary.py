#!/usr/bin/env python
"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files (run on the GPU box: the raw files of a few thousand
dispatches exceed what gpurun copies back).  usage: pmc_summary.py <out.json> <dir> [<dir> ...]"""
import collections
import csv
import glob
import json
import sys

out, dirs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[n][r["Counter_Name"]] += 1
res = {n: {c: {"mean": acc[n][c] / cnt[n][c], "launches": cnt[n][c]} for c in acc[n]} for n in acc}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(len(res), "kernels")
