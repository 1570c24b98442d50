#!/usr/bin/env python
"""Same-box A/B of the full-batch kernel plans (round 5): round 4's one-1024-thread-workgroup-per-CU tile / cached-HVP kernels
against the co-resident pairs of kernels_fbco.hpp, CPO on BASELINE configs[2] and TRPO-Lag on the configs[1] shape, alternated
`--rounds` times so that box drift shows.  One JSON line per (algorithm, plan, round) + a summary.

    python tools/ab_trust_co.py [--rounds 2] [--splits]      (under rocprofv3 --kernel-trace --stats for per-kernel times)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_trust  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--splits", action="store_true", help="also sweep forced 32-row tile counts of the co-resident launches")
    ap.add_argument("--only", default="cpo,trpo")
    ap.add_argument("--delays", action="store_true", help="sweep the start offset of every CU's second resident workgroup")
    ap.add_argument("--plan", default=None, help="run ONE plan 'tile_rows,hvp,wgrad' only (profiling runs)")
    ap.add_argument("--wgrad", action="store_true", help="r6: the weight-gradient kernels against each other on the default tile plan")
    a = ap.parse_args()
    plans = {"r4: one workgroup per CU, full R-op, one stream (96,7)": ("96,7,0", None),
             "co + Gauss-Newton, one stream (64,0)": ("64,0,0", None),
             "r4 kernels, critics beside the actor (32,7)": ("32,7,0", None),
             "co + Gauss-Newton + critics beside the actor (0,0) = default": ("0,0,0", None)}
    if a.wgrad:
        plans = {"r5 split-K weight gradients (0,0,2)": ("0,0,2", None),
                 "streaming weight gradients (0,0,3)": ("0,0,3", None),
                 "r6 tile jobs, two per CU, XCD-aware order (0,0,0) = default": ("0,0,0", None),
                 "r6 tile jobs, plain block order (0,0,4)": ("0,0,4", None),
                 "r6 tile jobs, half the row splits (0,0,5)": ("0,0,5", None)}
    if a.plan:
        plans = {f"plan ({a.plan})": (a.plan, None)}
    res = {}
    for rnd in range(a.rounds):
        for kind, od, ep in (("cpo", 60, 1000), ("trpo", 8, 250)):
            if kind not in a.only.split(","):
                continue
            for name, (plan, split) in plans.items():
                os.environ["FSRL_TR_PLAN"] = plan
                os.environ.pop("FSRL_TR_SPLIT", None)
                if split:
                    os.environ["FSRL_TR_SPLIT"] = split
                r = bench_trust.run(kind, od, 2, 256, ep=ep, timed=5, emit=False, no_cpu=True)
                res.setdefault((kind, name), []).append(r["hip_ms_per_update"])
                print(json.dumps({"alg": kind, "plan": name, "round": rnd, "ms": round(r["hip_ms_per_update"], 3)}), flush=True)
    if a.splits:
        os.environ["FSRL_TR_PLAN"] = "0,0,0"
        for kind, od, ep in (("cpo", 60, 1000), ("trpo", 8, 250)):
            if kind not in a.only.split(","):
                continue
            for n32 in (0, 300, 400, 450, 497, 530, 560, 590, 625):
                os.environ["FSRL_TR_SPLIT"] = f"{n32},{n32}"
                r = bench_trust.run(kind, od, 2, 256, ep=ep, timed=5, emit=False, no_cpu=True)
                res.setdefault((kind, f"co split n32={n32}"), []).append(r["hip_ms_per_update"])
                print(json.dumps({"alg": kind, "plan": f"co split n32={n32}", "ms": round(r["hip_ms_per_update"], 3)}), flush=True)
    if a.delays:
        os.environ["FSRL_TR_PLAN"] = "0,0,0"
        os.environ.pop("FSRL_TR_SPLIT", None)
        for kind, od, ep in (("cpo", 60, 1000), ("trpo", 8, 250)):
            if kind not in a.only.split(","):
                continue
            for dt, dh in ((0, 0), (2, 0), (4, 0), (6, 0), (0, 3), (0, 5), (0, 7), (0, 9), (4, 7), (3, 5), (5, 9)):
                os.environ["FSRL_TR_DELAY"] = f"{dt},{dh}"
                r = bench_trust.run(kind, od, 2, 256, ep=ep, timed=5, emit=False, no_cpu=True)
                res.setdefault((kind, f"co delay tile={dt} hvp={dh}"), []).append(r["hip_ms_per_update"])
                print(json.dumps({"alg": kind, "plan": f"co delay tile={dt} hvp={dh}", "ms": round(r["hip_ms_per_update"], 3)}), flush=True)
        os.environ.pop("FSRL_TR_DELAY", None)
    print(json.dumps({"summary_ms_median": {f"{k[0]} | {k[1]}": round(float(np.median(v)), 3) for k, v in res.items()}}, indent=1))


if __name__ == "__main__":
    main()
