#!/usr/bin/env python
"""Copy the summaries of tools/capture_profiles.sh from gpurun_out/ into profiles/ (tracked) and
reduce the PMC passes to per-launch HBM traffic of each kernel (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read ->
doubled; WRITE_SIZE is uncalibrated and kept as reported)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def one(pattern):
    """the file of the LAST capture's main process: gpurun merges into gpurun_out/ without deleting what earlier captures left,
    and a traced command that spawns processes (the bench's env workers) leaves one file per pid -- newest run, largest file"""
    f = glob.glob(os.path.join(go, pattern), recursive=True)
    if not f:
        return None
    newest = max(os.path.getmtime(x) for x in f)
    f = [x for x in f if newest - os.path.getmtime(x) < 1200]
    return max(f, key=os.path.getsize)


for src, dst in ((f"{tag}_bench.json", f"{tag}_bench.json"), (f"{tag}_bench_profiled.json", f"{tag}_bench_profiled.json"),
                 (f"{tag}_bench_sac.json", f"{tag}_bench_sac.json"), (f"{tag}_bench_trust.json", f"{tag}_bench_trust.json"),
                 (f"{tag}_bench_cvpo.json", f"{tag}_bench_cvpo.json"), (f"{tag}_bench_group.json", f"{tag}_bench_group.json"),
                 (f"{tag}_bench_group_hostreset.json", f"{tag}_bench_group_hostreset.json"),
                 (f"{tag}_bench_group_noclip.json", f"{tag}_bench_group_noclip.json"),
                 (f"{tag}_learning_curves.json", f"{tag}_learning_curves.json"),
                 (f"{tag}_learning_curves_pointcircle.json", f"{tag}_learning_curves_pointcircle.json"), (f"{tag}_bench_shmem.json", f"{tag}_bench_shmem.json"),
                 (f"{tag}_ubench_gridsync.txt", f"{tag}_ubench_gridsync.txt"), (f"{tag}_ubench_dispatch.txt", f"{tag}_ubench_dispatch.txt"),
                 (f"{tag}_ubench_chain.txt", f"{tag}_ubench_chain.txt"), (f"{tag}_ubench_mfma_pat.txt", f"{tag}_ubench_mfma_pat.txt"),
                 (f"{tag}_ubench_hwid.txt", f"{tag}_ubench_hwid.txt"), (f"{tag}_ab_trust_plans.json", f"{tag}_ab_trust_plans.json"), (f"{tag}_ab_wgrad_plans.json", f"{tag}_ab_wgrad_plans.json"),
                 (f"{tag}_pmc_trust_plans.json", f"{tag}_pmc_trust_plans.json")):
    if os.path.exists(os.path.join(go, src)):
        shutil.copy(os.path.join(go, src), os.path.join(pr, dst))
for sub, dst in ((f"{tag}_prof_bench", f"{tag}_kernel_stats.csv"), (f"{tag}_prof_sac", f"{tag}_sac_kernel_stats.csv"),
                 (f"{tag}_prof_cvpo", f"{tag}_cvpo_kernel_stats.csv"), (f"{tag}_prof_trust", f"{tag}_trust_kernel_stats.csv"),
                 (f"{tag}_prof_group", f"{tag}_group_kernel_stats.csv")):
    f = one(f"{sub}/**/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(pr, dst))
f = one(f"{tag}_prof_bench/**/*_domain_stats.csv")
if f:
    shutil.copy(f, os.path.join(pr, f"{tag}_domain_stats.csv"))

# what the capture ran: the source hash written next to it by capture_profiles.sh (bench.csrc_sha16) and this checkout's HEAD
meta = {}
try:
    import subprocess
    meta["head"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], text=True).strip()
except Exception:          # noqa: BLE001
    pass
hf = os.path.join(go, f"{tag}_csrc_sha16.txt")
if os.path.exists(hf):
    meta["csrc_sha16"] = open(hf).read().strip()
    sys.path.insert(0, root)
    from bench import csrc_sha16
    meta["csrc_sha16_of_this_checkout"] = csrc_sha16()

traffic = defaultdict(lambda: {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0})
for sub, key in ((f"{tag}_pmc_fetch", "fetch_kib"), (f"{tag}_pmc_write", "write_kib")):
    f = one(f"{sub}/**/*_counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        traffic[name][key] += float(r["Counter_Value"])
        if key == "fetch_kib":
            traffic[name]["launches"] += 1
out = {}
for name, t in traffic.items():
    n = max(t["launches"], 1)
    wl = t["launches"] if t["launches"] else 1
    out[name] = {"launches": t["launches"], "fetch_bytes_per_launch_reported": t["fetch_kib"] * 1024 / n,
                 "fetch_bytes_per_launch_x2_gfx950": 2 * t["fetch_kib"] * 1024 / n,
                 "write_bytes_per_launch_reported": t["write_kib"] * 1024 / wl,
                 "hbm_bytes_per_launch": (2 * t["fetch_kib"] + t["write_kib"]) * 1024 / n}
if out:
    print(json.dumps({k: round(v["hbm_bytes_per_launch"]) for k, v in out.items()}, indent=1))
    out["_meta"] = meta
    json.dump(out, open(os.path.join(pr, f"{tag}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)

# ---- HBM-side bytes per UPDATE of the full-batch / replay paths: every kernel's FETCH_SIZE x 2 + WRITE_SIZE summed over
#      the run, divided by the updates the traced command made (bench_trust.py: 1 warm-up + 5 timed; bench_sac.py: 200 + 3 x 200;
#      cpo_stream = CPO with the one-pass streaming weight-gradient kernel, fsrl_tr_set_plan(wgrad = 3))
upd = {}
for alg, n_upd in (("cpo", 6), ("trpo", 6), ("sac", 800), ("cpo_stream", 6), ("cpo_splitk", 6)):     # bench_sac.py: 200 warm-up + 3 blocks of --updates 200
    by = defaultdict(lambda: {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0})
    ok = False
    for cname, key in (("FETCH_SIZE", "fetch_kib"), ("WRITE_SIZE", "write_kib")):
        f = one(f"{tag}_pmc_{cname}_{alg}/**/*_counter_collection.csv")
        if not f:
            continue
        ok = True
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            by[name][key] += float(r["Counter_Value"])
            if key == "fetch_kib":
                by[name]["launches"] += 1
    if not ok:
        continue
    tot = sum((2 * t["fetch_kib"] + t["write_kib"]) * 1024 for t in by.values())
    upd[alg] = {"updates_in_run": n_upd, "hbm_bytes_per_update": tot / n_upd,
                "by_kernel_bytes_per_launch": {k: (2 * t["fetch_kib"] + t["write_kib"]) * 1024 / max(t["launches"], 1)
                                               for k, t in sorted(by.items(), key=lambda kv: -(2 * kv[1]["fetch_kib"] + kv[1]["write_kib"]))[:12]},
                "launches_per_update": {k: t["launches"] / n_upd for k, t in by.items() if t["launches"] >= n_upd}}
if upd:
    print(json.dumps({k: round(v["hbm_bytes_per_update"] / 1e6, 1) for k, v in upd.items()}), "MB per update")
    upd["_meta"] = meta
    json.dump(upd, open(os.path.join(pr, f"{tag}_pmc_traffic_updates.json"), "w"), indent=1, sort_keys=True)

# ---- MFMA utilisation per kernel from the SQ counter passes (one dispatch = one row per counter):
#      util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x SIMDs) -- rocprofv3's own MfmaUtil expression, 256 CUs x 4
#      SIMDs; SQ_VALU_MFMA_BUSY_CYCLES is the sum over all SIMDs of the cycles their matrix pipe was busy (K1: 192
#      workgroups x 16 waves x 128 v_mfma_f32_4x4x1 x 8 cycles = 3 145 728, the reported value to the digit) --
#      and the fp32 MFMA FLOPs the hardware counted per launch (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512).
SIMDS, XCDS = 256 * 4, 8      # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (checked: = 8 x kernel duration x clock),
                                # the MfmaUtil expression takes one instance (reduce(max)): divide by 8
mf = {}
for sub in (f"{tag}_pmc_mfma", f"{tag}_pmc_mfma_trust", f"{tag}_pmc_mfma_sac"):
    f = one(f"{sub}/**/*_counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[name].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    for name, c in acc.items():
        n = max(len(disp[name]), 1)
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        row = {"launches": n, "mfma_busy_cycles_per_launch": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n,
                    "gpu_active_cycles_per_launch": gui / n,
                    "mfma_util_pct": 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * SIMDS) if gui else None,
                    "mfma_f32_flops_per_launch": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512 / n,
                    "sq_busy_cycles_per_launch": c.get("SQ_BUSY_CYCLES", 0.0) / n, "source": sub}
        # one row per (kernel, source run): `name@headline` / `name@trust` / `name@sac` -- the same kernel template runs at other
        # shapes in other updates (fb_wgrad_kernel<256,false>: CPO's critic step over 20 000 rows vs SAC's 1 024-row batch);
        # the bare name keeps the first source that has it (what earlier rounds' files held)
        short = sub[len(tag) + 1:].replace("pmc_mfma_", "").replace("pmc_mfma", "headline")
        mf[f"{name}@{short}"] = row
        mf.setdefault(name, row)
if mf:
    json.dump(mf, open(os.path.join(pr, f"{tag}_pmc_mfma.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: (round(v["mfma_util_pct"], 2) if v["mfma_util_pct"] is not None else None) for k, v in mf.items()
                      if v["mfma_f32_flops_per_launch"] > 0}, indent=1))
