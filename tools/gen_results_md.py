#!/usr/bin/env python
"""Regenerate the measured-numbers tables of README.md and DESIGN.md from the committed profile summaries (profiles/<tag>_*),
so that no figure in the documents is typed by hand ("docs drift").

    python tools/gen_results_md.py r03          # rewrites the blocks between <!-- BEGIN GENERATED <tag> --> / <!-- END GENERATED <tag> -->
                                                # in README.md and DESIGN.md, and writes profiles/RESULTS_<tag>.md

Every row names the file it was read from.  Missing inputs leave their rows out."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PR = os.path.join(ROOT, "profiles")


def jl(name):
    """all JSON objects of a file with one object per line (or one object)"""
    f = os.path.join(PR, name)
    if not os.path.exists(f):
        return []
    out = []
    for line in open(f):
        line = line.strip()
        if line.startswith("{"):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


def kernels(name, top=6):
    f = os.path.join(PR, name)
    if not os.path.exists(f):
        return []
    rows = list(csv.DictReader(open(f)))
    return [(r["Name"].split("(")[0].replace("void ", ""), int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
            for r in rows[:top]]


def fmt(x, d=1):
    return "n/a" if x is None else f"{x:,.{d}f}"


def build(tag):
    L = []
    add = L.append
    b = (jl(f"{tag}_bench.json") or [None])[-1]
    if b:
        r = b.get("roofline", {})
        add(f"### Headline: `bench.py` (`profiles/{tag}_bench.json`)\n")
        add("| | value |\n|---|---|")
        add(f"| policy-updates/s (configs[1]: 256x256, N = 20 000, batch 256 x 4 passes, grad-clip 0.5) | **{fmt(b['value'])}** "
            f"({fmt(b['ms_per_step'], 2)} ms per update, {fmt(b['grad_steps_per_s'], 0)} optimiser steps/s) |")
        if isinstance(r, dict) and "frac" in r:
            add(f"| `roofline` of `ppo_fwd_bwd_kernel<256>` | {fmt(r['achieved'], 2)} TFLOP/s of {r['peak']} = **{fmt(r['frac'], 4)}**; "
                f"{fmt(r['avg_launch_us'], 2)} us per launch ({r['launches_timed']} launches, HIP events); HBM-side traffic "
                f"{fmt((r.get('traffic') or 0) / 1e6, 1)} MB per launch ({r.get('traffic_source', 'no PMC pass')}) |")
            if r.get("latency_floor_us"):
                add(f"| latency floor of an optimiser step (3 launch floors + MFMA work at peak) | {fmt(r['latency_floor_us'], 1)} us vs "
                    f"{fmt(r['step_us'], 1)} us measured per step = **{fmt(r['frac_of_latency_floor'], 3)}** of the floor |")
        for key, label in (("cpu_baseline", "CPU port, 4 threads (the reference default)"),
                           ("cpu_baseline_all_cores", "CPU port, all usable cores"),
                           ("cpu_baseline_c0", "CPU port at BASELINE configs[0] (128x128, 4 threads)")):
            c = b.get(key)
            if isinstance(c, dict) and "value" in c:
                add(f"| {label} | {fmt(c['value'], 3)} updates/s on {c['cores']} threads |")
        tb_ = b.get("torch_rocm_baseline")
        if isinstance(tb_, dict) and "value" in tb_:
            add(f"| stock PyTorch on THIS GPU (the reference's only GPU path: torch.nn / torch.optim with device=\"cuda\", ROCm torch; context, not a target) | "
                f"{fmt(tb_['value'], 2)} updates/s ({fmt(tb_.get('us_per_grad_step'), 0)} us per optimiser step); the HIP path: {fmt(b['value'] / tb_['value'], 0)}x |")
        if isinstance(b.get("gpu_c0"), dict) and "value" in b["gpu_c0"]:
            add(f"| HIP path at the configs[0] shape (128x128) | {fmt(b['gpu_c0']['value'])} updates/s ({fmt(b['gpu_c0']['us_per_step'], 1)} us per step) |")
        if "speedup_vs_cpu_port" in b:
            add(f"| HIP / CPU port (4 threads) | {fmt(b['speedup_vs_cpu_port'], 0)}x |")
        k = b.get("kl_on")
        if isinstance(k, dict) and "value" in k:
            add(f"| the same update with the KL early stop ON (target_kl 0.02, the reference default; SURVEY 8d \"report both\") | "
                f"{fmt(k['value'])} updates/s, {fmt(k['grad_steps_per_update_mean'], 0)} optimiser steps per update, "
                f"{k['updates_stopped_early']} of {k['updates']} updates stopped early |")
        for key, label in (("cpo_c2", "leg `cpo_c2`: CPO, BASELINE configs[2] (obs 60, 256x256, N = 20 000, CG 10, 4 repeats)"),
                           ("trpo_c1", "leg `trpo_c1`: TRPO-Lag on the configs[1] shape (N = 20 000 full batch)")):
            t = b.get(key)
            if isinstance(t, dict) and "hip_ms_per_update" in t:
                rf, cb = t.get("roofline", {}), t.get("cpu_baseline", {})
                add(f"| {label} | {fmt(t['hip_ms_per_update'], 1)} ms per update = {fmt(rf.get('frac'), 3)} of the fp32 MFMA peak over the whole update; "
                    f"CPU port {fmt(cb.get('value'), 3)} updates/s on {cb.get('cores')} threads ({fmt(t.get('speedup'), 0)}x) |")
        t = b.get("sac_c3")
        if isinstance(t, dict) and "value" in t:
            rf, cb = t.get("roofline", {}), t.get("cpu_baseline", {})
            add(f"| leg `sac_c3`: SAC-Lag, BASELINE configs[3] (1 M-row store in HBM, batch 1024, n_step 2) | {fmt(t['ms_per_update'] * 1e3, 1)} us per update = "
                f"{fmt(t['value'], 0)} updates/s, {fmt(rf.get('frac'), 3)} of the fp32 MFMA peak; CPU port {fmt(cb.get('value'), 1)} updates/s on {cb.get('cores')} threads |")
        lf = (r.get("latency_floor_parts_us") or {}) if isinstance(r, dict) else {}
        if "three_empty_launches_measured_in_this_run" in lf:
            e = lf.get("each_grid_behind_itself", {})
            add(f"| launch floors measured in this run (`fsrl_launch_floors`) | three empty launches of the step's grids behind each other "
                f"{fmt(lf['three_empty_launches_measured_in_this_run'], 2)} us (each grid behind itself: {fmt(e.get('fwdbwd'), 2)} / {fmt(e.get('wgrad'), 2)} / {fmt(e.get('adam'), 2)} us) |")
        if isinstance(b.get("no_clip"), dict) and "value" in b["no_clip"]:
            add(f"| same update without the gradient-norm clip (agent default; 2 launches per step) | {fmt(b['no_clip']['value'])} updates/s |")
        for key in ("grouped", "grouped_k8"):
            g = b.get(key)
            if isinstance(g, dict) and "aggregate_updates_per_s" in g:
                add(f"| grouped launches, {g['agents_per_gpu']} agents on one GPU | {fmt(g['aggregate_updates_per_s'])} updates/s aggregate "
                    f"({fmt(g['us_per_agent_step'], 1)} us per agent-step) |")
        if isinstance(b.get("multi_seed"), dict) and "value" in b["multi_seed"]:
            add(f"| {b['multi_seed']['seeds_per_gpu']} agents, one host thread + context each | {fmt(b['multi_seed']['value'])} updates/s aggregate |")
        for key, label in (("end_to_end", "end to end, in-process zero-cost env, 20 envs, device actor"),
                           ("end_to_end_host_actor", "the same with the host torch actor")):
            e = b.get(key)
            if isinstance(e, dict) and "env_steps_per_s" in e:
                ar = e.get("actor_resident")
                res = (f"; resident actor: {ar['per_collect']:g} kernel launch(es) per collect, {ar['calls_served']} calls through the doorbell"
                       if isinstance(ar, dict) else "")
                add(f"| {label} | {fmt(e['env_steps_per_s'], 0)} env-steps/s ({fmt(e['update_ms_per_collect'], 2)} ms update per collect{res}) |")
                ac = e.get("actor_call_us")
                if isinstance(ac, dict) and "resident" in ac:
                    add(f"| one device-actor call of that vector env (`fsrl_actor_sample`, {e.get('envs', 20)} rows, through ctypes) | {fmt(ac['resident'], 1)} us resident, {fmt(ac['launched'], 1)} us with one kernel launch per call |")
        for e in b.get("end_to_end_shmem", []) or []:
            if isinstance(e, dict) and "env_steps_per_s" in e:
                bound = e.get("env_bound_env_steps_per_s")
                procs = e.get("worker_processes", e["workers"])
                add(f"| worker-process env: {e['workers']} workers{'' if procs == e['workers'] else f' (capped: {procs} processes)'} x {e['busy_us']:g} us per env step, 32 envs ({e.get('collector_loop', '')}) | "
                    f"{fmt(e['env_steps_per_s'], 0)} env-steps/s" + (f" = {fmt(e['frac_of_env_bound'], 2)} of the env bound {fmt(bound, 0)}" if bound else "") + " |")
        add("")
    ks = kernels(f"{tag}_kernel_stats.csv")
    if ks:
        add(f"Kernel trace of the headline run (`profiles/{tag}_kernel_stats.csv`, rocprofv3 --kernel-trace --stats):\n")
        add("| kernel | calls | avg us | % |\n|---|---|---|---|")
        for n, c, us, pc in ks[:5]:
            add(f"| `{n[:70]}` | {c} | {fmt(us, 2)} | {fmt(pc, 1)} |")
        add("")
    tr = jl(f"{tag}_bench_trust.json")
    upd = {}
    f = os.path.join(PR, f"{tag}_pmc_traffic_updates.json")
    if os.path.exists(f):
        upd = json.load(open(f))
    if tr or jl(f"{tag}_bench_sac.json"):
        add(f"### Other update paths (`profiles/{tag}_bench_trust.json`, `_bench_sac.json`, `_bench_cvpo.json`, `_pmc_traffic_updates.json`)\n")
        add("| update | time | fp32-MFMA fraction (whole update) | HBM-side bytes per update (PMC) | CPU port (4 threads) |\n|---|---|---|---|---|")
        for t in tr:
            kind = t.get("bench")
            rf = t.get("roofline", {})
            tb = (upd.get(kind) or {}).get("hbm_bytes_per_update")
            cpu = t.get("cpu_oracle_ms_per_update_4thr")
            add(f"| {kind.upper()} ({'obs 60, ' if kind == 'cpo' else ''}256x256, N = 20 000) | {fmt(t['hip_ms_per_update'], 1)} ms | {fmt(rf.get('frac'), 3)} | "
                f"{(fmt(tb / 1e9, 2) + ' GB') if tb else 'n/a'} | {(fmt(cpu / 1e3, 1) + ' s = ' + fmt(t.get('speedup'), 0) + 'x') if cpu else 'n/a'} |")
        ck = (upd.get("cpo_splitk") or {}).get("hbm_bytes_per_update")
        if ck:
            add(f"| CPO with round 5's split-K weight-gradient kernel (`fsrl_tr_set_plan(wgrad = 2)`: the default up to r5) | see the plan table below | | {fmt(ck / 1e9, 2)} GB | |")
        cs = (upd.get("cpo_stream") or {}).get("hbm_bytes_per_update")
        if cs:
            add(f"| CPO with the one-pass streaming weight-gradient kernel (`fsrl_tr_set_plan(wgrad = 3)`, not the default) | same time within 1-2 % "
                f"(profiles/{tag}_wgrad2_ab_*.csv) | | {fmt(cs / 1e9, 2)} GB | |")
        for name, kind in ((f"{tag}_bench_sac.json", "sac"), (f"{tag}_bench_cvpo.json", "cvpo")):
            for t in jl(name)[-1:]:
                rf = t.get("roofline", {})
                us = t.get("ms_per_update", 0) * 1e3 if "ms_per_update" in t else t.get("us_per_update")
                tb = (upd.get(kind) or {}).get("hbm_bytes_per_update")
                cb = t.get("cpu_baseline", {})
                add(f"| {t['config']['workload'][:60]} | {fmt(us, 1)} us = {fmt(t['value'], 0)} updates/s | {fmt(rf.get('frac'), 3)} | "
                    f"{(fmt(tb / 1e6, 2) + ' MB') if tb else 'n/a'} | {(fmt(cb['value'], 1) + ' updates/s') if 'value' in cb else 'n/a'} |")
        add("")
    kt = kernels(f"{tag}_trust_kernel_stats.csv", top=8)
    if kt:
        add(f"Kernel trace of `tools/bench_trust.py` (CPO + TRPO-Lag + FOCOPS; `profiles/{tag}_trust_kernel_stats.csv`):\n")
        add("| kernel | calls | avg us | % |\n|---|---|---|---|")
        for n, c, us, pc in kt[:7]:
            add(f"| `{n[:70]}` | {c} | {fmt(us, 1)} | {fmt(pc, 1)} |")
        add("")
    f = os.path.join(PR, f"{tag}_pmc_mfma.json")
    if os.path.exists(f):
        mf = json.load(open(f))
        rows = sorted(((k, v) for k, v in mf.items() if v.get("mfma_util_pct") and v.get("mfma_f32_flops_per_launch", 0) > 0),
                      key=lambda kv: -kv[1]["mfma_util_pct"])
        if rows:
            add(f"MFMA utilisation (`SQ_VALU_MFMA_BUSY_CYCLES` / (GPU-active cycles x 1024 SIMDs), own PMC passes; `profiles/{tag}_pmc_mfma.json`):\n")
            add("| kernel | MFMA-busy % | fp32 MFMA FLOPs per launch (hardware count) |\n|---|---|---|")
            for k, v in rows[:10]:
                add(f"| `{k.replace('void ', '')[:70]}` | {fmt(v['mfma_util_pct'], 1)} | {fmt(v['mfma_f32_flops_per_launch'] / 1e6, 1)} M |")
            add("")
    g = jl(f"{tag}_bench_group.json")
    if g:
        add(f"Grouped launches (`profiles/{tag}_bench_group.json`): " +
            ", ".join(f"k = {x['agents_per_gpu']}: {fmt(x['aggregate_updates_per_s'])}" for x in g) + " updates/s aggregate" +
            (f" (every timed update restored from the {g[0]['reset']})" if "reset" in g[0] else "") + ".")
        for suffix, what in (("hostreset", "same box, the weights uploaded from the host inside the timed region (the harness up to round 4)"),
                             ("noclip", "same box, max_grad_norm off (the agent's default: 2 launches per step)")):
            h = jl(f"{tag}_bench_group_{suffix}.json")
            if h:
                add(f"  {what} (`profiles/{tag}_bench_group_{suffix}.json`): " +
                    ", ".join(f"k = {x['agents_per_gpu']}: {fmt(x['aggregate_updates_per_s'])}" for x in h) + ".")
        add("")
    # ---- r5: the full-batch kernel plans side by side (same box, alternated), the co-resident kernels' counters, the GEMM loops alone
    ab = [x for x in jl(f"{tag}_ab_trust_plans.json") if "alg" in x]
    f = os.path.join(PR, f"{tag}_ab_trust_plans.json")
    if ab:
        med = {}
        for x in ab:
            med.setdefault((x["alg"], x["plan"]), []).append(x["ms"])
        add(f"Full-batch kernel plans on one box, alternated (`tools/ab_trust_co.py`, `profiles/{tag}_ab_trust_plans.json`; ms per update, median):\n")
        plans = []
        for (alg, plan) in med:
            if plan not in plans:
                plans.append(plan)
        add("| plan (fsrl_tr_set_plan tile_rows, hvp) | CPO configs[2] | TRPO-Lag |\n|---|---|---|")
        for plan in plans:
            c_ = sorted(med.get(("cpo", plan), [float("nan")])); t_ = sorted(med.get(("trpo", plan), [float("nan")]))
            add(f"| {plan} | {fmt(c_[len(c_) // 2], 2)} | {fmt(t_[len(t_) // 2], 2)} |")
        add("")
    abw = [x for x in jl(f"{tag}_ab_wgrad_plans.json") if "alg" in x]
    if abw:
        med = {}
        for x in abw:
            med.setdefault((x["alg"], x["plan"]), []).append(x["ms"])
        add(f"Weight-gradient kernels on one box, alternated, default tile plan (`tools/ab_trust_co.py --wgrad`, `profiles/{tag}_ab_wgrad_plans.json`; ms per update, median):\n")
        plans = []
        for (alg, plan) in med:
            if plan not in plans:
                plans.append(plan)
        add("| weight-gradient plan (fsrl_tr_set_plan wgrad) | CPO configs[2] (obs 60) | TRPO-Lag (obs 8) |\n|---|---|---|")
        for plan in plans:
            c_ = sorted(med.get(("cpo", plan), [float("nan")])); t_ = sorted(med.get(("trpo", plan), [float("nan")]))
            add(f"| {plan} | {fmt(c_[len(c_) // 2], 2)} | {fmt(t_[len(t_) // 2], 2)} |")
        add("")
    f = os.path.join(PR, f"{tag}_pmc_trust_plans.json")
    if os.path.exists(f):
        pj = json.load(open(f))
        names = [n for n in pj if any(k in n for k in ("fb_hvp_co", "fb_hvp_mixed_kernel<256, true", "fb_tile_co", "fb_tile_mixed", "fb_wgrad_kernel", "fb_wgrad3_kernel"))]
        if names:
            add(f"Counter means per launch of the full-batch kernels (own PMC passes over `tools/ab_trust_co.py --only cpo`; `profiles/{tag}_pmc_trust_plans.json`; "
                "cycles per SIMD = SQ_VALU_MFMA_BUSY_CYCLES / 1024, GPU cycles = GRBM_GUI_ACTIVE / 8 XCDs, LDS cycles per CU = SQ_LDS_IDX_ACTIVE / 256):\n")
            add("| kernel | launches | GPU cycles | MFMA-busy cycles per SIMD | busy % | LDS-active cycles per CU | of them bank conflicts | wave residency (cycles, mean) |\n|---|---|---|---|---|---|---|---|")
            for n in sorted(names):
                v = pj[n]
                g = lambda k: (v.get(k) or {}).get("mean")     # noqa: E731
                gui = (g("GRBM_GUI_ACTIVE") or 0) / 8; mf = (g("SQ_VALU_MFMA_BUSY_CYCLES") or 0) / 1024
                lds = (g("SQ_LDS_IDX_ACTIVE") or 0) / 256; bc = (g("SQ_LDS_BANK_CONFLICT") or 0) / 256
                wc = g("SQ_WAVE_CYCLES"); wv = g("SQ_WAVES")
                add(f"| `{n[:60]}` | {int((v.get('GRBM_GUI_ACTIVE') or {}).get('launches', 0))} | {fmt(gui, 0)} | {fmt(mf, 0)} | "
                    f"{fmt(100 * mf / gui if gui else None, 1)} | {fmt(lds, 0)} | {fmt(bc, 0)} | {fmt(4 * wc / wv if wc and wv else None, 0)} |")
            add("")

    def newest(name):
        """this tag's file, or the newest earlier round's (a fast capture does not repeat what did not change)"""
        n0 = int(tag[1:])
        for n in range(n0, 0, -1):
            cand = f"r{n:02d}_{name}"
            if os.path.exists(os.path.join(PR, cand)):
                return cand
        return None
    for name, title in (("ubench_mfma_pat.txt", "The GEMM inner loops in isolation: 16x16x4 dependent chains vs 32x32x2 (`tools/ubench/mfma_pat.hip`)"),
                        ("ubench_gridsync.txt", "Device-wide barrier inside a kernel (`tools/ubench/gridsync.hip`)"),
                        ("ubench_dispatch.txt", "Workgroup dispatch rate (`tools/ubench/dispatch.hip`)"),
                        ("ubench_chain.txt", "Three launches per step vs ONE launch whose blocks wait for lower-numbered blocks (`tools/ubench/chain.hip`)")):
        cand = newest(name)
        if cand:
            add(f"{title}, `profiles/{cand}`:\n\n```\n" + open(os.path.join(PR, cand)).read().strip() + "\n```\n")
    return "\n".join(L)


def splice(path, tag, text):
    s = open(path).read()
    a, z = f"<!-- BEGIN GENERATED {tag} -->", f"<!-- END GENERATED {tag} -->"
    if a not in s:
        return False
    s = re.sub(re.escape(a) + r".*?" + re.escape(z), lambda m: a + "\n" + text + "\n" + z, s, flags=re.S)
    open(path, "w").write(s)
    return True


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    text = build(tag)
    open(os.path.join(PR, f"RESULTS_{tag}.md"), "w").write(f"# Measured numbers, {tag} (generated by tools/gen_results_md.py from profiles/{tag}_*)\n\n" + text + "\n")
    for doc in ("README.md", "DESIGN.md"):
        print(doc, "updated" if splice(os.path.join(ROOT, doc), tag, text) else "has no generated block for " + tag)
