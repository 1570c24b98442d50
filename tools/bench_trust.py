#!/usr/bin/env python
"""Secondary benchmarks: BASELINE.json configs[2] (CPO, SafetyPointGoal shape: obs 60, act 2,
256x256, N = 20 000 full batch, CG 10) and TRPO-Lagrangian on the configs[1] shape.  One JSON
line each; the CPU figure is the oracle (torch fp32, 4 threads) on the same inputs."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fsrl_amd.engine import Engine, EngineConfig  # noqa: E402
from oracle.ppo_lag import OnPolicyData  # noqa: E402
from oracle.trust_region import CPOConfig, CPOOracle, TRPOConfig, TRPOLagOracle  # noqa: E402


def pmc_update_traffic(alg):
    """HBM-side bytes per update from the committed PMC passes (profiles/rNN_pmc_traffic_updates.json: FETCH_SIZE x 2 +
    WRITE_SIZE over a whole traced run / its updates); counters cannot be read inside this process.  None without a capture."""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for tag in ("r06", "r05", "r04", "r03"):
        f = os.path.join(here, "profiles", f"{tag}_pmc_traffic_updates.json")
        if os.path.exists(f):
            j = json.load(open(f))
            d = j.get(alg)
            if d:
                meta = j.get("_meta", {})
                src = f"profiles/{tag}_pmc_traffic_updates.json"
                if meta:          # which sources the capture ran (tools/collect_profiles.py): a stale file says so
                    src += f" (captured at {meta.get('head', '?')}, csrc {meta.get('csrc_sha16', '?')})"
                return float(d["hbm_bytes_per_update"]), src
    return None, None


def inputs(rng, envs, T, obs_dim, act_dim, ep):
    obs = rng.standard_normal((T + 1, envs, obs_dim)).astype(np.float32)
    act = (0.3 * rng.standard_normal((T, envs, act_dim))).astype(np.float32)
    rew = rng.normal(0.5, 0.5, (T, envs)); cost = (rng.random((T, envs)) < 0.1).astype(np.float64)
    trunc = np.zeros((T, envs), bool); trunc[ep - 1::ep] = True
    return obs, act, rew, cost, np.zeros((T, envs), bool), trunc


def orth_theta(o, seed):
    torch.manual_seed(seed)
    parts = []
    for spec in o.specs:
        for name, shape in spec.items():
            if name == "sigma_param":
                parts.append(torch.full(shape, -0.5).reshape(-1))
            elif name.startswith("W"):
                w = torch.empty(shape); torch.nn.init.orthogonal_(w); parts.append(w.reshape(-1))
            else:
                parts.append(torch.zeros(shape).reshape(-1))
    return torch.cat(parts).numpy()


def run(kind, obs_dim, act_dim, hid, envs=20, T=1000, ep=1000, repeat=4, cpu_repeat=1, timed=5, emit=True, no_cpu=None):
    """emit: print the JSON line (command line use); bench.py's legs call with emit=False and take the returned dict."""
    no_cpu = bool(os.environ.get("FSRL_NO_CPU")) if no_cpu is None else no_cpu
    rng = np.random.default_rng(0)
    obs, act, rew, cost, term, trunc = inputs(rng, envs, T, obs_dim, act_dim, ep)
    hs = (hid, hid)
    lay = os.environ.get("FSRL_TR_LAYERED")                # "256x256x256": the same workload on a layered context
    if lay:
        hs = tuple(int(x) for x in lay.split("x"))
    eng = Engine(EngineConfig(obs_dim=obs_dim, act_dim=act_dim, hidden_sizes=hs, force_layered=bool(lay), env_num=envs, target_kl=None,
                              lr=1e-3 if kind == "cpo" else 5e-4))
    if kind == "cpo":
        ocfg = CPOConfig(obs_dim=obs_dim, act_dim=act_dim, hidden=hs, optim_critic_iters=10,
                         max_backtracks=10, cost_limit=10.0)
        o = CPOOracle(ocfg)
    else:
        ocfg = TRPOConfig(obs_dim=obs_dim, act_dim=act_dim, hidden=hs, optim_critic_iters=20)
        o = TRPOLagOracle(ocfg)
    theta = orth_theta(o, 0)
    ids = np.arange(envs)
    for t in range(T):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    eng.sync()

    plan = os.environ.get("FSRL_TR_PLAN")                  # "tile_rows,hvp" (fsrl_tr_set_plan): A/B of the kernel plans
    if plan:
        eng.tr_set_plan(*[int(x) for x in plan.split(",")])

    split = os.environ.get("FSRL_TR_SPLIT")                # "n32_tile,n32_hvp" (fsrl_tr_set_tile_split): A/B of the co-resident launches' tile mix
    if split:
        eng.tr_set_tile_split(*[int(x) for x in split.split(",")])

    delay = os.environ.get("FSRL_TR_DELAY")                # "tile,hvp" (fsrl_tr_set_co_delay): start offset of the second resident workgroup
    if delay:
        eng.tr_set_co_delay(*[int(x) for x in delay.split(",")])

    def device_update():
        # every timed update is the same workload: initial weights AND a fresh optimiser state -- round 1 restored the
        # weights only, so the critics' Adam moments of the previous update leaked into the next one and the reported
        # first-repeat vf losses drifted from the (fresh) oracle's (profiles/r01_bench_trust.json: 55.9 vs 43.3)
        eng.set_params(theta); eng.optim_reset()
        if kind == "cpo":
            eng.tr_begin(target_kl=0.01, l2_reg=0.001, critic_lr=1e-3, max_backtracks=10, optim_critic_iters=10,
                         cost_limit=10.0)
            return eng.cpo_learn(25.0, repeat)
        eng.tr_begin(target_kl=0.001, critic_lr=5e-4, max_backtracks=10, optim_critic_iters=20)
        return eng.trpo_learn([0.75], 1 / 1.75, repeat)

    device_update()
    times = []
    for _ in range(timed):      # the median of five: the box's CPU quota throttles a process for tens of ms now and then
        t0 = time.perf_counter()
        stats = device_update()
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    # ---- roofline of the whole update: algorithmic FLOPs (forward F = 2(Do H + H H + H out) per row and
    #      network, gradient = 3 F, Hessian-vector product = 3 F_actor) over the measured time, against the fp32 MFMA peak
    N = envs * T
    Fa = 2 * (obs_dim * hid + hid * hid + hid * act_dim); Fc = 2 * (obs_dim * hid + hid * hid + hid)
    evals = float(np.sum(eng.tr_linesearch_evals()))                       # line-search forwards of the last update
    if kind == "cpo":       # per repeat: 10 critic steps x 2 critics, 2 gradients, 2 CG solves x (10 + 1) HVPs
        flops = repeat * (10 * 2 * 3 * Fc + (2 + 22) * 3 * Fa) * N + evals * Fa * N
    else:                   # per repeat: 1 gradient, 10 + 1 HVPs, 20 critic steps x 2 critics
        flops = repeat * ((1 + 11) * 3 * Fa + 20 * 2 * 3 * Fc) * N + evals * Fa * N
    roof = {"bound": "mfma", "scope": "whole update (all launches, host line-search control included)",
            "achieved": flops / dt / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flops / dt / 1e12 / 157.3,
            "flops_per_update": flops, "traffic": None}
    tb, src = pmc_update_traffic(kind)
    if tb is not None:
        # algorithmic: every full-batch pass reads the 60 B/row inputs once + the side buffers it spills for the
        # weight-side kernel; the figure below is what the memory side actually moved per update
        roof.update(traffic=tb, traffic_source=src, traffic_gb_per_s=tb / dt / 1e9,
                    algorithmic_input_bytes_per_pass=60 * N)
    if no_cpu:
        res = {"bench": kind, "hip_ms_per_update": dt * 1e3, "roofline": roof}
        if emit:
            print(json.dumps(res))
        eng.close()
        return res
    em = lambda a: np.concatenate([a[:, e] for e in range(envs)])
    data = OnPolicyData(obs=em(obs[:-1]), act=em(act), rew=em(rew), cost=em(cost), terminated=em(term),
                        truncated=em(trunc), obs_next=em(obs[1:]), end_flag=em(term | trunc))
    torch.set_num_threads(4)
    o.set_params(theta)
    t0 = time.perf_counter()
    if kind == "cpo":
        _, rows = o.update(data, 25.0, cpu_repeat)
        ostat = rows[0][0]
    else:
        _, rows = o.update(data, [0.75], 1 / 1.75, cpu_repeat)
        ostat = rows[0][0]
    cdt = (time.perf_counter() - t0) / cpu_repeat * repeat
    res = {"bench": kind, "obs": obs_dim, "act": act_dim, "hidden": hid, "N": envs * T,
           "repeat": repeat, "hip_ms_per_update": dt * 1e3, "hip_updates_per_s": 1 / dt,
           "cpu_oracle_ms_per_update_4thr": cdt * 1e3, "speedup": cdt / dt,
           "roofline": roof, "cpu_baseline": {"value": 1.0 / cdt, "unit": "updates/s", "cores": 4, "kind": "port",
                                              "sample": f"{cpu_repeat} of {repeat} repeats of the same update (oracle, torch fp32)"},
           "hip_first_repeat": [float(x) for x in stats[0]],
           "oracle_first_repeat": {k: float(v) for k, v in ostat.items()}}
    if emit:
        print(json.dumps(res))
    eng.close()
    return res


def run_focops(obs_dim=8, act_dim=2, hid=256, envs=20, T=1000, ep=250, batch=256, repeat=4):
    """FOCOPS on the configs[1] shape (N = 20 000, batch 256, 4 passes = 312 minibatch steps)."""
    from fsrl_amd import _lib
    from oracle.focops import FOCOPSConfig, FOCOPSOracle
    rng = np.random.default_rng(0)
    obs, act, rew, cost, term, trunc = inputs(rng, envs, T, obs_dim, act_dim, ep)
    eng = Engine(EngineConfig(algo=_lib.ALGO_FOCOPS, obs_dim=obs_dim, act_dim=act_dim, hidden=hid, env_num=envs, target_kl=None))
    eng.focops_init(delta=1e9)                      # KL early stop off: every update runs all 312 steps
    if os.environ.get("FSRL_FOC_PLAN"):               # A/B: fsrl_focops_set_plan (1 four launches, 2 the narrow step kernel)
        eng.focops_set_plan(int(os.environ["FSRL_FOC_PLAN"]))
    o = FOCOPSOracle(FOCOPSConfig(obs_dim=obs_dim, act_dim=act_dim, hidden=(hid, hid), delta=1e9))
    theta = orth_theta(o, 0)
    ids = np.arange(envs)
    for t in range(T):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    eng.sync()

    def device_update(k):
        eng.set_params(theta); eng.optim_reset()
        return eng.focops_update(0.1, -15.0, batch, repeat, perms=None, seed=k + 1)

    device_update(0)
    times = []
    for k in range(5):          # the median of five: the box's CPU quota throttles a process for tens of ms now and then
        t0 = time.perf_counter()
        stats, _ = device_update(k + 1)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    Fa = 2 * (obs_dim * hid + hid * hid + hid * act_dim); Fc = 2 * (obs_dim * hid + hid * hid + hid)
    flops = repeat * envs * T * 3 * (Fa + 2 * Fc) + envs * T * (4 * Fc + Fa)          # learn + process_fn, like PPO-Lag
    out = {"bench": "focops", "obs": obs_dim, "act": act_dim, "hidden": hid, "N": envs * T, "batch": batch, "repeat": repeat,
           "steps": int(stats.shape[0]), "hip_ms_per_update": dt * 1e3, "hip_us_per_step": dt * 1e6 / stats.shape[0],
           "roofline": {"bound": "mfma", "scope": "whole update", "achieved": flops / dt / 1e12, "peak": 157.3,
                        "unit": "TFLOP/s", "frac": flops / dt / 1e12 / 157.3, "flops_per_update": flops, "traffic": None}}
    if not os.environ.get("FSRL_NO_CPU"):
        em = lambda a: np.concatenate([a[:, e] for e in range(envs)])  # noqa: E731
        data = OnPolicyData(obs=em(obs[:-1]), act=em(act), rew=em(rew), cost=em(cost), terminated=em(term),
                            truncated=em(trunc), obs_next=em(obs[1:]), end_flag=em(term | trunc))
        torch.set_num_threads(4)
        o.set_params(theta, nu=0.1)
        r2 = np.random.default_rng(1)
        t0 = time.perf_counter()
        o.update(data, 25.0, batch, 1, [r2.permutation(envs * T)])
        cdt = (time.perf_counter() - t0) * repeat
        out.update(cpu_oracle_ms_per_update_4thr=cdt * 1e3, speedup=cdt / dt)
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    only = os.environ.get("FSRL_ONLY", "")          # FSRL_ONLY=cpo|trpo|focops: one of the three (per-algorithm kernel traces)
    if only in ("", "cpo"):
        run("cpo", 60, 2, 256)
    if only in ("", "trpo"):
        run("trpo", 8, 2, 256, ep=250)
    if only in ("", "focops"):
        run_focops()
