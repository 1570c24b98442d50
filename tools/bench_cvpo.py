#!/usr/bin/env python
"""Secondary benchmark: CVPO.update at the reference agent's defaults (cvpo_agent.py:81-115: 128x128 networks,
SingleCritic pair, batch 256, K = 16 particles, n_step 2, gamma 0.98) on a SafetyCarCircle-sized problem (obs 40,
act 2 by default), replay store resident in HBM.  One JSON line; the CPU figure is the oracle (torch fp32, 4 threads)
on the same store."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fsrl_amd import _lib  # noqa: E402
from fsrl_amd.engine import Engine, EngineConfig  # noqa: E402
from oracle.cvpo import CVPOConfig, CVPOOracle  # noqa: E402
from oracle.sac_lag import ReplayIndex  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=200_000)
    ap.add_argument("--envs", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--obs", type=int, default=40)
    ap.add_argument("--act", type=int, default=2)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--updates", type=int, default=1000)
    ap.add_argument("--cpu-updates", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    Do, Da, H, E, B, K = a.obs, a.act, a.hidden, a.envs, a.batch, a.k
    T = a.rows // E
    rng = np.random.default_rng(0)
    ocfg = CVPOConfig(obs_dim=Do, act_dim=Da, hidden=(H, H), sample_act_num=K, max_episode_steps=300)
    eng = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=Do, act_dim=Da, hidden=H, n_critics=2, env_num=E,
                              buffer_size=a.rows, gamma=ocfg.gamma, target_kl=None))
    eng.cvpo_init(ocfg.qc_thres, sample_act_num=K)
    if os.environ.get("FSRL_SAC_SPLITK"):      # A/B: split-K weight gradients at every batch size (fsrl_sac_set_plan)
        eng.sac_set_plan(1)
    o = CVPOOracle(ocfg)
    torch.manual_seed(0)

    def orth(spec):
        parts = []
        for name, shape in spec.items():
            if name.startswith("W"):
                w = torch.empty(shape); torch.nn.init.orthogonal_(w); parts.append(w.reshape(-1))
            else:
                parts.append(torch.zeros(shape).reshape(-1))
        return torch.cat(parts).numpy()
    th_a = orth(o.aspec); th_c = np.concatenate([orth(o.cspec), orth(o.cspec)])
    eng.sac_set_params(th_a, th_c, 0.0)
    obs = rng.standard_normal((T + 1, E, Do)).astype(np.float32)
    act = np.clip(rng.standard_normal((T, E, Da)), -1, 1).astype(np.float32)
    rew = rng.normal(0.5, 0.5, (T, E)); cost = (rng.random((T, E)) < 0.1).astype(np.float64)
    trunc = np.zeros((T, E), bool); trunc[299::300] = True
    term = np.zeros((T, E), bool)
    ids = np.arange(E)
    for t in range(T):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    eng.sync()
    eng.cvpo_pre_update()
    for _ in range(20):
        eng.cvpo_update(B, seed=0)
    eng.sync()
    t0 = time.perf_counter()
    for u in range(a.updates):
        eng.cvpo_update(B, sync=False)
        if u % 50 == 49:                      # a collect cycle every 50 updates (update_per_step 0.2 x 250 steps)
            eng.cvpo_post_update(); eng.cvpo_pre_update()
    st = eng.sac_drain()
    dev = (time.perf_counter() - t0) / a.updates
    assert np.isfinite(st).all() and len(st) == min(a.updates, 4096), st[-1]
    out = {"metric": "cvpo policy-updates/sec", "value": 1.0 / dev, "unit": "updates/s", "us_per_update": dev * 1e6,
           "config": {"workload": f"CVPO defaults: obs {Do} act {Da} {H}x{H}, SingleCritic pair, store {T * E} rows in HBM, "
                                  f"batch {B}, K {K}, n_step 2", "updates": a.updates}, "dtype": "fp32",
           "last_stats": [float(x) for x in st[-1]]}
    # roofline of the whole update (13 launches), algorithmic FLOPs: forward F per row and network, gradient = 3 F
    Fq = 2 * ((Do + Da) * H + H * H + H); Fa = 2 * (Do * H + H * H + H * 2 * Da)
    flops = B * (Fa + 2 * Fq            # target action at s_{t+n} and the two target critics
                 + 2 * 3 * Fq           # critic step (SingleCritic pair)
                 + Fa + K * 2 * Fq      # actor_old at s_t and the K particles through both updated critics
                 + 3 * Fa)              # M-step: forward + backward of the actor
    out["roofline"] = {"bound": "mfma", "scope": "whole update (13 launches)", "achieved": flops / dev / 1e12, "peak": 157.3,
                       "unit": "TFLOP/s", "frac": flops / dev / 1e12 / 157.3, "flops_per_update": flops, "traffic": None}
    if not a.no_cpu:
        torch.set_num_threads(4)
        o.set_params(th_a, th_c)
        slot = lambda x: np.ascontiguousarray(np.swapaxes(x, 0, 1)).reshape((E * T, ) + x.shape[2:])  # noqa: E731
        store = {"obs": slot(obs[:-1]), "obs_next": slot(obs[1:]), "act": slot(act), "rew": slot(rew),
                 "cost": slot(cost), "terminated": slot(term)}
        index = ReplayIndex([T] * E, T, slot(term | trunc))
        r2 = np.random.default_rng(1)
        t0 = time.perf_counter()
        for _ in range(a.cpu_updates):
            idx = r2.integers(0, E * T, B)
            o.update(store, index, idx, r2.standard_normal((B, Da)).astype(np.float32),
                     r2.standard_normal((K, B, Da)).astype(np.float32))
        cpu = (time.perf_counter() - t0) / a.cpu_updates
        out["cpu_baseline"] = {"value": 1.0 / cpu, "unit": "updates/s", "cores": 4, "kind": "port",
                               "sample": f"{a.cpu_updates} updates of the same store/batch (oracle, torch fp32)"}
        out["speedup_vs_cpu"] = cpu / dev
    print(json.dumps(out))


if __name__ == "__main__":
    main()
