#!/usr/bin/env python
"""Secondary benchmark: BASELINE.json configs[3] -- SAC-Lagrangian on the SafetyAntRun shape
(obs 33, act 8, 256x256 actor + 2 double critics), 1 M-row replay store resident in HBM, batch 1024,
n_step 2, auto alpha.  One JSON line; the CPU figure is the oracle (torch fp32, 4 threads) on the
same store.  `--rows` shrinks the store for a quick run."""
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
from oracle.sac_lag import ReplayIndex, SACConfig, SACLagOracle  # noqa: E402


def main(argv=None, emit=True):
    """emit: print the JSON line (command line use); bench.py's leg calls main([...], emit=False) and takes the dict."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--envs", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--updates", type=int, default=1000)
    ap.add_argument("--cpu-updates", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args(argv)
    Do, Da, H, E, B = 33, 8, a.hidden, a.envs, a.batch
    T = a.rows // E
    rng = np.random.default_rng(0)
    eng = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=Do, act_dim=Da, hidden=H, n_critics=2, env_num=E,
                              buffer_size=a.rows, gamma=0.99, target_kl=None))
    eng.sac_init()
    if os.environ.get("FSRL_SAC_SPLITK"):      # A/B: split-K weight gradients at every batch size (fsrl_sac_set_plan)
        eng.sac_set_plan(1)
    if os.environ.get("FSRL_SAC_PLAN"):        # A/B: plan bits (2 = sample / gather / n-step as launches of their own)
        eng.sac_set_plan(int(os.environ["FSRL_SAC_PLAN"]))
    cfg = SACConfig(obs_dim=Do, act_dim=Da, hidden=(H, H))
    o = SACLagOracle(cfg)
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
    # ---- fill the store (slot-major host copy kept for the oracle)
    t0 = time.perf_counter()
    obs = rng.standard_normal((T + 1, E, Do)).astype(np.float32)
    act = np.tanh(rng.standard_normal((T, E, Da))).astype(np.float32)
    rew = rng.normal(0.5, 0.5, (T, E)); cost = (rng.random((T, E)) < 0.1).astype(np.float64)
    trunc = np.zeros((T, E), bool); trunc[999::1000] = True
    term = np.zeros((T, E), bool)
    ids = np.arange(E)
    for t in range(T):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    eng.sync()
    fill_s = time.perf_counter() - t0
    lag, resc = [0.3], 1.0 / 1.3
    for _ in range(200):                                   # 200 updates of warm-up: the GPU's clocks have ramped after the
        eng.sac_update(B, lag, resc, seed=0, sync=False)   # host-bound store fill (a 20-update warm-up left some runs at 260 us)
    eng.sac_drain()
    eng.sync()
    blocks = []
    for _ in range(3):                                     # the median of three blocks of `updates` consecutive updates
        t0 = time.perf_counter()
        for _ in range(a.updates):
            eng.sac_update(B, lag, resc, sync=False)       # enqueue only; stats drained in bulk (facade default)
        st = eng.sac_drain()
        blocks.append((time.perf_counter() - t0) / a.updates)
        assert np.isfinite(st).all() and len(st) == min(a.updates, 4096), st[-1]
    dev = float(np.median(blocks))
    out = {"metric": "sac_lag policy-updates/sec", "value": 1.0 / dev, "unit": "updates/s",
           "ms_per_update": dev * 1e3, "samples_per_s": B / dev,
           "config": {"workload": f"SAC-Lag SafetyAntRun shape obs {Do} act {Da} {H}x{H}, store {T * E} rows in HBM, "
                                  f"batch {B}, n_step 2", "updates": a.updates, "blocks_us_per_update": [round(b * 1e6, 1) for b in blocks]},
           "store_fill_rows_per_s": T * E / fill_s, "dtype": "fp32",
           # r5 default: sample + gather inside the actors' forward launch; FSRL_SAC_PLAN bits 4 / 1 / 2 add launches back
           "launches_per_update": 9 + sum(1 for bit in (16, 2, 4) if int(os.environ.get("FSRL_SAC_PLAN", "0")) & bit)}
    # roofline of the whole update (12 launches): algorithmic FLOPs per sample (~4.9 MFLOP at 256x256)
    Fq = 2 * ((Do + Da) * H + H * H + H); Fa = 2 * (Do * H + H * H + H * 2 * Da)
    per_sample = (2 * Fa            # actor forwards at s_{t+n} and s_t
                  + 4 * Fq          # four target Q-nets
                  + 4 * 3 * Fq      # critic step: forward + backward of four Q-nets
                  + 4 * 2 * Fq      # Q(s, a_pi) forward + input gradient of four Q-nets
                  + 2 * Fa)         # actor backward
    flops = per_sample * B
    out["roofline"] = {"bound": "mfma", "scope": "whole update (9 launches)", "achieved": flops / dev / 1e12, "peak": 157.3,
                       "unit": "TFLOP/s", "frac": flops / dev / 1e12 / 157.3, "flops_per_update": flops,
                       "algorithmic_gather_bytes": B * ((2 * Do + Da) * 4 + 17), "traffic": None}
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_trust import pmc_update_traffic
    tb, src = pmc_update_traffic("sac")
    if tb is not None:        # HBM-side bytes per update (PMC, profiles/): the gathered rows are ~0.3 MB of it, the rest is parameters,
        out["roofline"].update(traffic=tb, traffic_source=src, traffic_gb_per_s=tb / dev / 1e9)   # Adam state and side buffers
    if not a.no_cpu:
        torch.set_num_threads(4)
        o.set_params(th_a, th_c, 0.0)
        sub = -(-a.rows // E)
        slot = lambda x: np.ascontiguousarray(np.swapaxes(x, 0, 1)).reshape((E * T, ) + x.shape[2:])  # noqa: E731
        store = {"obs": slot(obs[:-1]), "obs_next": slot(obs[1:]), "act": slot(act), "rew": slot(rew),
                 "cost": slot(cost), "terminated": slot(term)}
        assert sub == T
        index = ReplayIndex([T] * E, sub, slot(term | trunc))
        r2 = np.random.default_rng(1)
        t0 = time.perf_counter()
        for _ in range(a.cpu_updates):
            idx = r2.integers(0, E * T, B)
            o.update(store, index, idx, r2.standard_normal((B, Da)).astype(np.float32),
                     r2.standard_normal((B, Da)).astype(np.float32), lag, resc)
        cpu = (time.perf_counter() - t0) / a.cpu_updates
        out["cpu_baseline"] = {"value": 1.0 / cpu, "unit": "updates/s", "cores": 4, "kind": "port",
                               "sample": f"{a.cpu_updates} updates of the same store/batch (oracle, torch fp32)"}
        out["speedup_vs_cpu"] = cpu / dev
    eng.close()
    if emit:
        print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
