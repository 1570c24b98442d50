#!/usr/bin/env python
"""Aggregate policy-updates/s of k independent PPO-Lagrangian agents on ONE MI355X, stepped in lock step by the grouped
launches (fsrl_group_ppo_update): the BASELINE configs[1] workload per agent (obs 8 / act 2 / 256x256 / N = 20 000 /
batch 256 / 4 passes / grad-clip 0.5).  One JSON line per k.

    python tools/bench_group.py [--ks 1 2 4 8] [--updates 6] [--no-clip] [--host-reset] [--tall -1 0 8]

Every timed update starts from the same state (initial weights, fresh Adam moments): restored from the HBM snapshot, as bench.py
does for the single agent (--host-reset: round 4's harness, which uploaded the weights from the host inside the timed region)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import ACT, BATCH, ENVS, F32_MFMA_PEAK_TFLOPS, HID, NROWS, OBS, REPEAT, flops_fwdbwd_launch, make_inputs, orthogonal_theta  # noqa: E402
from fsrl_amd.engine import Engine, EngineConfig, EngineGroup  # noqa: E402


def run(k, updates, clip, host_reset=False, tall=-1):
    engs, thetas = [], []
    for i in range(k):
        e = Engine(EngineConfig(obs_dim=OBS, act_dim=ACT, hidden=HID, env_num=ENVS, buffer_size=100000, max_grad_norm=clip,
                                target_kl=None))
        th = orthogonal_theta(i, e.n_params)
        obs, act, rew, cost, term, trunc = make_inputs(i)
        ids = np.arange(ENVS)
        for t in range(NROWS // ENVS):
            e.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
        e.sync()
        engs.append(e); thetas.append(th)
    grp = EngineGroup(engs)
    grp.set_plan(tall)
    lags, resc = np.full((k, 1), 0.75), np.full(k, 1 / 1.75)
    for e, th in zip(engs, thetas):
        e.set_params(th); e.optim_reset(); e.state_snapshot()

    def one(u):
        for e, th in zip(engs, thetas):                 # same workload every update: initial weights, fresh Adam state
            if host_reset:
                e.set_params(th); e.optim_reset()
            else:
                e.state_restore()
        return grp.ppo_update(lags, resc, BATCH, REPEAT, seed=u + 1)[0]
    one(0)
    for e in engs:
        e.sync()
    times = []
    for u in range(updates):                            # an update ends with its statistics on the host: every update is timed on its own
        t0 = time.perf_counter()
        st = one(u + 1)
        for e in engs:
            e.sync()
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))                        # the median: the box's CPU quota throttles a process for tens of ms now and then
    dt_mean = float(np.mean(times))
    steps = st[0].shape[0]
    assert all(np.isfinite(s).all() for s in st)
    grp.close()
    for e in engs:
        e.close()
    return {"agents_per_gpu": k, "aggregate_updates_per_s": k / dt, "per_agent_updates_per_s": 1 / dt,
            "ms_per_group_update": dt * 1e3, "timing": "median of %d updates" % updates, "aggregate_updates_per_s_mean": k / dt_mean, "us_per_step_all_agents": dt * 1e6 / steps,
            "us_per_agent_step": dt * 1e6 / steps / k, "grad_clip": clip,
            "reset": "host upload" if host_reset else "HBM snapshot", "tall_tiles_plan": tall,
            "fwdbwd_flops_per_step_all_agents": flops_fwdbwd_launch(NROWS / (steps / REPEAT)) * k}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--updates", type=int, default=10)
    ap.add_argument("--no-clip", action="store_true")
    ap.add_argument("--host-reset", action="store_true")
    ap.add_argument("--tall", type=int, nargs="+", default=[-1],
                    help="fsrl_group_set_plan values to run (each k once per value): -1 automatic, 0 = 16-row tiles only, n = count")
    a = ap.parse_args()
    for k in a.ks:
        for tall in a.tall:
            print(json.dumps(run(k, a.updates, None if a.no_clip else 0.5, a.host_reset, tall)), flush=True)
