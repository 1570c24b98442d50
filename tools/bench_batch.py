#!/usr/bin/env python
"""PPO-Lag update time against the minibatch size (configs[1] shape, N = 20 000, 4 passes): one line per batch size.
    python tools/bench_batch.py [64 128 ...]          FSRL_TALL=0 : 16-row tiles only (fsrl_ppo_set_plan; default automatic)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fsrl_amd.engine import Engine, EngineConfig  # noqa: E402

if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024, 2048]
    eng = Engine(EngineConfig(obs_dim=bench.OBS, act_dim=bench.ACT, hidden=bench.HID, env_num=bench.ENVS, buffer_size=100000,
                              max_grad_norm=0.5, target_kl=None))
    theta = bench.orthogonal_theta(0, eng.n_params)
    obs, act, rew, cost, term, trunc = bench.make_inputs(0)
    ids = np.arange(bench.ENVS)
    for t in range(bench.NROWS // bench.ENVS):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    eng.sync()
    if os.environ.get("FSRL_TALL"):
        eng.ppo_set_plan(int(os.environ["FSRL_TALL"]))
    lag, resc = np.array([0.75]), 1 / 1.75
    for B in sizes:
        for k in range(2):
            eng.set_params(theta); eng.optim_reset(); st, _ = eng.ppo_update(lag, resc, B, 4, perms=None, seed=k + 1)
        eng.sync(); t0 = time.perf_counter(); n = 5
        for k in range(n):
            eng.set_params(theta); eng.optim_reset(); st, _ = eng.ppo_update(lag, resc, B, 4, perms=None, seed=10 + k)
        eng.sync(); dt = (time.perf_counter() - t0) / n
        print(f"batch {B}: {dt * 1e3:.2f} ms per update, {st.shape[0]} steps, {dt * 1e6 / st.shape[0]:.1f} us per step, "
              f"{bench.NROWS * 4 / dt / 1e6:.2f} M rows/s", flush=True)
