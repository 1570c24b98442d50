#!/usr/bin/env python
"""Is the step's slow mode (25.2 vs 24.05 us, one process in four to six) tied to the process or to the context?  Eight engines created,
measured and closed one after the other INSIDE one process: us per optimiser step of the headline update for each."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fsrl_amd.engine import Engine, EngineConfig  # noqa: E402

obs, act, rew, cost, term, trunc = bench.make_inputs(0)
ids = np.arange(bench.ENVS)
out = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    e = Engine(EngineConfig(obs_dim=bench.OBS, act_dim=bench.ACT, hidden=bench.HID, env_num=bench.ENVS, buffer_size=100000,
                            max_grad_norm=0.5, target_kl=None))
    for t in range(bench.NROWS // bench.ENVS):
        e.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    e.sync()
    th = bench.orthogonal_theta(0, e.n_params)
    e.set_params(th); e.optim_reset(); e.state_snapshot()
    ts = []
    for u in range(8):
        e.state_restore()
        t0 = time.perf_counter(); st, _ = e.ppo_update([0.75], 1 / 1.75, bench.BATCH, bench.REPEAT, seed=u + 1); e.sync()
        ts.append(time.perf_counter() - t0)
    out.append(float(np.median(ts[2:])) * 1e6 / st.shape[0])
    e.close()
print("us per step, engine by engine in one process:", " ".join(f"{x:.2f}" for x in out))
