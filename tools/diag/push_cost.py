#!/usr/bin/env python
"""What fsrl_store_push costs per call (host bookkeeping + the pinned window's flushes), by row count."""
import json
import time

import numpy as np

from fsrl_amd.engine import Engine, EngineConfig

out = {}
for k in (8, 20, 32):
    eng = Engine(EngineConfig(obs_dim=8, act_dim=2, hidden=256, env_num=k, buffer_size=k * 2000, max_grad_norm=0.5, target_kl=None))
    rng = np.random.default_rng(0)
    ids = np.arange(k)
    obs = rng.standard_normal((k, 8)).astype(np.float32); act = rng.standard_normal((k, 2)).astype(np.float32)
    rew = rng.standard_normal(k); cost = np.zeros(k); term = np.zeros(k, bool); trunc = np.zeros(k, bool)
    for _ in range(200):
        eng.push(ids, obs, act, rew, cost, term, trunc, obs)
    n = 1500
    t0 = time.perf_counter()
    for _ in range(n):
        eng.push(ids, obs, act, rew, cost, term, trunc, obs)
    out[f"k{k}_us_per_push"] = round((time.perf_counter() - t0) / n * 1e6, 2)
    eng.close()
print(json.dumps(out))
