#!/usr/bin/env python
"""PPO-Lag update time at other widths (the agents' default is 128x128; BASELINE configs use 256x256)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsrl_amd.engine import Engine, EngineConfig  # noqa: E402

for H in (64, 128, 256):
    eng = Engine(EngineConfig(obs_dim=8, act_dim=2, hidden=H, env_num=20, buffer_size=100000, max_grad_norm=0.5, target_kl=None))
    rng = np.random.default_rng(0)
    eng.set_params((0.1 * rng.standard_normal(eng.n_params)).astype(np.float32))
    ids = np.arange(20)
    for t in range(1000):
        eng.push(ids, rng.standard_normal((20, 8)).astype(np.float32), rng.standard_normal((20, 2)).astype(np.float32),
                 rng.normal(0.5, 0.5, 20), (rng.random(20) < 0.1).astype(np.float64), np.zeros(20, bool),
                 np.full(20, t % 250 == 249), rng.standard_normal((20, 8)).astype(np.float32))
    eng.sync()
    for k in range(3):
        eng.optim_reset(); eng.ppo_update([0.75], 1 / 1.75, 256, 4, seed=k + 1)
    eng.sync(); t0 = time.perf_counter()
    for k in range(10):
        eng.optim_reset(); st, _ = eng.ppo_update([0.75], 1 / 1.75, 256, 4, seed=k + 10)
    eng.sync(); dt = (time.perf_counter() - t0) / 10
    print(f"H={H}: {dt * 1e3:.2f} ms per update, {dt / st.shape[0] * 1e6:.1f} us per optimiser step ({st.shape[0]} steps)")
    eng.close()
