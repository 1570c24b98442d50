#!/usr/bin/env python
"""PPO-Lag update time against the observation / action width (256x256, N = 20 000, batch 256, 4 passes): looks for cliffs
outside the headline shape (obs 8 / act 2)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsrl_amd.engine import Engine, EngineConfig  # noqa: E402

for Do, Da in ((8, 2), (16, 2), (17, 6), (33, 8), (60, 2), (128, 16)):
    eng = Engine(EngineConfig(obs_dim=Do, act_dim=Da, hidden=256, env_num=20, buffer_size=100000, max_grad_norm=0.5, target_kl=None))
    rng = np.random.default_rng(0)
    eng.set_params((0.05 * rng.standard_normal(eng.n_params)).astype(np.float32))
    ids = np.arange(20)
    for t in range(1000):
        eng.push(ids, rng.standard_normal((20, Do)).astype(np.float32), (0.3 * rng.standard_normal((20, Da))).astype(np.float32),
                 rng.normal(0.5, 0.5, 20), (rng.random(20) < 0.1).astype(np.float64), np.zeros(20, bool),
                 np.full(20, t % 250 == 249), rng.standard_normal((20, Do)).astype(np.float32))
    eng.sync()
    theta = eng.get_params()
    for k in range(2):
        eng.set_params(theta); eng.optim_reset(); eng.ppo_update([0.75], 1 / 1.75, 256, 4, seed=k + 1)
    eng.sync(); t0 = time.perf_counter()
    for k in range(5):
        eng.set_params(theta); eng.optim_reset(); st, _ = eng.ppo_update([0.75], 1 / 1.75, 256, 4, seed=k + 10)
    eng.sync(); dt = (time.perf_counter() - t0) / 5
    print(f"obs {Do} act {Da}: {dt * 1e3:.2f} ms per update, {dt / st.shape[0] * 1e6:.1f} us per optimiser step ({st.shape[0]} steps)", flush=True)
    eng.close()
