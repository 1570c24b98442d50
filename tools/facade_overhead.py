#!/usr/bin/env python
"""What the Python facade adds to a PPO-Lag update: PPOLagrangian.update() with a real BaseLogger vs the engine call."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsrl_amd.agent import PPOLagAgent  # noqa: E402
from fsrl_amd.data import FastCollector, HipVectorReplayBuffer  # noqa: E402
from fsrl_amd.env import SyntheticSafetyVectorEnv  # noqa: E402
from fsrl_amd.utils import BaseLogger  # noqa: E402

env = SyntheticSafetyVectorEnv(env_num=20, obs_dim=8, act_dim=2, episode_len=1000, seed=0)
for name, logger, tkl in (("BaseLogger", BaseLogger(tempfile.mkdtemp(), name="x"), None), ("no logger", None, None),
                          ("KL check on", None, 1e6)):       # target_kl set (never reached): one readback per pass
    agent = PPOLagAgent(env, logger, cost_limit=10, device="cuda:0", seed=0, hidden_sizes=(256, 256), max_grad_norm=0.5,
                        target_kl=tkl, training_num=20)
    agent.policy.train()
    buf = HipVectorReplayBuffer(agent.policy.engine, 100000, 20)
    col = FastCollector(agent.policy, env, buf, exploration_noise=True, device_actor=True)
    col.collect(n_episode=20)                                    # 20 000 rows
    for _ in range(2):
        agent.policy.update(0, buf, batch_size=256, repeat=4)
    agent.policy.engine.sync(); t0 = time.perf_counter()
    for _ in range(10):
        agent.policy.update(0, buf, batch_size=256, repeat=4)
    agent.policy.engine.sync()
    print(f"{name:10s}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per PPOLagrangian.update() (N = {len(buf)}, 312 optimiser steps)")
    agent.policy.engine.close()
