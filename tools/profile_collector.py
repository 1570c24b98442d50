#!/usr/bin/env python
"""Where does a vector step of FastCollector (device_actor path) spend its time?  cProfile over a few collects."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsrl_amd.agent import PPOLagAgent  # noqa: E402
from fsrl_amd.data import FastCollector, HipVectorReplayBuffer  # noqa: E402
from fsrl_amd.env import SyntheticSafetyVectorEnv  # noqa: E402

env = SyntheticSafetyVectorEnv(env_num=20, obs_dim=8, act_dim=2, episode_len=300, seed=0)
agent = PPOLagAgent(env, cost_limit=10, device="cuda:0", seed=0, hidden_sizes=(256, 256), training_num=20)
agent.policy.train()
buf = HipVectorReplayBuffer(agent.policy.engine, 100000, 20)
col = FastCollector(agent.policy, env, buf, exploration_noise=True, device_actor=True)
col.collect(n_episode=20)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    buf.reset()
    col.collect(n_episode=20)
pr.disable()
print("env-steps/s", col.collect_step / col.collect_time)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
