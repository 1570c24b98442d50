import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from fsrl_amd.engine import Engine, EngineConfig
eng = Engine(EngineConfig(obs_dim=8, act_dim=2, hidden=256, env_num=20, buffer_size=100000, max_grad_norm=0.5, target_kl=None))
theta = bench.orthogonal_theta(0, eng.n_params)
for _ in range(5): eng.set_params(theta); eng.optim_reset()
eng.sync(); t0 = time.perf_counter()
for _ in range(200): eng.set_params(theta)
eng.sync(); t1 = time.perf_counter()
for _ in range(200): eng.optim_reset()
eng.sync(); t2 = time.perf_counter()
print("set_params %.1f us, optim_reset %.1f us" % ((t1 - t0) / 200 * 1e6, (t2 - t1) / 200 * 1e6))
