import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from fsrl_amd.engine import Engine, EngineConfig
eng = Engine(EngineConfig(obs_dim=8, act_dim=2, hidden=256, env_num=20, max_grad_norm=0.5, target_kl=None))
eng.set_params(bench.orthogonal_theta(0, eng.n_params))
obs, act, rew, cost, term, trunc = bench.make_inputs(0)
ids = np.arange(20)
for t in range(1000):
    eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
for k in range(30):
    stats, _ = eng.ppo_update(np.array([0.75]), 1/1.75, 256, 4, seed=k + 1)
    fin = np.isfinite(stats).all(1)
    print(k, "finite", fin.all(), "kl max", np.nanmax(np.abs(stats[:, 5])), "entropy", stats[-1, 10], "vf0", stats[-1, 6], "actor_total", stats[-1, 4])
    if not fin.all():
        bad = np.argmin(fin); print(" first bad step", bad, stats[max(0, bad - 2):bad + 1]); break
