import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from fsrl_amd.engine import Engine, EngineConfig
from test_gpu_fullsize import _inputs
envs, T, obs_dim = 20, 1000, 60
rng = np.random.default_rng(11)
obs, act, rew, cost, term, trunc = _inputs(rng, envs, T, obs_dim, 2, 250)
eng = Engine(EngineConfig(obs_dim=obs_dim, act_dim=2, hidden=256, env_num=envs, target_kl=None, lr=1e-3))
ids = np.arange(envs)
for t in range(T):
    eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
theta = (0.1 * np.random.default_rng(3).standard_normal(eng.n_params)).astype(np.float32)
def run(tr, hv, wg, split):
    eng.tr_set_plan(tr, hv, wg); eng.tr_set_tile_split(*split)
    eng.set_params(theta); eng.optim_reset()
    eng.tr_begin(target_kl=0.01, l2_reg=0.001, critic_lr=1e-3, max_backtracks=10, optim_critic_iters=3, cost_limit=10.0)
    return [eng.tr_grad(w) for w in range(3)] + [eng.tr_eval()]
for wg in (3, 0):
    ref = run(16, 1, wg, (-1, -1))
    for name, pl in (("ref again", (16, 1, wg, (-1, -1))), ("r4 mixed", (32, 3, wg, (-1, -1))), ("co static", (0, 2, wg, (-2, -2))), ("co persistent", (0, 2, wg, (-1, -1))),
                     ("co persistent again", (0, 2, wg, (-1, -1))), ("co persistent n32=625", (0, 2, wg, (625, 625))), ("co persistent n32=0", (0, 2, wg, (0, 0)))):
        got = run(*pl)
        for k in range(4):
            d = np.abs(np.asarray(ref[k], np.float64) - got[k])
            if d.max() > 0:
                idx = np.flatnonzero(d > 0)
                print(f"wgrad {wg} | {name}: item {k} differs: max {d.max():.3g} at {idx[:8]} ... n={idx.size} of {d.size}")
        print(f"wgrad {wg} | {name}: done")
eng.close()
