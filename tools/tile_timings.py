"""Per-tile timings of the full-batch kernels: the cached HVP and the tile kernel on batches of exactly k x 256 32-row tiles (and
16-row ones) under every kernel plan; run under `rocprofv3 --kernel-trace` and reduce with tools/trace_by_grid.py."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from fsrl_amd.engine import Engine, EngineConfig
from test_gpu_fullsize import _inputs
for N in (4096, 8192, 16384, 24576):
    envs = 16; T = N // envs
    rng = np.random.default_rng(11)
    obs, act, rew, cost, term, trunc = _inputs(rng, envs, T, 60, 2, 250)
    eng = Engine(EngineConfig(obs_dim=60, act_dim=2, hidden=256, env_num=envs, target_kl=None, lr=1e-3))
    ids = np.arange(envs)
    for t in range(T):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    theta = (0.1 * np.random.default_rng(3).standard_normal(eng.n_params)).astype(np.float32)
    v = np.random.default_rng(4).standard_normal(eng.n_actor_params).astype(np.float32)
    for (tr, hv, split) in ((32, 3, (-1, -1)), (0, 0, (N // 32, N // 32)), (0, 0, (0, 0)), (16, 3, (-1, -1))):
        eng.tr_set_plan(tr, hv, 0); eng.tr_set_tile_split(*split)
        eng.set_params(theta); eng.optim_reset()
        eng.tr_begin(target_kl=0.01, l2_reg=0.001, critic_lr=1e-3, max_backtracks=10, optim_critic_iters=3, cost_limit=10.0)
        eng.tr_hvp(v)
        for _ in range(6):
            eng.tr_hvp_cached(v)
            eng.tr_grad(2)
    eng.close()
