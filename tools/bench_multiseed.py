#!/usr/bin/env python
"""S independent PPO-Lag agents (seeds) on ONE GPU, one host thread and one library context each (every context
has its own HIP streams): aggregate policy-updates/s.  The kernels of one agent leave latency gaps (launch floors,
dependent phases) that another agent's kernels can fill."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from fsrl_amd.engine import Engine, EngineConfig  # noqa: E402


def agent(seed):
    eng = Engine(EngineConfig(obs_dim=B.OBS, act_dim=B.ACT, hidden=B.HID, env_num=B.ENVS, buffer_size=100000, max_grad_norm=0.5,
                              target_kl=None))
    theta = B.orthogonal_theta(seed, eng.n_params)
    obs, act, rew, cost, term, trunc = B.make_inputs(seed)
    ids = np.arange(B.ENVS)
    for t in range(B.NROWS // B.ENVS):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    eng.sync()
    return eng, theta


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    out = {}
    for S in (1, 2, 3, 4):
        agents = [agent(s) for s in range(S)]
        lag, resc = np.array([0.75]), 1 / 1.75

        def work(i, n):
            eng, theta = agents[i]
            for k in range(n):
                eng.set_params(theta); eng.optim_reset()
                eng.ppo_update(lag, resc, B.BATCH, B.REPEAT, perms=None, seed=1000 * i + k + 1)

        for i in range(S):
            work(i, 1)
        th = [threading.Thread(target=work, args=(i, steps)) for i in range(S)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        out[f"seeds_{S}"] = S * steps / dt
        for eng, _ in agents:
            eng.close()
    print(json.dumps({"metric": "aggregate policy-updates/s, S independent agents on one MI355X", **out}))


if __name__ == "__main__":
    main()
