#!/usr/bin/env python
"""Does the step's per-process slow mode come with the library's code object?  Two COPIES of libfsrl_hip.so (two code objects, loaded one
after the other) in one process, engines alternating between them: us per optimiser step of the headline update for each."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fsrl_amd import _lib  # noqa: E402
from fsrl_amd.engine import Engine, EngineConfig  # noqa: E402

tmp = tempfile.mkdtemp()
libs = []
for k in range(2):
    path = os.path.join(tmp, f"libfsrl_copy{k}.so")
    shutil.copy(_lib.LIB_PATH, path)
    _lib._lib = None; _lib.LIB_PATH = path
    libs.append(_lib.load())
obs, act, rew, cost, term, trunc = bench.make_inputs(0)
ids = np.arange(bench.ENVS)
out = []
for k in range(6):
    _lib._lib = libs[k & 1]
    e = Engine(EngineConfig(obs_dim=bench.OBS, act_dim=bench.ACT, hidden=bench.HID, env_num=bench.ENVS, buffer_size=100000,
                            max_grad_norm=0.5, target_kl=None))
    for t in range(bench.NROWS // bench.ENVS):
        e.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    e.sync()
    th = bench.orthogonal_theta(0, e.n_params)
    e.set_params(th); e.optim_reset(); e.state_snapshot()
    ts = []
    for u in range(8):
        e.state_restore()
        t0 = time.perf_counter(); st, _ = e.ppo_update([0.75], 1 / 1.75, bench.BATCH, bench.REPEAT, seed=u + 1); e.sync()
        ts.append(time.perf_counter() - t0)
    out.append(float(np.median(ts[2:])) * 1e6 / st.shape[0])
    e.close()
print("us per step, engines alternating between two copies of the library (A B A B A B):", " ".join(f"{x:.2f}" for x in out))
