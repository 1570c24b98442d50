#!/usr/bin/env python
"""What one device-actor call of the collector costs (fsrl_actor_sample through ctypes: launch, kernel, completion words), by
row count -- the part of a vector step a persistent actor could shorten."""
import ctypes as C
import json
import time

import numpy as np

from fsrl_amd import _lib
from fsrl_amd.engine import Engine, EngineConfig

import sys
eng = Engine(EngineConfig(obs_dim=8, act_dim=2, hidden=256, env_num=32, max_grad_norm=0.5, target_kl=None))
resident = "--launch" not in sys.argv            # --launch: one kernel launch per call (fsrl_actor_set_resident(0))
eng.actor_set_resident(resident)
out = {"resident": resident}
for k in (1, 16, 20, 32):
    obs = np.random.default_rng(0).standard_normal((k, 8)).astype(np.float32)
    act = np.empty((k, 2), np.float32)
    po, pa = obs.ctypes.data_as(C.POINTER(C.c_float)), act.ctypes.data_as(C.POINTER(C.c_float))
    f = eng.lib.fsrl_actor_sample
    for _ in range(200):
        f(eng._ctx, po, k, 0, 0, pa)
    n = 5000
    t0 = time.perf_counter()
    for _ in range(n):
        f(eng._ctx, po, k, 0, 0, pa)
    out[f"k{k}_us_per_call"] = round((time.perf_counter() - t0) / n * 1e6, 2)
out.update(eng.actor_resident_stats())
print(json.dumps(out))
eng.close()
