#!/usr/bin/env python
"""Does the resident actor's idle timeout fire where it says?  (wall_clock64 is taken as 100 MHz: timeout_ticks = us * 100.)
idle_timeout_us = 1000: calls 0.4 ms apart must share one kernel, calls 3 ms apart must each find it gone."""
import json
import time

import numpy as np

from fsrl_amd.engine import Engine, EngineConfig

eng = Engine(EngineConfig(obs_dim=8, act_dim=2, hidden=256, env_num=16, max_grad_norm=0.5, target_kl=None))
eng.actor_set_resident(True, idle_timeout_us=1000.0)
obs = np.zeros((16, 8), np.float32)
eng.actor_forward(obs)


def run(gap_s, n=40):
    l0 = eng.actor_resident_stats()["launches"]
    for _ in range(n):
        t = time.perf_counter()
        while time.perf_counter() - t < gap_s:
            pass
        eng.actor_forward(obs)
    return eng.actor_resident_stats()["launches"] - l0


print(json.dumps({"relaunches_of_40_calls": {"gap_0.4ms": run(0.0004), "gap_0.8ms": run(0.0008), "gap_1.3ms": run(0.0013), "gap_3ms": run(0.003)}}))
eng.close()
