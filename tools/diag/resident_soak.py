#!/usr/bin/env python
"""Soak of the resident actor's end / relaunch protocol: calls spaced AROUND the idle timeout (so that workgroups time out while a
doorbell is being rung), random row counts (1 .. 64: one to four workgroups with rows), parameter uploads in between.  Every answer is
compared with the one-launch-per-call actor's.  usage: resident_soak.py [calls] [idle_timeout_us]"""
import json
import sys
import time

import numpy as np

from fsrl_amd.engine import Engine, EngineConfig

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
mk = lambda: Engine(EngineConfig(obs_dim=8, act_dim=2, hidden=256, env_num=64, max_grad_norm=0.5, target_kl=None))  # noqa: E731
a, b = mk(), mk()
a.actor_set_resident(True, idle_timeout_us=idle)
b.actor_set_resident(False)
rng = np.random.default_rng(0)
thetas = [(0.2 * rng.standard_normal(a.n_params)).astype(np.float32) for _ in range(3)]
pool = [rng.standard_normal((64, 8)).astype(np.float32) for _ in range(8)]
want = {}
for ti, th in enumerate(thetas):
    b.set_params(th)
    for pi, o in enumerate(pool):
        want[(ti, pi)] = [x.copy() for x in b.actor_forward(o)]
ti = 0
a.set_params(thetas[ti])
bad = 0
t_start = time.perf_counter()
for i in range(calls):
    if i % 5000 == 4999:
        ti = (ti + 1) % 3
        a.set_params(thetas[ti])
    gap = rng.uniform(0.6, 1.4) * idle * 1e-6 if i % 3 else 0.0
    t = time.perf_counter()
    while time.perf_counter() - t < gap:
        pass
    pi, k = int(rng.integers(0, 8)), int(rng.integers(1, 65))
    mu, sg = a.actor_forward(pool[pi][:k])
    if not (np.array_equal(mu, want[(ti, pi)][0][:k]) and np.array_equal(sg, want[(ti, pi)][1][:k])):
        bad += 1
st = a.actor_resident_stats()
print(json.dumps({"calls": calls, "idle_timeout_us": idle, "wrong_answers": bad, "kernel_launches": st["launches"],
                  "calls_served": st["requests"], "seconds": round(time.perf_counter() - t_start, 1)}))
a.close(); b.close()
sys.exit(1 if bad else 0)
