#!/usr/bin/env python
"""The host vector-env side of the headline metric on its own: the whole training loop (collector over
`ShmemVectorEnv` worker processes -> HIP store -> PPO-Lag update) for {4, 32} workers x {0, 100} us of simulated step cost,
futex handshake (libfsrl_env.so), split-phase collection on (the bench default) and off (`--no-split`), optional spin before the
sleep (`--spin-us`).  One JSON line per configuration."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--spin-us", type=float, default=None)
    ap.add_argument("--no-split", action="store_true")
    a = ap.parse_args()
    if a.no_split:
        from fsrl_amd.data import fast_collector
        orig_c = fast_collector.FastCollector.__init__

        def init_c(self, *args, **kw):
            kw["split_phase"] = False
            orig_c(self, *args, **kw)
        fast_collector.FastCollector.__init__ = init_c
    if a.spin_us is not None:
        from fsrl_amd.env import shmem
        orig = shmem.ShmemVectorEnv.__init__

        def init(self, *args, **kw):
            kw.setdefault("spin_us", a.spin_us)
            orig(self, *args, **kw)
        shmem.ShmemVectorEnv.__init__ = init
    for w in (4, 32):
        for b in (0.0, 100.0):
            r = bench.end_to_end(0, 0, seconds=a.seconds, device_actor=True, workers=w, busy_us=b, envs=32)
            print(json.dumps({"workers": w, "busy_us": b, "spin_us": r["handshake"], "split_phase": r["split_phase"],
                              "env_steps_per_s": round(r["env_steps_per_s"]),
                              "collector_only_env_steps_per_s": round(r["collector_only_env_steps_per_s"]),
                              "env_bound": r["env_bound_env_steps_per_s"], "frac_of_env_bound": r["frac_of_env_bound"]}), flush=True)
