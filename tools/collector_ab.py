#!/usr/bin/env python
"""The end-to-end leg of bench.py (host collector over the worker-process env -> HIP store -> update) at a given env step cost, for
several worker-process counts: env-steps/s and the fraction of the env bound.  usage: collector_ab.py [busy_us] [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    busy = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    for rep in range(reps):
        for w in (32, 16, 8):
            r = bench.end_to_end(0, 0, seconds=3.0, device_actor=True, workers=w, busy_us=busy, envs=32, cap_workers=False)
            print(json.dumps({k: r[k] for k in ("workers", "worker_processes", "busy_us", "env_steps_per_s", "frac_of_env_bound",
                                                 "env_bound_env_steps_per_s", "split_phase", "host_cpus_usable")}), flush=True)
