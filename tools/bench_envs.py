#!/usr/bin/env python
"""The training loop's env-steps/s against the number of (in-process, zero-cost) envs: the collector's per-vector-step cost must not
grow with the env count faster than the rows it moves."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    for envs in [int(a) for a in sys.argv[1:]] or [20, 64, 256, 1024]:
        r = bench.end_to_end(0, 0, seconds=3.0, device_actor=True, envs=envs)
        print(json.dumps({"envs": envs, "env_steps_per_s": round(r["env_steps_per_s"]), "collector_only": round(r["collector_only_env_steps_per_s"]),
                          "us_per_vector_step": round(envs / r["collector_only_env_steps_per_s"] * 1e6, 1),
                          "update_ms_per_collect": round(r["update_ms_per_collect"], 2)}), flush=True)
