#!/usr/bin/env python
"""bench.end_to_end legs with the resident-actor counters (kernel launches vs calls served through the doorbell)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import bench
    for w, busy in ((0, 0.0), (4, 0.0), (32, 100.0)):
        r = bench.end_to_end(0, 0, seconds=2.0, device_actor=True, workers=w, busy_us=busy, envs=(20 if w == 0 else 32),
                             cap_workers="auto" if w else False)
        print(json.dumps({k: r[k] for k in ("workers", "busy_us", "env_steps_per_s", "collector_only_env_steps_per_s", "collector_loop",
                                            "actor_resident", "collects")}))


if __name__ == "__main__":
    main()
