#!/usr/bin/env python
"""bench.end_to_end legs with the resident-actor counters (kernel launches vs calls served through the doorbell) and the native
collect loop's own split of a vector step (seconds waiting for the env workers | in store + actor calls: fsrl_collect_timing).
--launch: one kernel launch per actor call (fsrl_actor_set_resident(0)) for the same legs."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import bench
    from fsrl_amd import engine as E
    acc = {"t_env": 0.0, "t_act": 0.0, "steps": 0}
    resident = "--launch" not in sys.argv
    orig_init, orig_ce = E.Engine.__init__, E.Engine.collect_episodes

    def init(self, *a, **k):
        orig_init(self, *a, **k)
        self.actor_set_resident(resident)

    def ce(self, *a, **k):
        r = orig_ce(self, *a, **k)
        acc["t_env"] += r["t_env"]; acc["t_act"] += r["t_act"]; acc["steps"] += r["steps"]
        return r
    E.Engine.__init__, E.Engine.collect_episodes = init, ce
    for w, busy, envs in ((0, 0.0, 20), (4, 0.0, 32), (32, 100.0, 32)):
        acc.update(t_env=0.0, t_act=0.0, steps=0)
        r = bench.end_to_end(0, 0, seconds=2.0, device_actor=True, workers=w, busy_us=busy, envs=envs, cap_workers="auto" if w else False)
        o = {k: r[k] for k in ("workers", "busy_us", "env_steps_per_s", "collector_only_env_steps_per_s", "split_phase", "actor_resident", "collects")}
        o["resident"] = resident
        if acc["steps"]:
            vs = acc["steps"] / envs
            o["per_vector_step_us"] = {"env_wait": round(acc["t_env"] / vs * 1e6, 2), "store_and_actor": round(acc["t_act"] / vs * 1e6, 2)}
        print(json.dumps(o))


if __name__ == "__main__":
    main()
