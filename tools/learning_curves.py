#!/usr/bin/env python
"""Results evidence in the spirit of BASELINE.md's returns / costs tables: every agent of the reference trained through the
drop-in surface (Agent.learn -> trainer -> collector -> HIP store -> HIP update) on the synthetic safety task
(reward = s0*a0 - 0.1|a|^2 + 0.5, cost = [|s1| > 1], 100-step episodes, cost limit 20; no simulator in this image), one JSON
line per agent with the per-epoch training reward / cost, the multiplier, wall time and env-steps/s.
    python tools/learning_curves.py [--epochs 20] [--agents ppol,cpo,...] > profiles/rNN_learning_curves.json"""
import argparse
import contextlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--agents", default="ppol,focops,cpo,trpol,sacl,ddpgl,cvpo")
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--hidden-sizes", default="", help="e.g. 64x48x32: hidden_sizes as a tuple (layered contexts when it is not two "
                                                       "layers of at most 256 units); overrides --hidden")
    ap.add_argument("--task", choices=["synthetic", "point-circle"], default="synthetic",
                    help="point-circle: per-instance gym-style envs (fsrl_amd.env.PointCircleEnv) built from factories and stepped in "
                         "worker processes, hidden_sizes (100, 50): the env-factory path and the zero-padded widths end to end")
    ap.add_argument("--cost-limit", type=float, default=20.0)
    a = ap.parse_args()
    from fsrl_amd.agent import CPOAgent, CVPOAgent, DDPGLagAgent, FOCOPSAgent, PPOLagAgent, SACLagAgent, TRPOLagAgent
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    from fsrl_amd.utils import BaseLogger
    on = dict(repeat_per_collect=4, batch_size=256)
    full = dict(repeat_per_collect=2, batch_size=99999)
    off = dict(update_per_step=0.2, batch_size=256)
    table = {"ppol": (PPOLagAgent, dict(max_grad_norm=0.5), on), "focops": (FOCOPSAgent, {}, on), "cpo": (CPOAgent, {}, full),
             "trpol": (TRPOLagAgent, {}, full), "sacl": (SACLagAgent, dict(buffer_size=50000), off),
             "ddpgl": (DDPGLagAgent, dict(buffer_size=50000), off), "cvpo": (CVPOAgent, dict(buffer_size=50000), off)}
    for name in a.agents.split(","):
        cls, akw, lkw = table[name]
        if a.task == "point-circle":
            from fsrl_amd.env import DummyVectorEnv, PointCircleEnv, ShmemVectorEnv
            fns = [lambda: PointCircleEnv(max_episode_steps=100) for _ in range(10)]
            env = ShmemVectorEnv(fns, workers=5, seed=0)
            test = DummyVectorEnv(fns[:4], seed=500)
            hidden = (100, 50)
        else:
            env = SyntheticSafetyVectorEnv(env_num=10, episode_len=100, seed=0)
            test = SyntheticSafetyVectorEnv(env_num=4, episode_len=100, seed=5)
            hidden = (a.hidden, a.hidden)
        if a.hidden_sizes:
            hidden = tuple(int(x) for x in a.hidden_sizes.split("x"))
        agent = cls(env, BaseLogger(tempfile.mkdtemp(), name=name), cost_limit=a.cost_limit, device="cuda:0", seed=1,
                    hidden_sizes=hidden, training_num=10, **akw)
        before = agent.evaluate(test, eval_episodes=8)
        curve, t0, steps = [], time.time(), 0
        for ep in range(a.epochs):          # one epoch per learn() call: the statistics of that epoch come back
            with contextlib.redirect_stdout(sys.stderr):      # the logger's "Early stop ..." lines stay out of the JSON
                _, stat, info = agent.learn(env, None, epoch=1, episode_per_collect=10, step_per_epoch=2000, verbose=False,
                                            save_ckpt=False, device_actor=True, **lkw)
            steps += 2000
            curve.append([round(float(stat.get("train/reward", 0.0)), 2), round(float(stat.get("train/cost", 0.0)), 2),
                          round(float(stat.get("loss/lagrangian", stat.get("loss/optim_nu", stat.get("loss/nu_value", 0.0)))), 4)])
        wall = time.time() - t0
        after = agent.evaluate(test, eval_episodes=8)
        print(json.dumps({"agent": name, "task": a.task, "hidden": list(hidden), "epochs": a.epochs, "cost_limit": a.cost_limit,
                          "eval_before_reward_len_cost": [round(float(x), 2) for x in before],
                          "eval_after_reward_len_cost": [round(float(x), 2) for x in after],
                          "curve_train_reward_cost_multiplier": curve, "wall_s": round(wall, 3),
                          "env_steps_per_s_incl_updates": round(steps / wall)}), flush=True)
        agent.policy.engine.close()
        env.close()


if __name__ == "__main__":
    main()
