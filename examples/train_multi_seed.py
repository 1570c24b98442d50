#!/usr/bin/env python
"""Train several seeds of one agent on ONE MI355X from one process: a thread and a HIP context per seed.

Separate processes sharing a GPU time-slice (2 processes reach 70 updates/s together where one reaches
111); contexts of one process overlap instead -- an agent's dependent kernel chain leaves latency gaps that another agent's
kernels fill (3 agents: ~190 updates/s together).  ctypes releases the GIL inside the library calls, so the update phases of
the seeds overlap; the Python collector loops take turns.

    python examples/train_multi_seed.py --algo ppol --seeds 3 --epoch 2
    python examples/train_multi_seed.py --algo ppol --seeds 4 --epoch 2 --grouped     # PPO-Lag only

--grouped: ONE thread; the seeds collect one after the other and their updates run in lock step through the grouped
launches (fsrl_amd.policy.PolicyGroup -> fsrl_group_ppo_update: every launch of the minibatch step carries all seeds):
~2x the aggregate updates/s of the thread-per-seed mode at 4 seeds (tools/bench_group.py).
"""
import argparse
import os
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsrl_amd.agent import CPOAgent, CVPOAgent, DDPGLagAgent, FOCOPSAgent, PPOLagAgent, SACLagAgent, TRPOLagAgent  # noqa: E402
from fsrl_amd.env import SyntheticSafetyVectorEnv  # noqa: E402
from fsrl_amd.utils import BaseLogger  # noqa: E402

AGENTS = {"ppol": PPOLagAgent, "cpo": CPOAgent, "trpol": TRPOLagAgent, "focops": FOCOPSAgent, "sacl": SACLagAgent,
          "ddpgl": DDPGLagAgent, "cvpo": CVPOAgent}


def run_grouped(a):
    """k PPO-Lag seeds, one host thread: collect each seed's episodes, step each PID multiplier, ONE grouped update."""
    assert a.algo == "ppol", "--grouped: PPO-Lagrangian"
    from fsrl_amd.data import FastCollector, HipVectorReplayBuffer
    from fsrl_amd.policy import PolicyGroup
    agents, cols, bufs = [], [], []
    for seed in range(a.seeds):
        env = SyntheticSafetyVectorEnv(env_num=a.envs, obs_dim=8, act_dim=2, episode_len=300, seed=seed)
        logger = BaseLogger(tempfile.mkdtemp(prefix=f"fsrl_amd_s{seed}_"), name=f"ppol-s{seed}")
        agent = PPOLagAgent(env, logger, cost_limit=10.0, device=a.device, seed=seed, hidden_sizes=(128, 128),
                            training_num=a.envs)
        agent.policy.train()
        buf = HipVectorReplayBuffer(agent.policy.engine, None, a.envs)
        agents.append(agent); bufs.append(buf)
        cols.append(FastCollector(agent.policy, env, buf, exploration_noise=True, device_actor=True))
    group = PolicyGroup([ag.policy for ag in agents])
    t0, steps, updates = time.time(), 0, 0
    for ep in range(a.epoch):
        budget = 6000
        while budget > 0:
            for ag, col in zip(agents, cols):
                st = col.collect(n_episode=a.envs)
                ag.policy.pre_update_fn(stats_train=st)
                ag.logger.store(**{"train/reward": st["rew"], "train/cost": st["cost"]})
                steps += st["n/st"]
            budget -= st["n/st"]
            group.update(bufs, batch_size=256, repeat=4)
            updates += len(agents)
            for col in cols:
                col.reset_buffer(keep_statistics=True)
        for seed, ag in enumerate(agents):
            print(f"epoch {ep + 1} seed {seed}: reward {ag.logger.get_mean('train/reward'):.2f} "
                  f"cost {ag.logger.get_mean('train/cost'):.2f} lambda {ag.policy.lag_optims[0].get_lag():.3f}")
            ag.logger.write(steps, display=False)
    dt = time.time() - t0
    print(f"{a.seeds} seeds x {a.epoch} epochs grouped on {a.device}: {steps / dt:.0f} env-steps/s, {updates / dt:.1f} updates/s "
          f"aggregate in {dt:.1f} s")
    group.close()
    for ag in agents:
        ag.policy.engine.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", choices=sorted(AGENTS), default="ppol")
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--epoch", type=int, default=2)
    ap.add_argument("--envs", type=int, default=20)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--grouped", action="store_true", help="PPO-Lag: lock-step grouped updates from one thread")
    a = ap.parse_args()
    out, errs = {}, []
    if a.grouped:
        return run_grouped(a)

    def run(seed):
        try:
            env = SyntheticSafetyVectorEnv(env_num=a.envs, obs_dim=8, act_dim=2, episode_len=300, seed=seed)
            logger = BaseLogger(tempfile.mkdtemp(prefix=f"fsrl_amd_s{seed}_"), name=f"{a.algo}-s{seed}")
            agent = AGENTS[a.algo](env, logger, cost_limit=10.0, device=a.device, seed=seed, hidden_sizes=(128, 128),
                                   training_num=a.envs)
            kw = dict(epoch=a.epoch, episode_per_collect=a.envs, step_per_epoch=6000, device_actor=True, verbose=False,
                      save_ckpt=False, show_progress=False)
            if a.algo in ("sacl", "ddpgl", "cvpo"):
                ep, stat, info = agent.learn(env, None, update_per_step=0.2, batch_size=256, **kw)
            else:
                ep, stat, info = agent.learn(env, None, repeat_per_collect=4,
                                             batch_size=256 if a.algo in ("ppol", "focops") else 99999, **kw)
            out[seed] = {k: round(float(v), 3) for k, v in stat.items() if k in ("train/reward", "train/cost", "update/env_step")}
        except Exception as e:      # a failing seed must not hang the others
            errs.append((seed, repr(e)))

    t0 = time.time()
    threads = [threading.Thread(target=run, args=(s, )) for s in range(a.seeds)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for s in sorted(out):
        print(f"seed {s}: {out[s]}")
    print(f"{len(out)} seeds x {a.epoch} epochs in {time.time() - t0:.1f} s on {a.device}")
    if errs:
        raise SystemExit(f"failed seeds: {errs}")


if __name__ == "__main__":
    main()
