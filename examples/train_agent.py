#!/usr/bin/env python
"""Train any of the HIP-backed agents the way the reference's examples/mlp/train_*_agent.py do, on the
synthetic SafetyCarCircle-shaped vector env (this image has no safety-gymnasium / bullet-safety-gym; with them
installed pass real vector envs exposing reset(ids) / step(act, ids) -> (obs, rew, terminated, truncated, info["cost"])).

    python examples/train_agent.py --algo ppol --epoch 3
    python examples/train_agent.py --algo sacl --epoch 2 --device-actor
"""
import argparse
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsrl_amd.agent import CPOAgent, CVPOAgent, DDPGLagAgent, FOCOPSAgent, PPOLagAgent, SACLagAgent, TRPOLagAgent  # noqa: E402
from fsrl_amd.env import SyntheticSafetyVectorEnv  # noqa: E402
from fsrl_amd.utils import BaseLogger  # noqa: E402

AGENTS = {"ppol": PPOLagAgent, "cpo": CPOAgent, "trpol": TRPOLagAgent, "focops": FOCOPSAgent, "sacl": SACLagAgent,
          "ddpgl": DDPGLagAgent, "cvpo": CVPOAgent}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", choices=sorted(AGENTS), default="ppol")
    ap.add_argument("--epoch", type=int, default=3)
    ap.add_argument("--envs", type=int, default=20)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--hidden-sizes", default="", help="hidden_sizes as a tuple, e.g. 64x48x32 or 512x512 (the reference's agents take any tuple); "
                                                       "two layers of at most 256 units run on the fused kernels, anything else on the layered ones")
    ap.add_argument("--cost-limit", type=float, default=10.0)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--seed", type=int, default=10)
    ap.add_argument("--device-actor", action="store_true", help="collector actions from fsrl_actor_sample (library RNG)")
    ap.add_argument("--batch-size", type=int, default=256, help="minibatch size of PPO-Lag / FOCOPS and the off-policy agents")
    ap.add_argument("--tr-batch-size", type=int, default=99999,
                    help="CPO / TRPO-Lag: Batch.split size inside learn (the reference's default 99999 = the whole buffer)")
    ap.add_argument("--task", choices=["synthetic", "point-circle"], default="synthetic",
                    help="synthetic: the vectorised stand-in dynamics; point-circle: per-instance gym-style envs built from factories")
    ap.add_argument("--workers", type=int, default=0, help="> 0: the training envs step in that many worker processes (ShmemVectorEnv)")
    # options of the on-policy agents (the reference's constructor arguments of the same names)
    ap.add_argument("--unbounded", action="store_true", help="actor mean without max_action * tanh")
    ap.add_argument("--reward-normalization", action="store_true", help="critics learn returns / running std")
    ap.add_argument("--value-clip", action="store_true", help="PPO-Lag clipped value loss (needs --reward-normalization)")
    ap.add_argument("--recompute-advantage", action="store_true", help="PPO-Lag / FOCOPS: GAE from the current critics before every pass")
    a = ap.parse_args()
    if a.task == "point-circle":
        # the reference's own construction (train_ppol_agent.py:120-123): one factory per env, in process or in worker processes
        from fsrl_amd.env import DummyVectorEnv, PointCircleEnv, ShmemVectorEnv
        fns = [lambda: PointCircleEnv(max_episode_steps=100) for _ in range(a.envs)]
        env = ShmemVectorEnv(fns, workers=a.workers, seed=a.seed) if a.workers > 0 else DummyVectorEnv(fns, seed=a.seed)
        test_env = DummyVectorEnv(fns[:2], seed=a.seed + 1000)
    elif a.workers > 0:
        from fsrl_amd.env import ShmemVectorEnv
        env = ShmemVectorEnv(env_num=a.envs, workers=a.workers, obs_dim=8, act_dim=2, episode_len=300, seed=a.seed)
    else:
        env = SyntheticSafetyVectorEnv(env_num=a.envs, obs_dim=8, act_dim=2, episode_len=300, seed=a.seed)
    if a.task != "point-circle":
        test_env = SyntheticSafetyVectorEnv(env_num=2, obs_dim=8, act_dim=2, episode_len=300, seed=a.seed + 1)
    logger = BaseLogger(tempfile.mkdtemp(prefix="fsrl_amd_"), name=a.algo)
    kw = dict(cost_limit=a.cost_limit, device=a.device, seed=a.seed, hidden_sizes=(tuple(int(x) for x in a.hidden_sizes.split("x")) if a.hidden_sizes else (a.hidden, a.hidden)),
              training_num=a.envs)
    if a.algo in ("ppol", "cpo", "trpol", "focops"):
        kw.update(unbounded=a.unbounded, reward_normalization=a.reward_normalization)
    if a.algo == "ppol":
        kw.update(value_clip=a.value_clip)
    if a.algo in ("ppol", "focops"):
        kw.update(recompute_advantage=a.recompute_advantage)
    agent = AGENTS[a.algo](env, logger, **kw)
    if a.algo in ("sacl", "ddpgl", "cvpo"):
        out = agent.learn(env, test_env, epoch=a.epoch, episode_per_collect=a.envs, step_per_epoch=6000, update_per_step=0.2,
                          batch_size=a.batch_size, testing_num=2, device_actor=a.device_actor, verbose=True, save_ckpt=False)
    else:
        out = agent.learn(env, test_env, epoch=a.epoch, episode_per_collect=a.envs, step_per_epoch=6000, repeat_per_collect=4,
                          batch_size=a.batch_size if a.algo in ("ppol", "focops") else a.tr_batch_size, testing_num=2, device_actor=a.device_actor,
                          verbose=True, save_ckpt=False)
    print("final:", {k: round(float(v), 4) for k, v in out[1].items() if isinstance(v, (int, float))})
    rew, length, cost = agent.evaluate(test_env, eval_episodes=2)
    print(f"eval: reward {rew:.2f} length {length:.1f} cost {cost:.2f}")


if __name__ == "__main__":
    main()
