#!/usr/bin/env python
"""Golden vectors G11: a CLOSED training loop -- collect with the policy, update, collect with the updated policy ...
-- driven by the UNMODIFIED reference PPOLagrangian (forward / map_action / pre_update_fn / update) over the shim's
VectorReplayBuffer, on the build-owned synthetic vector env.  Records what a learning curve is made of: per collect
the mean episode reward / cost / length, the PID multiplier, the first and last logged minibatch rows, and the
parameters at the end.  The rollout loop itself (n_episode = env_num, lock-step) is written here and again in
tests/test_gpu_loop.py: what is pinned is acting + storing + updating + the PID controller across cycles.

    python tests/golden/gen_golden_loop.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ref_shim.install()
from gen_golden import CaptureLogger, build_ppo, flat_params, seed_all  # noqa: E402
from ref_shim import Batch, VectorReplayBuffer  # noqa: E402

from fsrl_amd.env.synthetic import SyntheticSafetyVectorEnv  # noqa: E402  (numpy only; no engine is touched)


def rollout(policy, env, buf, noise=False):
    """One collect of exactly env_num episodes (every env runs one episode, lock-step)."""
    obs, _ = env.reset()
    E = len(env)
    ids = np.arange(E)
    ep_rew, ep_cost, steps = np.zeros(E), np.zeros(E), 0
    while True:
        with torch.no_grad():
            res = policy(Batch(obs=obs, info={}), None)
        act = res.act.numpy()
        if noise:
            act = policy.exploration_noise(act, None)
        obs_next, rew, term, trunc, info = env.step(policy.map_action(act), ids)
        buf.add({"obs": obs, "act": act, "rew": rew, "terminated": term, "truncated": trunc, "done": term | trunc,
                 "obs_next": obs_next, "info.cost": info["cost"]}, ids)
        ep_rew += rew; ep_cost += info["cost"]; steps += E
        obs = obs_next
        if (term | trunc).all():
            break
    return dict(reward=float(ep_rew.mean()), cost=float(ep_cost.mean()), steps=steps)


def gen(name, obs_dim, act_dim, hidden, env_num, ep_len, cycles, batch_size, repeat, seed, cost_limit, **kw):
    logger = CaptureLogger()
    policy, ac, _ = build_ppo(obs_dim, act_dim, hidden, seed, logger=logger, cost_limit=cost_limit, **kw)
    policy.train()
    env = SyntheticSafetyVectorEnv(env_num=env_num, obs_dim=obs_dim, act_dim=act_dim, episode_len=ep_len, seed=seed + 11)
    buf = VectorReplayBuffer(env_num * ep_len * 2, env_num)
    out = {"theta0": flat_params(ac)}
    seed_all(seed + 7)
    curve, first_rows, last_rows, lags, steps_per = [], [], [], [], []
    for c in range(cycles):
        buf.reset()
        st = rollout(policy, env, buf)
        policy.pre_update_fn(stats_train={"cost": st["cost"]})
        lags.append([o.get_lag() for o in policy.lag_optims])
        n0 = len(logger.rows)
        policy.update(0, buf, batch_size=batch_size, repeat=repeat)
        rows = [r for r in logger.rows[n0:] if "update/gradient_steps" not in r]
        assert len(rows) % 3 == 0                       # actor stats, critic stats, {total, entropy} per minibatch
        keys = list(rows[0].keys()) + list(rows[1].keys()) + list(rows[2].keys())
        first_rows.append([{**rows[0], **rows[1], **rows[2]}[k] for k in keys])
        last_rows.append([{**rows[-3], **rows[-2], **rows[-1]}[k] for k in keys])
        steps_per.append(len(rows) // 3)
        curve.append([st["reward"], st["cost"], st["steps"]])
    out.update(curve=np.array(curve), first_rows=np.array(first_rows), last_rows=np.array(last_rows), lagrangians=np.array(lags),
               steps_per_update=np.array(steps_per), stat_keys=np.array(keys), theta_final=flat_params(ac))
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, ep_len=ep_len, cycles=cycles,
               batch_size=batch_size, repeat=repeat, seed=seed, cost_limit=cost_limit, max_action=1.0, lr=5e-4,
               lagrangian_pid=[0.05, 0.0005, 0.1])
    cfg.update(kw)
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"loop_{name}.npz"), **out)
    print(f"G11 loop_{name}.npz cycles={cycles} reward {curve[0][0]:.2f} -> {curve[-1][0]:.2f} cost {curve[0][1]:.2f} -> {curve[-1][1]:.2f} "
          f"lag {lags[0][0]:.3f} -> {lags[-1][0]:.3f} steps/update {steps_per[0]}")


def gen_trust(kind, name, obs_dim, act_dim, hidden, env_num, ep_len, cycles, repeat, seed, cost_limit):
    """CPO / TRPO-Lagrangian closed loop (full-batch updates, batch_size = 99999)."""
    from fsrl.policy import CPO, TRPOLagrangian
    from gen_golden_trust import build_nets, dist
    from torch import nn
    from ref_shim import _Box
    actor, critic, ac = build_nets(obs_dim, act_dim, hidden, seed)
    logger = CaptureLogger()
    spaces = dict(observation_space=_Box(-np.inf, np.inf, (obs_dim, )), action_space=_Box(-1, 1, (act_dim, )))
    if kind == "cpo":
        policy = CPO(actor, critic, torch.optim.Adam(nn.ModuleList(critic).parameters(), lr=1e-3), dist, logger=logger,
                     cost_limit=cost_limit, optim_critic_iters=10, **spaces)
    else:
        policy = TRPOLagrangian(actor, critic, torch.optim.Adam(nn.ModuleList(critic).parameters(), lr=1e-3), dist, logger=logger,
                                cost_limit=cost_limit, optim_critic_iters=10, **spaces)
    policy.train()
    env = SyntheticSafetyVectorEnv(env_num=env_num, obs_dim=obs_dim, act_dim=act_dim, episode_len=ep_len, seed=seed + 11)
    buf = VectorReplayBuffer(env_num * ep_len * 2, env_num)
    out = {"theta0": flat_params(ac)}
    seed_all(seed + 7)
    curve, last_rows = [], []
    for c in range(cycles):
        buf.reset()
        st = rollout(policy, env, buf)
        policy.pre_update_fn(stats_train={"cost": st["cost"]})
        n0 = len(logger.rows)
        policy.update(0, buf, batch_size=99999, repeat=repeat)
        rows = [r for r in logger.rows[n0:] if "update/gradient_steps" not in r]
        per = len(rows) // repeat
        keys = [k for r in rows[-per:] for k in r.keys()]
        merged = {}
        for r in rows[-per:]:
            merged.update(r)
        last_rows.append([merged[k] for k in keys])
        if c == 0:
            out["theta_after_first"] = flat_params(ac)
            first_merged = [dict(r) for r in rows]
            out["first_update_rows"] = np.array([[m.get(k, np.nan) for k in keys] for m in
                                                 [{**rows[i * per + 0], **(rows[i * per + 1] if per > 1 else {}),
                                                   **(rows[i * per + 2] if per > 2 else {})} for i in range(repeat)]])
        curve.append([st["reward"], st["cost"], st["steps"]])
    out.update(curve=np.array(curve), last_rows=np.array(last_rows), stat_keys=np.array(keys), theta_final=flat_params(ac))
    cfg = dict(kind=kind, obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, ep_len=ep_len, cycles=cycles,
               repeat=repeat, seed=seed, cost_limit=cost_limit, lr=1e-3, optim_critic_iters=10)
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"loop_{name}.npz"), **out)
    print(f"G11 loop_{name}.npz cycles={cycles} reward {curve[0][0]:.2f} -> {curve[-1][0]:.2f} cost {curve[0][1]:.2f} -> {curve[-1][1]:.2f} keys {keys}")


def gen_focops(name, obs_dim, act_dim, hidden, env_num, ep_len, cycles, batch_size, repeat, seed, cost_limit):
    from fsrl.policy import FOCOPS
    from gen_golden_trust import build_nets, dist
    from torch import nn
    from ref_shim import _Box
    actor, critic, ac = build_nets(obs_dim, act_dim, hidden, seed)
    logger = CaptureLogger()
    policy = FOCOPS(actor, critic, torch.optim.Adam(actor.parameters(), lr=5e-4),
                    torch.optim.Adam(nn.ModuleList(critic).parameters(), lr=1e-3), dist, logger=logger, cost_limit=cost_limit,
                    nu=(2.0, 1e-2, torch.zeros(1)), observation_space=_Box(-np.inf, np.inf, (obs_dim, )),
                    action_space=_Box(-1, 1, (act_dim, )))
    policy.train()
    env = SyntheticSafetyVectorEnv(env_num=env_num, obs_dim=obs_dim, act_dim=act_dim, episode_len=ep_len, seed=seed + 11)
    buf = VectorReplayBuffer(env_num * ep_len * 2, env_num)
    out = {"theta0": flat_params(ac)}
    seed_all(seed + 7)
    curve, last_rows, nus, steps_per = [], [], [], []
    for c in range(cycles):
        buf.reset()
        st = rollout(policy, env, buf)
        policy.pre_update_fn(stats_train={"cost": st["cost"]})
        n0 = len(logger.rows)
        policy.update(0, buf, batch_size=batch_size, repeat=repeat)
        rows = [r for r in logger.rows[n0:] if "update/gradient_steps" not in r]
        assert len(rows) % 3 == 0
        keys = list(rows[0].keys()) + list(rows[1].keys()) + list(rows[2].keys())
        last_rows.append([{**rows[-3], **rows[-2], **rows[-1]}[k] for k in keys])
        steps_per.append(len(rows) // 3)
        nus.append(float(policy._nu))
        curve.append([st["reward"], st["cost"], st["steps"]])
    out.update(curve=np.array(curve), last_rows=np.array(last_rows), nus=np.array(nus), steps_per_update=np.array(steps_per),
               stat_keys=np.array(keys), theta_final=flat_params(ac))
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, ep_len=ep_len, cycles=cycles,
               batch_size=batch_size, repeat=repeat, seed=seed, cost_limit=cost_limit, actor_lr=5e-4, critic_lr=1e-3,
               nu_max=2.0, nu_lr=1e-2)
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"loop_{name}.npz"), **out)
    print(f"G11 loop_{name}.npz cycles={cycles} reward {curve[0][0]:.2f} -> {curve[-1][0]:.2f} cost {curve[0][1]:.2f} -> {curve[-1][1]:.2f} "
          f"nu {nus[0]:.3f} -> {nus[-1]:.3f} steps/update {steps_per}")


def gen_ddpg(name, obs_dim, act_dim, hidden, env_num, ep_len, cycles, batch_size, updates_per_cycle, seed, cost_limit, n_step=3):
    from fsrl.policy import DDPGLagrangian
    from fsrl.utils.net.common import ActorCritic
    from ref_shim import Actor, Critic, Net, _Box
    from torch import nn

    class GaussianNoise:                      # tianshou.exploration.GaussianNoise: numpy's global RNG
        def __init__(self, sigma): self._sigma = sigma
        def __call__(self, size): return np.random.normal(0.0, self._sigma, size)
    seed_all(seed)
    actor = Actor(Net((obs_dim, ), hidden_sizes=hidden), (act_dim, ), max_action=1.0)
    critics = [Critic(Net((obs_dim, ), (act_dim, ), hidden_sizes=hidden, concat=True)) for _ in range(2)]
    for m in ActorCritic(actor, critics).modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    logger = CaptureLogger()
    policy = DDPGLagrangian(actor=actor, critics=critics, actor_optim=torch.optim.Adam(actor.parameters(), lr=1e-4),
                            critic_optim=torch.optim.Adam(nn.ModuleList(critics).parameters(), lr=1e-3), logger=logger,
                            exploration_noise=GaussianNoise(0.1), n_step=n_step, cost_limit=cost_limit,
                            observation_space=_Box(-np.inf, np.inf, (obs_dim, )), action_space=_Box(-1, 1, (act_dim, )))
    policy.train()
    env = SyntheticSafetyVectorEnv(env_num=env_num, obs_dim=obs_dim, act_dim=act_dim, episode_len=ep_len, seed=seed + 11)
    buf = VectorReplayBuffer(env_num * ep_len * cycles, env_num)
    flat = lambda mods: torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()]).numpy().copy()  # noqa: E731
    out = {"theta_actor0": flat([actor]), "theta_critics0": flat(critics)}
    seed_all(seed + 7)
    curve, last_rows, lags = [], [], []
    for c in range(cycles):
        st = rollout(policy, env, buf, noise=True)
        policy.pre_update_fn(stats_train={"cost": st["cost"]})
        lags.append([o.get_lag() for o in policy.lag_optims])
        for _ in range(updates_per_cycle):
            policy.update(batch_size, buf)
        rows = logger.rows
        keys = list(rows[-2].keys()) + list(rows[-1].keys())
        last_rows.append([{**rows[-2], **rows[-1]}[k] for k in keys])
        curve.append([st["reward"], st["cost"], st["steps"]])
    out.update(curve=np.array(curve), last_rows=np.array(last_rows), lagrangians=np.array(lags), stat_keys=np.array(keys),
               theta_actor_final=flat([actor]), theta_critics_final=flat(critics))
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, ep_len=ep_len, cycles=cycles,
               batch_size=batch_size, updates_per_cycle=updates_per_cycle, seed=seed, cost_limit=cost_limit, n_step=n_step,
               actor_lr=1e-4, critic_lr=1e-3, tau=0.05, gamma=0.99, exploration_sigma=0.1)
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"loop_{name}.npz"), **out)
    print(f"G11 loop_{name}.npz cycles={cycles} reward {curve[0][0]:.2f} -> {curve[-1][0]:.2f} cost {curve[0][1]:.2f} -> {curve[-1][1]:.2f} "
          f"lag {lags[0][0]:.3f} -> {lags[-1][0]:.3f}")


def gen_sac(name, obs_dim, act_dim, hidden, env_num, ep_len, cycles, batch_size, updates_per_cycle, seed, cost_limit, n_step=2):
    """Off-policy closed loop: the store accumulates over cycles; `updates_per_cycle` SACLagrangian.update calls after
    every collect (fsrl/trainer/offpolicy.py:100-105)."""
    from fsrl.policy import SACLagrangian
    from fsrl.utils.net.common import ActorCritic
    from fsrl.utils.net.continuous import DoubleCritic
    from ref_shim import ActorProb, Net, _Box
    from torch import nn
    seed_all(seed)
    actor = ActorProb(Net((obs_dim, ), hidden_sizes=hidden), (act_dim, ), max_action=1.0, conditioned_sigma=True, unbounded=True)
    critics = [DoubleCritic(Net((obs_dim, ), (act_dim, ), hidden_sizes=hidden, concat=True),
                            Net((obs_dim, ), (act_dim, ), hidden_sizes=hidden, concat=True)) for _ in range(2)]
    for m in ActorCritic(actor, critics).modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    log_alpha = torch.zeros(1, requires_grad=True)
    logger = CaptureLogger()
    policy = SACLagrangian(actor=actor, critics=critics, actor_optim=torch.optim.Adam(actor.parameters(), lr=5e-4),
                           critic_optim=torch.optim.Adam(nn.ModuleList(critics).parameters(), lr=1e-3), logger=logger,
                           alpha=(-float(act_dim), log_alpha, torch.optim.Adam([log_alpha], lr=3e-4)), n_step=n_step,
                           cost_limit=cost_limit, observation_space=_Box(-np.inf, np.inf, (obs_dim, )),
                           action_space=_Box(-1, 1, (act_dim, )))
    policy.train()
    env = SyntheticSafetyVectorEnv(env_num=env_num, obs_dim=obs_dim, act_dim=act_dim, episode_len=ep_len, seed=seed + 11)
    buf = VectorReplayBuffer(env_num * ep_len * cycles, env_num)           # never wraps
    flat = lambda mods: torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()]).numpy().copy()  # noqa: E731
    out = {"theta_actor0": flat([actor]), "theta_critics0": flat(critics)}
    seed_all(seed + 7)
    curve, last_rows, lags, alphas = [], [], [], []
    for c in range(cycles):
        st = rollout(policy, env, buf)
        policy.pre_update_fn(stats_train={"cost": st["cost"]})
        lags.append([o.get_lag() for o in policy.lag_optims])
        n0 = len(logger.rows)
        for _ in range(updates_per_cycle):
            policy.update(batch_size, buf)
        rows = logger.rows[n0:]
        keys = list(rows[-2].keys()) + list(rows[-1].keys())
        last_rows.append([{**rows[-2], **rows[-1]}[k] for k in keys])
        alphas.append(float(policy._alpha))
        curve.append([st["reward"], st["cost"], st["steps"]])
    out.update(curve=np.array(curve), last_rows=np.array(last_rows), lagrangians=np.array(lags), alphas=np.array(alphas),
               stat_keys=np.array(keys), theta_actor_final=flat([actor]), theta_critics_final=flat(critics),
               theta_critics_old_final=flat(list(policy.critics_old)))
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, ep_len=ep_len, cycles=cycles,
               batch_size=batch_size, updates_per_cycle=updates_per_cycle, seed=seed, cost_limit=cost_limit, n_step=n_step,
               actor_lr=5e-4, critic_lr=1e-3, alpha_lr=3e-4, tau=0.05, gamma=0.99)
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"loop_{name}.npz"), **out)
    print(f"G11 loop_{name}.npz cycles={cycles} reward {curve[0][0]:.2f} -> {curve[-1][0]:.2f} cost {curve[0][1]:.2f} -> {curve[-1][1]:.2f} "
          f"alpha {alphas[0]:.4f} -> {alphas[-1]:.4f} lag {lags[0][0]:.3f} -> {lags[-1][0]:.3f}")


def gen_cvpo(name, obs_dim, act_dim, hidden, env_num, ep_len, cycles, batch_size, updates_per_cycle, seed, cost_limit, n_step=2):
    """CVPO closed loop (fsrl/trainer/offpolicy.py:96-105): collect -> pre_update_fn -> updates -> post_update_fn
    (actor_old <- actor).  Default networks of cvpo_agent.py: Gaussian actor (bounded mean), SingleCritic pair."""
    from fsrl.policy import CVPO
    from fsrl.utils.net.common import ActorCritic
    from fsrl.utils.net.continuous import SingleCritic
    from ref_shim import ActorProb, Net, _Box
    from torch import nn
    from torch.distributions import Independent, Normal
    seed_all(seed)
    actor = ActorProb(Net((obs_dim, ), hidden_sizes=hidden), (act_dim, ), max_action=1.0, conditioned_sigma=True, unbounded=False)
    critics = [SingleCritic(Net((obs_dim, ), (act_dim, ), hidden_sizes=hidden, concat=True)) for _ in range(2)]
    for m in ActorCritic(actor, critics).modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    logger = CaptureLogger()
    policy = CVPO(actor=actor, critics=critics, actor_optim=torch.optim.Adam(actor.parameters(), lr=5e-4),
                  critic_optim=torch.optim.Adam(nn.ModuleList(critics).parameters(), lr=1e-3), logger=logger,
                  action_space=_Box(-1, 1, (act_dim, )), dist_fn=lambda *l: Independent(Normal(*l), 1),
                  max_episode_steps=ep_len, cost_limit=cost_limit, gamma=0.98, n_step=n_step)
    policy.train()
    env = SyntheticSafetyVectorEnv(env_num=env_num, obs_dim=obs_dim, act_dim=act_dim, episode_len=ep_len, seed=seed + 11)
    buf = VectorReplayBuffer(env_num * ep_len * cycles, env_num)           # never wraps
    flat = lambda mods: torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()]).numpy().copy()  # noqa: E731
    out = {"theta_actor0": flat([actor]), "theta_critics0": flat(critics)}
    seed_all(seed + 7)
    curve, last_rows, duals = [], [], []
    for c in range(cycles):
        st = rollout(policy, env, buf)
        policy.pre_update_fn(stats_train={"cost": st["cost"]})
        n0 = len(logger.rows)
        for _ in range(updates_per_cycle):
            policy.update(batch_size, buf)
        policy.post_update_fn(stats_train={"cost": st["cost"]})
        last, i = {}, len(logger.rows) - 1
        while i >= n0 and not (last and "loss/q_total" in logger.rows[i]):      # the rows of the LAST update
            last = {**logger.rows[i], **last}
            i -= 1
        keys = [k for k in last if not k.endswith("_time")]
        last_rows.append([last[k] for k in keys])
        duals.append(list(policy.estep_dual.detach().numpy()) + [policy.mstep_dual_mu.item(), policy.mstep_dual_std.item()])
        curve.append([st["reward"], st["cost"], st["steps"]])
    out.update(curve=np.array(curve), last_rows=np.array(last_rows), duals=np.array(duals, np.float64),
               stat_keys=np.array(keys), theta_actor_final=flat([actor]), theta_critics_final=flat(critics),
               theta_critics_old_final=flat(list(policy.critics_old)), theta_actor_old_final=flat([policy.actor_old]))
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, ep_len=ep_len, cycles=cycles,
               batch_size=batch_size, updates_per_cycle=updates_per_cycle, seed=seed, cost_limit=cost_limit, n_step=n_step,
               actor_lr=5e-4, critic_lr=1e-3, tau=0.05, gamma=0.98)
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"loop_{name}.npz"), **out)
    print(f"G11 loop_{name}.npz cycles={cycles} reward {curve[0][0]:.2f} -> {curve[-1][0]:.2f} cost {curve[0][1]:.2f} -> {curve[-1][1]:.2f} "
          f"duals {duals[0]} -> {duals[-1]}")


if __name__ == "__main__":
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "cvpo":
        gen_cvpo("cvpo", 8, 2, (64, 64), env_num=6, ep_len=50, cycles=10, batch_size=64, updates_per_cycle=30, seed=76, cost_limit=2.0)
        sys.exit(0)
    gen_sac("sac", 8, 2, (64, 64), env_num=6, ep_len=50, cycles=10, batch_size=64, updates_per_cycle=30, seed=71, cost_limit=5.0)
    gen_focops("focops", 8, 2, (64, 64), env_num=8, ep_len=60, cycles=10, batch_size=128, repeat=4, seed=72, cost_limit=8.0)
    gen_trust("cpo", "cpo", 8, 2, (64, 64), env_num=8, ep_len=60, cycles=8, repeat=2, seed=74, cost_limit=20.0)
    gen_trust("trpo", "trpo", 8, 2, (64, 64), env_num=8, ep_len=60, cycles=8, repeat=2, seed=75, cost_limit=20.0)
    gen_cvpo("cvpo", 8, 2, (64, 64), env_num=6, ep_len=50, cycles=10, batch_size=64, updates_per_cycle=30, seed=76, cost_limit=2.0)
    gen_ddpg("ddpg", 8, 2, (64, 64), env_num=6, ep_len=50, cycles=10, batch_size=64, updates_per_cycle=30, seed=73, cost_limit=5.0)
    gen("ppo", 8, 2, (64, 64), env_num=8, ep_len=60, cycles=12, batch_size=128, repeat=4, seed=70, cost_limit=8.0,
        target_kl=0.5, max_grad_norm=0.5)
