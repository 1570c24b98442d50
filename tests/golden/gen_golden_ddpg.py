#!/usr/bin/env python
"""Golden vectors G9: DDPGLagrangian.update (fsrl/policy/ddpg_lag.py) from the UNMODIFIED reference,
build container only.  Records the replay store, parameters and the sampled indices (the update itself
draws no other random numbers) and the per-update stats / parameters after K updates.

    python tests/golden/gen_golden_ddpg.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from fsrl.policy import DDPGLagrangian  # noqa: E402
from fsrl.utils.net.common import ActorCritic  # noqa: E402
from torch import nn  # noqa: E402

from gen_golden import CaptureLogger, fill_buffer, seed_all  # noqa: E402
from ref_shim import Actor, Critic, Net, _Box  # noqa: E402


def flat(mods):
    return torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()]).numpy().copy()


def gen(name, obs_dim, act_dim, hidden, env_num, ep_lens, batch_size, n_updates, seed, n_step=3, max_action=1.0,
        cost_stat=25.0, cost_limit=10.0, tau=0.05, actor_lr=1e-4, critic_lr=1e-3, gamma=0.99, use_lagrangian=True):
    seed_all(seed)
    actor = Actor(Net((obs_dim, ), hidden_sizes=hidden), (act_dim, ), max_action=max_action)
    actor_optim = torch.optim.Adam(actor.parameters(), lr=actor_lr)
    critics = [Critic(Net((obs_dim, ), (act_dim, ), hidden_sizes=hidden, concat=True)) for _ in range(2)]
    critic_optim = torch.optim.Adam(nn.ModuleList(critics).parameters(), lr=critic_lr)
    ac = ActorCritic(actor, critics)
    for m in ac.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    g = torch.Generator().manual_seed(seed + 3)
    with torch.no_grad():
        for p in ac.parameters():
            if p.ndim == 1:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    logger = CaptureLogger()
    policy = DDPGLagrangian(actor=actor, critics=critics, actor_optim=actor_optim, critic_optim=critic_optim,
                            logger=logger, tau=tau, gamma=gamma, n_step=n_step, exploration_noise=None,
                            use_lagrangian=use_lagrangian, cost_limit=cost_limit,
                            observation_space=_Box(-np.inf, np.inf, (obs_dim, )),
                            action_space=_Box(-max_action, max_action, (act_dim, )))
    policy.train()
    rng = np.random.default_rng(seed + 1000)
    buf = fill_buffer(rng, env_num, ep_lens, obs_dim, act_dim, buffer_size=100000)
    buf._meta["act"][:] = max_action * np.tanh(buf._meta["act"])
    out = {"theta_actor0": flat([actor]), "theta_critics0": flat(critics)}
    meta = buf._meta
    used = np.concatenate([np.arange(o, o + len(b)) for o, b in zip(buf._offset, buf.buffers)])
    out["slots"] = used
    for k in ("obs", "act", "rew", "terminated", "truncated", "obs_next"):
        out["st_" + k] = meta[k][used]
    out["st_cost"] = meta["info.cost"][used]
    out["env_rows"] = np.array([len(b) for b in buf.buffers])
    out["sub_size"] = np.array(buf.buffers[0].maxsize)
    policy.pre_update_fn(stats_train={"cost": cost_stat})
    out["lagrangian"] = np.array([o.get_lag() for o in policy.lag_optims], np.float64)
    idx_log = []
    orig_sample = buf.sample

    def rec_sample(bs):
        b, idx = orig_sample(bs)
        idx_log.append(np.asarray(idx).copy())
        return b, idx

    buf.sample = rec_sample
    seed_all(seed + 7)
    for _ in range(n_updates):
        policy.update(batch_size, buf)
    out["indices"] = np.stack(idx_log)
    rows = logger.rows
    assert len(rows) == 2 * n_updates
    ka, kc = list(rows[0].keys()), list(rows[1].keys())
    out["stats_actor_keys"], out["stats_critic_keys"] = np.array(ka), np.array(kc)
    out["stats_actor"] = np.array([[rows[2 * i][k] for k in ka] for i in range(n_updates)], np.float64)
    out["stats_critic"] = np.array([[rows[2 * i + 1][k] for k in kc] for i in range(n_updates)], np.float64)
    out["theta_actor_final"] = flat([actor])
    out["theta_actor_old_final"] = flat([policy.actor_old])
    out["theta_critics_final"] = flat(critics)
    out["theta_critics_old_final"] = flat(list(policy.critics_old))
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, batch_size=batch_size,
               n_updates=n_updates, seed=seed, n_step=n_step, cost_stat=cost_stat, cost_limit=cost_limit, tau=tau,
               actor_lr=actor_lr, critic_lr=critic_lr, gamma=gamma, max_action=max_action,
               use_lagrangian=use_lagrangian, lagrangian_pid=[0.05, 0.0005, 0.1], buffer_size=100000)
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"ddpg_{name}.npz"), **out)
    print(f"G9 ddpg_{name}.npz rows={len(used)} updates={n_updates} keys={ka} {kc} "
          f"q_total last={out['stats_critic'][-1][kc.index('loss/q_total')]:.4f}")


if __name__ == "__main__":
    torch.set_num_threads(4)
    eps = [[60, 50, -17], [70, 55], [40, 40, 40, -9]]
    if sys.argv[1:] == ["depths"]:
        # hidden_sizes the fused kernels do not hold (ddpg_lag_agent.py: any tuple): layered contexts on the HIP side
        gen("deep3", 6, 3, (40, 56, 32), 3, eps, batch_size=64, n_updates=5, seed=43, n_step=2)
        sys.exit(0)
    gen("small", 6, 3, (64, 64), 3, eps, batch_size=64, n_updates=6, seed=40, n_step=3)
    gen("scaled", 8, 2, (128, 128), 3, eps, batch_size=100, n_updates=5, seed=41, n_step=1, max_action=2.0)
    gen("nolag", 17, 4, (256, 256), 2, [[90, -30], [100]], batch_size=256, n_updates=4, seed=42, n_step=2,
        use_lagrangian=False)
