"""CPU oracle of the DDPG-Lagrangian update (torch fp32 + float64 n-step returns).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  A functional restatement of

  BasePolicy.compute_nstep_returns   fsrl/policy/base_policy.py:453-512
  DDPGLagrangian._target_q           fsrl/policy/ddpg_lag.py:125-131   (target actor + target critics)
  DDPGLagrangian.critics_loss        fsrl/policy/ddpg_lag.py:166-187
  DDPGLagrangian.policy_loss         fsrl/policy/ddpg_lag.py:189-213
  DDPGLagrangian.learn / sync_weight fsrl/policy/ddpg_lag.py:215-223, 120-123
  LagrangianPolicy.safety_loss       fsrl/policy/lagrangian_base.py:145-166
  tianshou 0.5 Actor / Critic        (restated: max_action * tanh(MLP(obs)); Q = MLP(concat(obs, act)))

Parameter layout (torch `parameters()` order):
  actor   : W1[H,Do] b1 W2[H,H] b2 W3[Da,H] b3
  critics : for i in (reward, cost): W1[H,Do+Da] b1 W2 b2 W3[1,H] b3
The only randomness of an update is `indices` (buffer.sample), injected by the caller.
Pinned against tests/golden/ddpg_*.npz (recorded from the unmodified reference)."""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .sac_lag import _leaves
from .scans import nstep_return_np


@dataclass
class DDPGConfig:
    obs_dim: int
    act_dim: int
    hidden: Tuple[int, ...] = (128, 128)
    max_action: float = 1.0
    gamma: float = 0.99
    n_step: int = 3
    tau: float = 0.05
    actor_lr: float = 1e-4
    critic_lr: float = 1e-3
    use_lagrangian: bool = True


def mlp_spec(d_in, d_out, hidden):
    """W1 b1 ... W{L+1} b{L+1}: hidden_sizes of any length (ddpg_lag_agent.py takes a tuple), then the output Linear"""
    sizes = [int(d_in)] + [int(h) for h in hidden] + [int(d_out)]
    items = []
    for l in range(1, len(sizes)):
        items += [(f"W{l}", (sizes[l], sizes[l - 1])), (f"b{l}", (sizes[l], ))]
    return OrderedDict(items)


def mlp(p, x):
    n_lin = sum(1 for k in p if k[0] == "W")
    h = x
    for l in range(1, n_lin):
        h = torch.relu(F.linear(h, p[f"W{l}"], p[f"b{l}"]))
    return F.linear(h, p[f"W{n_lin}"], p[f"b{n_lin}"])


class DDPGLagOracle:
    def __init__(self, cfg: DDPGConfig):
        self.cfg = cfg
        self.aspec = mlp_spec(cfg.obs_dim, cfg.act_dim, cfg.hidden)
        self.cspec = mlp_spec(cfg.obs_dim + cfg.act_dim, 1, cfg.hidden)
        self.n_actor = sum(int(np.prod(s)) for s in self.aspec.values())
        self.n_critic = sum(int(np.prod(s)) for s in self.cspec.values())

    def set_params(self, actor_flat, critics_flat):
        a = torch.as_tensor(np.asarray(actor_flat, np.float32))
        c = torch.as_tensor(np.asarray(critics_flat, np.float32))
        self.actor, _ = _leaves(a, self.aspec, 0)
        self.actor_old = OrderedDict((k, v.detach().clone()) for k, v in self.actor.items())
        self.critics, self.critics_old, off = [], [], 0
        for _ in range(2):
            leaves, off = _leaves(c, self.cspec, off)
            self.critics.append(leaves)
            self.critics_old.append(OrderedDict((k, v.detach().clone()) for k, v in leaves.items()))
        self.actor_optim = torch.optim.Adam(list(self.actor.values()), lr=self.cfg.actor_lr)
        self.critic_optim = torch.optim.Adam([p for cr in self.critics for p in cr.values()], lr=self.cfg.critic_lr)

    @staticmethod
    def _flat(nets):
        return torch.cat([p.detach().reshape(-1) for n in nets for p in n.values()]).numpy().copy()

    def actor_flat(self, old=False):
        return self._flat([self.actor_old if old else self.actor])

    def critics_flat(self, old=False):
        return self._flat(self.critics_old if old else self.critics)

    def pi(self, p, obs):
        return self.cfg.max_action * torch.tanh(mlp(p, obs))

    def update(self, store, index, indices, lagrangians, rescaling):
        cfg = self.cfg
        B = len(indices)
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)  # noqa: E731
        chain = [np.asarray(indices)]
        for _ in range(cfg.n_step - 1):
            chain.append(index.next(chain[-1]))
        chain = np.stack(chain)
        terminal = chain[-1]
        value_mask = (~np.asarray(store["terminated"], bool)[terminal]).reshape(-1, 1)
        end_flag = index.done.copy()
        end_flag[index.unfinished_index()] = True
        with torch.no_grad():
            obs_n = t(store["obs_next"][terminal])
            a_n = self.pi(self.actor_old, obs_n)
            targets = [mlp(self.critics_old[i], torch.cat([obs_n, a_n], 1)) for i in range(2)]
        metrics = [np.asarray(store["rew"], np.float64), np.asarray(store["cost"]).astype(np.float64)]
        rets = []
        for i in range(2):
            tq = targets[i].reshape(B, -1).numpy() * value_mask
            rets.append(torch.from_numpy(nstep_return_np(metrics[i], end_flag, tq, chain, cfg.gamma, cfg.n_step)).to(
                torch.float32))
        obs, act = t(store["obs"][indices]), t(store["act"][indices])
        x = torch.cat([obs, act], 1)
        stats_c, loss_c = {}, 0
        for i in range(2):
            td = mlp(self.critics[i], x).flatten() - rets[i].flatten()
            li = (td.pow(2) * 1.0).mean()
            loss_c = loss_c + li
            stats_c["loss/q" + str(i)] = li.item()
        self.critic_optim.zero_grad()
        loss_c.backward()
        self.critic_optim.step()
        stats_c["loss/q_total"] = loss_c.item()
        a_pi = self.pi(self.actor, obs)
        xa = torch.cat([obs, a_pi], 1)
        loss_rew = -mlp(self.critics[0], xa).mean()
        stats_a = {"loss/rescaling": rescaling}
        loss_safety = 0.0
        if cfg.use_lagrangian:
            ls = torch.mean(mlp(self.critics[1], xa).mean() * lagrangians[0])
            loss_safety = loss_safety + ls
            stats_a["loss/lagrangian"] = lagrangians[0]
            stats_a["loss/actor_safety"] = ls.item()
        loss_a = rescaling * (loss_rew + loss_safety)
        self.actor_optim.zero_grad()
        for cr in self.critics:
            for p in cr.values():
                p.grad = None
        loss_a.backward()
        self.actor_optim.step()
        stats_a.update({"loss/actor_rew": loss_rew.item(), "loss/actor_total": loss_a.item()})
        with torch.no_grad():
            for tgt, src in [(self.actor_old, self.actor)] + list(zip(self.critics_old, self.critics)):
                for k in src:
                    tgt[k].copy_(cfg.tau * src[k].data + (1 - cfg.tau) * tgt[k].data)
        return stats_a, stats_c, torch.stack(rets, -1)
