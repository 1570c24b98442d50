"""CPU oracle of the SAC-Lagrangian update (torch fp32 + float64 n-step returns).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  A functional restatement of

  BasePolicy.compute_nstep_returns   fsrl/policy/base_policy.py:453-512
  SACLagrangian._target_q            fsrl/policy/sac_lag.py:136-145
  SACLagrangian.forward              fsrl/policy/sac_lag.py:155-183   (tanh-Gaussian, eps = finfo(f32).eps)
  SACLagrangian.critics_loss         fsrl/policy/sac_lag.py:185-210
  SACLagrangian.policy_loss          fsrl/policy/sac_lag.py:212-258
  SACLagrangian.learn / sync_weight  fsrl/policy/sac_lag.py:260-269, 132-134
  DoubleCritic                       fsrl/utils/net/continuous.py:13-101
  tianshou 0.5 ReplayBuffer.next / unfinished_index   (restated)

Parameter layout (torch `parameters()` order):
  actor   : W1[H,Do] b1 W2[H,H] b2 Wmu[Da,H] bmu Wsig[Da,H] bsig        (ActorProb, conditioned sigma)
  critics : for i in (reward, cost): pre1(W1[H,Do+Da] b1 W2 b2) pre2(...) last1(W[1,H] b) last2(W b)
Randomness is injected: `indices` (buffer.sample), `eps_target` / `eps_pi` (rsample's N(0,1)).
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .scans import nstep_return_np

SIGMA_MIN, SIGMA_MAX = -20.0, 2.0
F32_EPS = float(np.finfo(np.float32).eps)


@dataclass
class SACConfig:
    obs_dim: int
    act_dim: int
    hidden: Tuple[int, ...] = (128, 128)
    gamma: float = 0.99
    n_step: int = 2
    tau: float = 0.05
    alpha: float = 0.005
    auto_alpha: bool = True
    target_entropy: float = None
    actor_lr: float = 5e-4
    critic_lr: float = 1e-3
    alpha_lr: float = 3e-4
    use_lagrangian: bool = True


def _hidden_items(d_in, hidden, suffix=""):
    """W1 b1 ... WL bL of the preprocess MLP (tianshou Net, hidden_sizes of any length: sac_lag_agent.py takes a tuple)"""
    sizes = [int(d_in)] + [int(h) for h in hidden]
    items = []
    for l in range(1, len(sizes)):
        items += [(f"W{l}{suffix}", (sizes[l], sizes[l - 1])), (f"b{l}{suffix}", (sizes[l], ))]
    return items


def actor_spec(Do, Da, hidden):
    hl = int(hidden[-1])
    return OrderedDict(_hidden_items(Do, hidden) + [("Wmu", (Da, hl)), ("bmu", (Da, )), ("Wsig", (Da, hl)), ("bsig", (Da, ))])


def double_critic_spec(Do, Da, hidden):
    L, hl = len(hidden), int(hidden[-1])
    s = OrderedDict()
    for j in (1, 2):
        s.update(_hidden_items(Do + Da, hidden, f"_{j}"))
    for j in (1, 2):
        s.update({f"W{L + 1}_{j}": (1, hl), f"b{L + 1}_{j}": (1, )})
    return s


def _trunk(p, x, n_hidden, suffix=""):
    h = x
    for l in range(1, n_hidden + 1):
        h = torch.relu(F.linear(h, p[f"W{l}{suffix}"], p[f"b{l}{suffix}"]))
    return h


def _leaves(flat, spec, off):
    out = OrderedDict()
    for k, shape in spec.items():
        n = int(np.prod(shape))
        out[k] = flat[off:off + n].reshape(shape).clone().requires_grad_(True)
        off += n
    return out, off


class ReplayIndex:
    """Index semantics of tianshou-0.5 VectorReplayBuffer for rows stored without wrap-around:
    env e owns slots [e*sub, e*sub + rows[e])."""

    def __init__(self, env_rows, sub_size, done):
        self.rows, self.sub = np.asarray(env_rows), int(sub_size)
        self.done = np.asarray(done, bool)            # indexed by SLOT (size = n_env*sub)
        self.last = np.array([e * self.sub + r - 1 for e, r in enumerate(self.rows)])

    def next(self, idx):
        idx = np.asarray(idx)
        stop = self.done[idx] | np.isin(idx, self.last)
        return idx + (~stop).astype(idx.dtype)

    def unfinished_index(self):
        return np.array([l for l, r in zip(self.last, self.rows) if r > 0 and not self.done[l]], int)


class SACLagOracle:
    def __init__(self, cfg: SACConfig):
        self.cfg = cfg
        self.aspec = actor_spec(cfg.obs_dim, cfg.act_dim, cfg.hidden)
        self.cspec = double_critic_spec(cfg.obs_dim, cfg.act_dim, cfg.hidden)
        self.n_actor = sum(int(np.prod(s)) for s in self.aspec.values())
        self.n_critic = sum(int(np.prod(s)) for s in self.cspec.values())
        self.target_entropy = cfg.target_entropy if cfg.target_entropy is not None else -float(cfg.act_dim)

    def set_params(self, actor_flat, critics_flat, log_alpha=0.0):
        a = torch.as_tensor(np.asarray(actor_flat, np.float32))
        c = torch.as_tensor(np.asarray(critics_flat, np.float32))
        self.actor, _ = _leaves(a, self.aspec, 0)
        self.critics, self.critics_old, off = [], [], 0
        for _ in range(2):
            leaves, off2 = _leaves(c, self.cspec, off)
            self.critics.append(leaves)
            self.critics_old.append(OrderedDict((k, v.detach().clone()) for k, v in leaves.items()))
            off = off2
        self.actor_optim = torch.optim.Adam(list(self.actor.values()), lr=self.cfg.actor_lr)
        self.critic_optim = torch.optim.Adam([p for cr in self.critics for p in cr.values()],
                                             lr=self.cfg.critic_lr)
        self.log_alpha = torch.full((1, ), float(log_alpha), requires_grad=True)
        self.alpha_optim = torch.optim.Adam([self.log_alpha], lr=self.cfg.alpha_lr)
        self.alpha = self.log_alpha.detach().exp() if self.cfg.auto_alpha else self.cfg.alpha

    def actor_flat(self):
        return torch.cat([p.detach().reshape(-1) for p in self.actor.values()]).numpy().copy()

    def critics_flat(self, old=False):
        src = self.critics_old if old else self.critics
        return torch.cat([p.detach().reshape(-1) for cr in src for p in cr.values()]).numpy().copy()

    # ------------------------------------------------------------------ nets
    def pi(self, obs, eps):
        p = self.actor
        h = _trunk(p, obs, len(self.cfg.hidden))
        mu = F.linear(h, p["Wmu"], p["bmu"])
        sigma = torch.clamp(F.linear(h, p["Wsig"], p["bsig"]), min=SIGMA_MIN, max=SIGMA_MAX).exp()
        u = mu + eps * sigma                                          # Normal.rsample
        var = sigma**2
        logn = (-((u - mu)**2) / (2 * var) - sigma.log() - np.log(np.sqrt(2 * np.pi))).sum(-1, keepdim=True)
        a = torch.tanh(u)
        logp = logn - torch.log((1 - a.pow(2)) + F32_EPS).sum(-1, keepdim=True)
        return a, logp

    @staticmethod
    def q_pair(cr, obs, act):
        x = torch.cat([obs, act], dim=1)
        out = []
        for j in (1, 2):
            L = sum(1 for k in cr if k.startswith("W") and k.endswith("_1")) - 1      # hidden layers
            h = _trunk(cr, x, L, f"_{j}")
            out.append(F.linear(h, cr[f"W{L + 1}_{j}"], cr[f"b{L + 1}_{j}"]))
        return out

    # ------------------------------------------------------------------ update
    def update(self, store, index: ReplayIndex, indices, eps_target, eps_pi, lagrangians, rescaling):
        """store: dict of SLOT-indexed arrays obs, act, rew, cost, terminated, obs_next."""
        cfg = self.cfg
        B = len(indices)
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)  # noqa: E731
        # ---- n-step index chain and returns (float64)
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
            a_n, logp_n = self.pi(obs_n, t(eps_target))
            targets = []
            for i in range(2):
                q1, q2 = self.q_pair(self.critics_old[i], obs_n, a_n)
                targets.append(torch.min(q1, q2) - self.alpha * logp_n)
        metrics = [np.asarray(store["rew"], np.float64), np.asarray(store["cost"]).astype(np.float64)]
        rets = []
        for i in range(2):
            tq = targets[i].reshape(B, -1).numpy() * value_mask
            rets.append(torch.from_numpy(nstep_return_np(metrics[i], end_flag, tq, chain, cfg.gamma,
                                                         cfg.n_step)).to(torch.float32))
        rets = torch.stack(rets, -1)                                   # [B, 1, 2]
        obs, act = t(store["obs"][indices]), t(store["act"][indices])
        # ---- critics
        stats_c, loss_c = {}, 0
        for i in range(2):
            y = rets[..., i].flatten()
            q = self.q_pair(self.critics[i], obs, act)
            li = 0
            for j in range(2):
                td = q[j].flatten() - y
                li = li + (td.pow(2) * 1.0).mean()
            loss_c = loss_c + li
            stats_c["loss/q" + str(i)] = li.item()
        self.critic_optim.zero_grad()
        loss_c.backward()
        self.critic_optim.step()
        stats_c["loss/q_total"] = loss_c.item()
        # ---- actor
        a_pi, logp = self.pi(obs, t(eps_pi))
        qr = self.q_pair(self.critics[0], obs, a_pi)
        cur_q = torch.min(qr[0], qr[1]).flatten()
        loss_rew = (self.alpha * logp.flatten() - cur_q).mean()
        stats_a = {"loss/rescaling": rescaling}
        loss_safety = 0.0
        if cfg.use_lagrangian:
            qc = self.q_pair(self.critics[1], obs, a_pi)
            safety_q = torch.min(qc[0], qc[1]).flatten()
            ls = torch.mean(safety_q * lagrangians[0])
            loss_safety = loss_safety + ls
            stats_a["loss/lagrangian"] = lagrangians[0]
            stats_a["loss/actor_safety"] = ls.item()
        loss_a = rescaling * (loss_rew + loss_safety)
        self.actor_optim.zero_grad()
        for cr in self.critics:                       # the reference lets these grads accumulate;
            for p in cr.values():                     # critic_optim.zero_grad() clears them next time
                p.grad = None
        loss_a.backward()
        self.actor_optim.step()
        if cfg.auto_alpha:
            lp = logp.detach() + self.target_entropy
            alpha_loss = -(self.log_alpha * lp).mean()
            self.alpha_optim.zero_grad()
            alpha_loss.backward()
            self.alpha_optim.step()
            self.alpha = self.log_alpha.detach().exp()
            stats_a.update({"loss/alpha_loss": alpha_loss.item(), "loss/alpha_value": self.alpha.item()})
        stats_a.update({"loss/actor_rew": loss_rew.item(), "loss/actor_total": loss_a.item()})
        # ---- Polyak
        with torch.no_grad():
            for i in range(2):
                for k in self.critics[i]:
                    tgt, src = self.critics_old[i][k], self.critics[i][k]
                    tgt.copy_(cfg.tau * src.data + (1 - cfg.tau) * tgt.data)
        return stats_a, stats_c, rets
