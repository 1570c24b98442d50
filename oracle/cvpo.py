"""CPU oracle of the CVPO update (torch fp32 + float64 n-step returns).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  A functional restatement of

  BasePolicy.compute_nstep_returns   fsrl/policy/base_policy.py:453-512
  CVPO._target_q / process_fn        fsrl/policy/cvpo.py:206-222
  CVPO.forward                       fsrl/policy/cvpo.py:224-246    (Independent(Normal(mu, sigma)).sample())
  CVPO.critics_loss                  fsrl/policy/cvpo.py:248-276
  CVPO._estep_dual_loss              fsrl/policy/cvpo.py:278-288
  CVPO.gaussian_kl                   fsrl/policy/cvpo.py:290-317
  CVPO.policy_loss (E-step, M-step)  fsrl/policy/cvpo.py:319-420
  CVPO.learn / sync_weight           fsrl/policy/cvpo.py:422-430, 202-204
  CVPO.pre_update_fn/post_update_fn  fsrl/policy/cvpo.py:178-193
  SingleCritic / DoubleCritic        fsrl/utils/net/continuous.py:103-160, 13-101
  ActorProb (unbounded=False)        tianshou 0.5 utils/net/continuous.py: mu = max_action * tanh(head),
                                     sigma = exp(clamp(head, -20, 2))

Two behaviours of the reference that look accidental are restated as they are, because parity is the bar:
  * `_estep_dual_loss` builds `combined_q` as `q_values[0].detach()`, which SHARES storage with q_values[0],
    and then subtracts lambda*Qc from it in place.  Every E-step iteration therefore leaves q_values[0]
    lowered by (pre-step lambda)*Qc, and the softmax weights subtract (post-step lambda)*Qc once more.
  * the M-step duals are used CLIPPED to [0, mstep_dual_max] but stored unclipped (they do go negative).

Parameter layout (torch `parameters()` order):
  actor   : W1[H,Do] b1 W2[H,H] b2 Wmu[Da,H] bmu Wsig[Da,H] bsig
  critics : for i in (reward, cost): SingleCritic: W1[H,Do+Da] b1 W2 b2 W3[1,H] b3
                                     DoubleCritic: as oracle/sac_lag.py
Randomness is injected: `indices` (buffer.sample), `eps_target` [B,Da], `eps_particles` [K,B,Da].
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .sac_lag import ReplayIndex, _leaves, actor_spec, double_critic_spec  # noqa: F401
from .scans import nstep_return_np

SIGMA_MIN, SIGMA_MAX = -20.0, 2.0
EPS10 = float(np.finfo(np.float32).eps.item() * 10)
LOG_SQRT_2PI = float(np.log(np.sqrt(2 * np.pi)))


@dataclass
class CVPOConfig:
    obs_dim: int
    act_dim: int
    hidden: Tuple[int, ...] = (128, 128)
    max_action: float = 1.0
    gamma: float = 0.98
    n_step: int = 2
    tau: float = 0.05
    actor_lr: float = 5e-4
    critic_lr: float = 1e-3
    double_critic: bool = False
    sample_act_num: int = 16
    estep_iter_num: int = 1
    estep_kl: float = 0.02
    estep_dual_max: float = 20.0
    estep_dual_lr: float = 0.02
    mstep_iter_num: int = 1
    mstep_kl_mu: float = 0.005
    mstep_kl_std: float = 0.0005
    mstep_dual_max: float = 0.5
    mstep_dual_lr: float = 0.1
    cost_limit: float = 10.0
    max_episode_steps: int = 100

    @property
    def qc_thres(self):
        g, T = self.gamma, self.max_episode_steps
        return self.cost_limit * (1 - g**T) / (1 - g) / T          # cvpo.py:138-141


def single_critic_spec(Do, Da, hidden):
    """SingleCritic: the preprocess MLP (hidden_sizes of any length) then the last Linear"""
    from .sac_lag import _hidden_items
    L = len(hidden)
    return OrderedDict(_hidden_items(Do + Da, hidden, "_1") + [(f"W{L + 1}_1", (1, int(hidden[-1]))), (f"b{L + 1}_1", (1, ))])


class CVPOOracle:
    def __init__(self, cfg: CVPOConfig):
        self.cfg = cfg
        self.aspec = actor_spec(cfg.obs_dim, cfg.act_dim, cfg.hidden)
        self.cspec = (double_critic_spec if cfg.double_critic else single_critic_spec)(cfg.obs_dim, cfg.act_dim,
                                                                                       cfg.hidden)
        self.n_q = 2 if cfg.double_critic else 1

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
        self.estep_dual = torch.tensor([1.0, 0.0], requires_grad=True)            # cvpo.py:150-155
        self.estep_optim = torch.optim.Adam([self.estep_dual], lr=self.cfg.estep_dual_lr)
        self.pre_update()

    def pre_update(self):
        """cvpo.py:178-188: fresh M-step duals and optimiser at every collect cycle."""
        self.mstep_dual_mu = torch.zeros(1, requires_grad=True)
        self.mstep_dual_std = torch.zeros(1, requires_grad=True)
        self.mstep_optim = torch.optim.Adam([self.mstep_dual_mu, self.mstep_dual_std], lr=self.cfg.mstep_dual_lr)

    def post_update(self):
        """cvpo.py:190-193: actor_old <- actor."""
        with torch.no_grad():
            for k in self.actor:
                self.actor_old[k].copy_(self.actor[k])

    def actor_flat(self, old=False):
        src = self.actor_old if old else self.actor
        return torch.cat([p.detach().reshape(-1) for p in src.values()]).numpy().copy()

    def critics_flat(self, old=False):
        src = self.critics_old if old else self.critics
        return torch.cat([p.detach().reshape(-1) for cr in src for p in cr.values()]).numpy().copy()

    # ------------------------------------------------------------------ nets
    def pi(self, p, obs):
        from .sac_lag import _trunk
        h = _trunk(p, obs, len(self.cfg.hidden))
        mu = self.cfg.max_action * torch.tanh(F.linear(h, p["Wmu"], p["bmu"]))
        sigma = torch.clamp(F.linear(h, p["Wsig"], p["bsig"]), min=SIGMA_MIN, max=SIGMA_MAX).exp()
        return mu, sigma

    def q_list(self, cr, obs, act):
        x = torch.cat([obs, act], dim=1)
        out = []
        for j in range(1, self.n_q + 1):
            from .sac_lag import _trunk
            L = len(self.cfg.hidden)
            h = _trunk(cr, x, L, f"_{j}")
            out.append(F.linear(h, cr[f"W{L + 1}_{j}"], cr[f"b{L + 1}_{j}"]))
        return out

    def q_predict(self, cr, obs, act):
        q = self.q_list(cr, obs, act)
        return q[0] if self.n_q == 1 else torch.min(q[0], q[1])

    @staticmethod
    def _logp(a, mu, sigma):
        """Independent(Normal(mu, sigma), 1).log_prob(a)"""
        return (-((a - mu)**2) / (2 * sigma**2) - sigma.log() - LOG_SQRT_2PI).sum(-1)

    # ------------------------------------------------------------------ update
    def update(self, store, index: ReplayIndex, indices, eps_target, eps_particles):
        """One CVPO.update: process_fn + learn.  store: dict of SLOT-indexed arrays."""
        cfg = self.cfg
        B, K = len(indices), cfg.sample_act_num
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)  # noqa: E731
        # ---- n-step returns (float64)
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
            mu_n, sig_n = self.pi(self.actor, obs_n)
            a_n = t(eps_target) * sig_n + mu_n
            targets = [self.q_predict(self.critics_old[i], obs_n, a_n) for i in range(2)]
        metrics = [np.asarray(store["rew"], np.float64), np.asarray(store["cost"]).astype(np.float64)]
        rets = []
        for i in range(2):
            tq = targets[i].reshape(B, -1).numpy() * value_mask
            rets.append(torch.from_numpy(nstep_return_np(metrics[i], end_flag, tq, chain, cfg.gamma,
                                                         cfg.n_step)).to(torch.float32))
        rets = torch.stack(rets, -1)
        obs, act = t(store["obs"][indices]), t(store["act"][indices])
        stats = {}
        # ---- critics (cvpo.py:248-276)
        loss_c = 0
        for i in range(2):
            y = rets[..., i].flatten()
            li = 0
            for q in self.q_list(self.critics[i], obs, act):
                li = li + (q.flatten() - y).pow(2).mean()
            loss_c = loss_c + li
            stats["loss/loss_q" + str(i)] = li.item()
            stats["estep/val_q" + str(i)] = y.mean().item()
        stats["estep/thres_q1"] = cfg.qc_thres
        self.critic_optim.zero_grad()
        loss_c.backward()
        self.critic_optim.step()
        stats["loss/q_total"] = loss_c.item()
        # ---- E-step (cvpo.py:319-371)
        with torch.no_grad():
            mu_old, std_old = self.pi(self.actor_old, obs)
            particles = t(eps_particles) * std_old.expand(K, B, -1) + mu_old.expand(K, B, -1)      # [K,B,Da]
            obs_k = obs[None].expand(K, -1, -1).reshape(K * B, -1)
            q = [self.q_predict(self.critics[i], obs_k, particles.reshape(K * B, -1)).reshape(K, B).T.clone()
                 for i in range(2)]                                                               # [B,K] each
        for it in range(cfg.estep_iter_num):
            self.estep_optim.zero_grad()
            eta = self.estep_dual[0]
            combined = q[0] - self.estep_dual[1] * q[1]
            loss = eta * cfg.estep_kl + self.estep_dual[1] * cfg.qc_thres
            loss = loss + eta * torch.mean(torch.logsumexp(combined / eta, dim=1) - np.log(K))
            loss.backward()
            self.estep_optim.step()
            q[0] = combined.detach()                       # the in-place aliasing of cvpo.py:282-285
            if it == 0:
                stats["loss/estep_loss"] = loss.item()
        self.estep_dual.data.clamp_(min=EPS10, max=cfg.estep_dual_max)
        d0, d1 = self.estep_dual[0].item(), self.estep_dual[1].item()
        stats["estep/dual0"], stats["estep/dual1"] = d0, d1
        w = torch.softmax((q[0].T - d1 * q[1].T) / d0, dim=0)                                    # [K,B]
        # ---- M-step (cvpo.py:378-417)
        for it in range(cfg.mstep_iter_num):
            mu, std = self.pi(self.actor, obs)
            ll = self._logp(particles, mu.expand(K, B, -1), std_old.expand(K, B, -1)) + \
                self._logp(particles, mu_old.expand(K, B, -1), std.expand(K, B, -1))
            loss_mle = -torch.mean(w * ll)
            var_old, var = torch.clamp_min(std_old**2, 1e-6), torch.clamp_min(std**2, 1e-6)
            kl_mu = (0.5 * (mu_old - mu)**2 / var_old).sum(-1).mean()
            kl_std = (0.5 * (torch.log(var / var_old) + var_old / var - 1)).sum(-1).mean()
            dual_loss = self.mstep_dual_mu * (cfg.mstep_kl_mu - kl_mu).detach() + \
                self.mstep_dual_std * (cfg.mstep_kl_std - kl_std).detach()
            self.mstep_optim.zero_grad()
            dual_loss.backward()
            self.mstep_optim.step()
            dual_mu = float(np.clip(self.mstep_dual_mu.item(), 0.0, cfg.mstep_dual_max))
            dual_std = float(np.clip(self.mstep_dual_std.item(), 0.0, cfg.mstep_dual_max))
            loss_kl = dual_mu * (kl_mu - cfg.mstep_kl_mu) + dual_std * (kl_std - cfg.mstep_kl_std)
            loss_actor = loss_mle + loss_kl
            self.actor_optim.zero_grad()
            loss_actor.backward()
            self.actor_optim.step()
            if it == 0:
                ent = (0.5 + 0.5 * np.log(2 * np.pi) + torch.log(std_old)).sum(-1) + \
                    (0.5 + 0.5 * np.log(2 * np.pi) + torch.log(std)).sum(-1)
                stats.update({"mstep/mstep_kl_mu": kl_mu.item(), "mstep/mstep_kl_std": kl_std.item(),
                              "mstep/mstep_loss_kl": loss_kl.item(), "mstep/mstep_loss_mle": loss_mle.item(),
                              "mstep/mstep_loss_total": loss_actor.item(), "mstep/mstep_dual_mu": dual_mu,
                              "mstep/mstep_dual_std": dual_std, "mstep/entropy": ent.mean().item()})
        # ---- Polyak (cvpo.py:202-204)
        with torch.no_grad():
            for i in range(2):
                for k in self.critics[i]:
                    tgt, src = self.critics_old[i][k], self.critics[i][k]
                    tgt.copy_(cfg.tau * src.data + (1 - cfg.tau) * tgt.data)
        return stats, rets, w
