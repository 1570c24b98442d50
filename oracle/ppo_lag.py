"""CPU oracle of the PPO-Lagrangian policy update (torch fp32 + float64 GAE).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  A functional restatement of

  BasePolicy.update                 fsrl/policy/base_policy.py:332-355
  BasePolicy.compute_gae_returns    fsrl/policy/base_policy.py:384-451
  PPOLagrangian.process_fn          fsrl/policy/ppo_lag.py:134-150
  PPOLagrangian.policy_loss         fsrl/policy/ppo_lag.py:173-212
  PPOLagrangian.critics_loss        fsrl/policy/ppo_lag.py:152-171
  PPOLagrangian.learn               fsrl/policy/ppo_lag.py:214-257
  LagrangianPolicy.safety_loss      fsrl/policy/lagrangian_base.py:145-166
  tianshou 0.5 Batch.split          (restated)

over a flat parameter vector (oracle/layout.py).  Network math goes through torch CPU
fp32 (autograd, Adam, clip_grad_norm_, Normal) -- the same third-party arithmetic the
reference itself calls -- so on one machine the oracle tracks the reference to rounding.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch.distributions import Independent, Normal

from . import layout
from .scans import gae_return_c

STAT_KEYS = ("loss/rescaling", "loss/lagrangian", "loss/actor_safety", "loss/actor_rew",
             "loss/actor_total", "loss/kl", "loss/vf0", "loss/vf1", "loss/vf_total",
             "loss/total", "loss/entropy")


@dataclass
class PPOLagConfig:
    obs_dim: int
    act_dim: int
    hidden: Tuple[int, ...] = (128, 128)     # hidden_sizes of the agents: any depth (fsrl/agent/ppo_lag_agent.py:91)
    n_critics: int = 2
    max_action: float = 1.0
    gamma: float = 0.99
    gae_lambda: float = 0.95
    eps_clip: float = 0.2
    dual_clip: Optional[float] = None
    vf_coef: float = 0.25
    max_grad_norm: Optional[float] = None
    target_kl: float = 0.02
    advantage_normalization: bool = True
    use_lagrangian: bool = True
    lr: float = 5e-4
    betas: Tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8
    recompute_advantage: bool = False
    unbounded: bool = False              # ActorProb(unbounded=True): no max_action * tanh on the mean head
    reward_normalization: bool = False   # base_policy.py:114, 430-444
    value_clip: bool = False             # ppo_lag.py:158-164 (the reference asserts reward_normalization with it)


@dataclass
class OnPolicyData:
    """Transitions in `buffer.sample(0)` order (env-major, chronological per env)."""
    obs: np.ndarray  # [N, Do] f32
    act: np.ndarray  # [N, Da] f32
    rew: np.ndarray  # [N] f64
    cost: np.ndarray  # [N] f64
    terminated: np.ndarray  # [N] bool
    truncated: np.ndarray  # [N] bool
    obs_next: np.ndarray  # [N, Do] f32
    end_flag: np.ndarray = field(default=None)  # done | unfinished tail (base_policy.py:409-411)

    def __len__(self):
        return len(self.rew)


def split_chunks(n, size, perm=None, merge_last=True):
    """Index chunks of tianshou-0.5 `Batch.split(size, shuffle, merge_last)`; `perm` is the
    np.random.permutation(n) the reference would draw (None => arange, shuffle=False)."""
    idx = np.arange(n) if perm is None else np.asarray(perm)
    merge_last = merge_last and n % size > 0
    out = []
    for i in range(0, n, size):
        if merge_last and i + 2 * size >= n:
            out.append(idx[i:])
            break
        out.append(idx[i:i + size])
    return out


class PPOLagOracle:
    def __init__(self, cfg: PPOLagConfig, dtype=torch.float32):
        """dtype=torch.float64 gives an "exact arithmetic" yardstick: the tests bound the HIP
        path's distance to it by the fp32 oracle's own distance (same algorithm, same inputs)."""
        self.cfg = cfg
        self.dtype = dtype
        self.specs = layout.onpolicy_specs(cfg.obs_dim, cfg.act_dim, cfg.hidden, cfg.n_critics)
        self.n_params = sum(layout.spec_size(s) for s in self.specs)
        self.nets: List[dict] = []
        self._leaves: List[torch.Tensor] = []
        self.set_params(np.zeros(self.n_params, np.float32))
        self.gradient_steps = 0
        # BasePolicy.ret_rms (base_policy.py:111): tianshou 0.5 RunningMeanStd() per critic -> rows (mean, var, count)
        self.ret_rms = np.array([[0.0, 1.0, 0.0]] * cfg.n_critics, np.float64)

    def _rms_update(self, i, x):
        """tianshou 0.5 utils/statistics.py:89-103 RunningMeanStd.update"""
        mean, var, count = self.ret_rms[i]
        b_mean, b_var, b_count = np.mean(x), np.var(x), len(x)
        delta = b_mean - mean
        tot = count + b_count
        new_mean = mean + delta * b_count / tot
        m_2 = var * count + b_var * b_count + delta ** 2 * count * b_count / tot
        self.ret_rms[i] = (new_mean, m_2 / tot, tot)

    # ------------------------------------------------------------------ params
    def set_params(self, flat):
        flat = torch.as_tensor(np.asarray(flat, np.float32)).clone().to(self.dtype)
        assert flat.numel() == self.n_params
        self.nets, self._leaves, off = [], [], 0
        for spec in self.specs:
            v, off = layout.views(flat, spec, off)
            net = {k: t.clone().requires_grad_(True) for k, t in v.items()}
            self.nets.append(net)
            self._leaves += list(net.values())
        self.optim = torch.optim.Adam(self._leaves, lr=self.cfg.lr, betas=self.cfg.betas,
                                      eps=self.cfg.adam_eps)

    def get_params(self):
        return torch.cat([t.detach().reshape(-1) for t in self._leaves]).to(torch.float32).numpy().copy()

    # ------------------------------------------------------------------ nets
    @staticmethod
    def _trunk(p, x):
        n_lin = sum(1 for k in p if k[0] == "W")        # hidden layers + the head
        h = x
        for l in range(1, n_lin):
            h = torch.relu(F.linear(h, p[f"W{l}"], p[f"b{l}"]))
        return F.linear(h, p[f"W{n_lin}"], p[f"b{n_lin}"])

    def actor_dist(self, obs):
        p = self.nets[0]
        mu = self._trunk(p, obs)
        if not self.cfg.unbounded:
            mu = self.cfg.max_action * torch.tanh(mu)
        sigma = (p["sigma_param"].view(1, -1) + torch.zeros_like(mu)).exp()
        return Independent(Normal(mu, sigma), 1)

    def value(self, i, obs):
        return self._trunk(self.nets[1 + i], obs).flatten()

    # ------------------------------------------------------------------ process_fn
    def process(self, data: OnPolicyData):
        """GAE for every critic + old log-prob.  Returns dict of torch f32 tensors."""
        cfg = self.cfg
        dt = self.dtype
        obs = torch.as_tensor(data.obs, dtype=dt)
        obs_next = torch.as_tensor(data.obs_next, dtype=dt)
        metrics = [np.asarray(data.rew, np.float64), np.asarray(data.cost).astype(np.float64)]
        value_mask = ~np.asarray(data.terminated, bool)
        values, rets, advs = [], [], []
        with torch.no_grad():
            for i in range(cfg.n_critics):
                v = self.value(i, obs)
                vn = self.value(i, obs_next).numpy() * value_mask  # f32 * bool -> f32
                if cfg.reward_normalization:       # un-normalise v_s, v_s_ (float32 array * np.float64 -> float64)
                    from .scans import gae_return_np
                    scale = np.sqrt(self.ret_rms[i][1] + 1e-8)
                    vs = v.numpy().astype(np.float64) * scale
                    adv = gae_return_np(vs, vn.astype(np.float64) * scale, metrics[i], data.end_flag, cfg.gamma,
                                        cfg.gae_lambda)
                    ret = (adv + vs) / scale
                    self._rms_update(i, ret)
                    values.append(v)
                    rets.append(torch.from_numpy(ret).to(dt))
                    advs.append(torch.from_numpy(adv).to(dt))
                    continue
                if dt == torch.float32:
                    adv = gae_return_c(v.numpy(), vn, metrics[i], data.end_flag, cfg.gamma,
                                       cfg.gae_lambda)
                else:  # yardstick mode: keep the critic values in float64
                    from .scans import gae_return_np
                    adv = gae_return_np(v.numpy(), vn, metrics[i], data.end_flag, cfg.gamma,
                                        cfg.gae_lambda)
                ret = adv + v.numpy()  # f64 + f32 -> f64
                values.append(v)
                rets.append(torch.from_numpy(ret).to(dt))
                advs.append(torch.from_numpy(adv).to(dt))
            act = torch.as_tensor(data.act, dtype=dt)
            logp_old = self.actor_dist(obs).log_prob(act)
        return dict(obs=obs, act=act, values=torch.stack(values, -1), rets=torch.stack(rets, -1),
                    advs=torch.stack(advs, -1), logp_old=logp_old)

    # ------------------------------------------------------------------ learn
    def _minibatch_losses(self, pb, idx, lagrangians, rescaling):
        cfg = self.cfg
        idx_t = torch.as_tensor(idx, dtype=torch.long)
        obs, act = pb["obs"][idx_t], pb["act"][idx_t]
        advs = pb["advs"][idx_t].clone()  # fancy index = copy; normalised per minibatch copy
        rets, logp_old = pb["rets"][idx_t], pb["logp_old"][idx_t]
        vold = pb["values"][idx_t]
        dist = self.actor_dist(obs)
        logp = dist.log_prob(act)
        ratio = (logp - logp_old).exp().to(self.dtype)
        ratio = ratio.reshape(ratio.size(0), -1).transpose(0, 1)  # (1, B) quirk, ppo_lag.py:177
        if cfg.advantage_normalization:
            for i in range(cfg.n_critics):
                a = advs[..., i]
                advs[..., i] = (a - a.mean()) / a.std()
        a_r = advs[..., 0]
        s1 = ratio * a_r
        s2 = ratio.clamp(1.0 - cfg.eps_clip, 1.0 + cfg.eps_clip) * a_r
        if cfg.dual_clip:
            c1 = torch.min(s1, s2)
            c2 = torch.max(c1, cfg.dual_clip * a_r)
            loss_rew = -torch.where(a_r < 0, c2, c1).mean()
        else:
            loss_rew = -torch.min(s1, s2).mean()
        loss_safety = 0.0
        stats = {"loss/rescaling": rescaling}
        if cfg.use_lagrangian:
            for i in range(1, cfg.n_critics):
                lam = lagrangians[i - 1]
                li = torch.mean(ratio * advs[..., i] * lam)
                loss_safety = loss_safety + li
                suffix = "" if i == 1 else "_" + str(i - 1)
                stats["loss/lagrangian" + suffix] = lam
                stats["loss/actor_safety" + suffix] = li.item()
        loss_actor = rescaling * (loss_rew + loss_safety)
        stats["loss/actor_rew"] = loss_rew.item()
        stats["loss/actor_total"] = loss_actor.item()
        stats["loss/kl"] = (logp_old - logp).mean().item()
        loss_vf = 0
        for i in range(cfg.n_critics):
            v = self.value(i, obs)
            if cfg.value_clip:
                v_clip = vold[..., i] + (v - vold[..., i]).clamp(-cfg.eps_clip, cfg.eps_clip)
                vf = torch.max((rets[..., i] - v).pow(2), (rets[..., i] - v_clip).pow(2)).mean()
            else:
                vf = (rets[..., i] - v).pow(2).mean()
            loss_vf = loss_vf + vf
            stats["loss/vf" + str(i)] = vf.item()
        stats["loss/vf_total"] = loss_vf.item()
        loss = loss_actor + cfg.vf_coef * loss_vf
        return loss, dist, stats

    def learn(self, pb, lagrangians, rescaling, batch_size, repeat, perms=None, data=None):
        """Minibatch SGD passes.  `perms[k]` = permutation of pass k (None: draw from the
        numpy global RNG like Batch.split).  Returns (stats [steps, 11] f64, early_stop_pass)."""
        cfg = self.cfg
        n = len(pb["logp_old"])
        rows = []
        stopped_at = -1
        for k in range(repeat):
            if cfg.recompute_advantage and k > 0:      # ppo_lag.py:218-221: values / rets / advs from the current critics
                fresh = self.process(data)
                pb = dict(pb, values=fresh["values"], rets=fresh["rets"], advs=fresh["advs"])
            perm = np.random.permutation(n) if perms is None else perms[k]
            approx_kl, iters = 0.0, 0
            for idx in split_chunks(n, batch_size, perm, merge_last=True):
                loss, dist, st = self._minibatch_losses(pb, idx, lagrangians, rescaling)
                approx_kl += st["loss/kl"]
                iters += 1
                self.optim.zero_grad()
                loss.backward()
                if cfg.max_grad_norm:
                    torch.nn.utils.clip_grad_norm_(self._leaves, max_norm=cfg.max_grad_norm)
                self.optim.step()
                self.gradient_steps += 1
                st["loss/total"] = loss.item()
                st["loss/entropy"] = dist.entropy().mean().item()
                rows.append([st.get(key, 0.0) for key in STAT_KEYS])
            approx_kl /= iters + 1e-7
            if approx_kl > 1.5 * cfg.target_kl:
                stopped_at = k
                break
        return np.array(rows, np.float64), stopped_at

    def update(self, data: OnPolicyData, lagrangians, rescaling, batch_size, repeat, perms=None):
        pb = self.process(data)
        stats, stopped_at = self.learn(pb, lagrangians, rescaling, batch_size, repeat, perms, data=data)
        return pb, stats, stopped_at
