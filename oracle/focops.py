"""CPU oracle of the FOCOPS update (torch fp32 + float64 GAE scan).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  A functional restatement of

  FOCOPS.process_fn     fsrl/policy/focops.py:130-152   (GAE, logp_old, mean_old, std_old; no full-batch norm)
  FOCOPS.nu_loss        fsrl/policy/focops.py:154-159   (nu += nu_lr * (J_c - d), clamp [0, nu_max])
  FOCOPS.critics_loss   fsrl/policy/focops.py:161-177   (per-critic MSE + l2 * sum(theta^2), own Adam)
  FOCOPS.policy_loss    fsrl/policy/focops.py:179-215   (KL(new||old) - ratio (A_r - nu A_c) / lambda, masked by
                                                         KL <= eta; per-minibatch advantage normalisation;
                                                         clip_grad_norm_ on the ACTOR only; own Adam)
  FOCOPS.learn          fsrl/policy/focops.py:217-251   (pass-level KL early stop with the +1e-7 quirk)

Nets / layout / GAE are those of the PPO oracle.  Pinned against tests/golden/focops_*.npz."""
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch
from torch.distributions import Independent, Normal, kl_divergence

from .ppo_lag import OnPolicyData, PPOLagConfig, PPOLagOracle, split_chunks


@dataclass
class FOCOPSConfig:
    obs_dim: int
    act_dim: int
    hidden: Tuple[int, int] = (128, 128)
    max_action: float = 1.0
    gamma: float = 0.99
    gae_lambda: float = 0.95
    actor_lr: float = 5e-4
    critic_lr: float = 1e-3
    l2_reg: float = 1e-3
    delta: float = 0.02
    eta: float = 0.02
    tem_lambda: float = 0.95
    max_grad_norm: Optional[float] = 0.5
    advantage_normalization: bool = True
    nu_max: float = 2.0
    nu_lr: float = 1e-2
    cost_limit: float = 10.0
    unbounded: bool = False      # ActorProb(unbounded=True)
    recompute_advantage: bool = False    # focops.py:223-226


class FOCOPSOracle(PPOLagOracle):
    def __init__(self, cfg: FOCOPSConfig, dtype=torch.float32):
        self.fcfg = cfg
        super().__init__(PPOLagConfig(obs_dim=cfg.obs_dim, act_dim=cfg.act_dim, hidden=cfg.hidden, max_action=cfg.max_action,
                                      gamma=cfg.gamma, gae_lambda=cfg.gae_lambda, lr=cfg.actor_lr,
                                      unbounded=cfg.unbounded), dtype)

    def set_params(self, flat, nu=0.0):
        super().set_params(flat)
        self.actor_optim = torch.optim.Adam(list(self.nets[0].values()), lr=self.fcfg.actor_lr)
        self.critic_optim = torch.optim.Adam([t for net in self.nets[1:] for t in net.values()], lr=self.fcfg.critic_lr)
        self.nu = float(nu)

    def process(self, data: OnPolicyData):
        pb = super().process(data)
        with torch.no_grad():
            d = self.actor_dist(pb["obs"])
            pb["mean_old"], pb["std_old"] = d.base_dist.loc.clone(), d.base_dist.scale.clone()
        return pb

    def nu_step(self, ave_cost_return):
        loss_nu = self.fcfg.cost_limit - ave_cost_return
        nu = torch.zeros(1) + self.nu                      # the reference keeps nu as a float32 tensor
        nu += -self.fcfg.nu_lr * loss_nu
        self.nu = float(torch.clamp(nu, 0, self.fcfg.nu_max))
        return {"loss/nu_loss": loss_nu, "loss/nu_value": self.nu}

    def step(self, pb, chunk):
        c = self.fcfg
        idx = torch.as_tensor(np.asarray(chunk))
        obs, act = pb["obs"][idx], pb["act"][idx]
        # ---- critics
        total, sc = torch.zeros(1), {}
        for i in range(2):
            vf = (pb["rets"][idx][..., i] - self.value(i, obs)).pow(2).mean()
            for p in self.nets[1 + i].values():
                vf = vf + p.pow(2).sum() * c.l2_reg
            total = total + vf
            sc["loss/vf" + str(i)] = vf.item()
        self.critic_optim.zero_grad()
        total.backward()
        self.critic_optim.step()
        sc["loss/vf_total"] = total.item()
        # ---- actor
        dist = self.actor_dist(obs)
        ent = dist.entropy().mean()
        ratio = (dist.log_prob(act) - pb["logp_old"][idx]).exp()
        kl = kl_divergence(dist, Independent(Normal(pb["mean_old"][idx], pb["std_old"][idx]), 1))
        advs = pb["advs"][idx].clone()
        if c.advantage_normalization:
            for i in range(2):
                a = advs[..., i]
                advs[..., i] = (a - a.mean()) / a.std()
        loss = ((kl - 1 / c.tem_lambda * ratio * (advs[..., 0] - self.nu * advs[..., 1])) * (kl.detach() <= c.eta)).mean()
        self.actor_optim.zero_grad()
        loss.backward()
        if c.max_grad_norm:
            torch.nn.utils.clip_grad_norm_(list(self.nets[0].values()), max_norm=c.max_grad_norm)
        self.actor_optim.step()
        return {"loss/actor_loss": loss.item(), "loss/kl": kl.mean().item(), "loss/entropy": ent.item()}, sc

    def update(self, data: OnPolicyData, ave_cost_return, batch_size, repeat, perms):
        pb = pb0 = self.process(data)
        snu = self.nu_step(ave_cost_return)
        rows, stopped = [], -1
        for k in range(repeat):
            if self.fcfg.recompute_advantage and k > 0:      # values / rets / advs from the current critics; old dist stays
                fresh = PPOLagOracle.process(self, data)
                pb = dict(pb, values=fresh["values"], rets=fresh["rets"], advs=fresh["advs"])
            kl_sum, n = 0.0, 0
            for chunk in split_chunks(len(data), batch_size, perms[k]):
                sa, sc = self.step(pb, chunk)
                rows.append((dict(snu), sa, sc))
                kl_sum += sa["loss/kl"]; n += 1
            if kl_sum / (n + 1e-7) > self.fcfg.delta:
                stopped = k
                break
        return pb0, rows, stopped
