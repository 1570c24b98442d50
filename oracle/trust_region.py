"""CPU oracle of the trust-region policy updates: CPO and TRPO-Lagrangian (torch fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  A functional restatement of

  CPO.process_fn / critics_loss / _MVP / _conjugate_gradients / policy_loss / learn
      fsrl/policy/cpo.py:123-145, 147-162, 177-182, 184-204, 234-351, 353-370
  TRPOLagrangian.process_fn / policy_loss / learn / _MVP / _conjugate_gradients
      fsrl/policy/trpo_lag.py:117-134, 148-171, 173-251, 253-259, 261-283

on the flat parameter layout of oracle/layout.py.  Hessian-vector products use torch's double
backward exactly like the reference (grad of (grad KL . v)); the HIP path computes the same
quantity with an analytic R-op, so this oracle is what pins it.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch
from torch.distributions import Independent, Normal, kl_divergence

from .ppo_lag import OnPolicyData, PPOLagConfig, PPOLagOracle

EPS = 1e-8


def _permuted(pb, perm):
    idx = torch.as_tensor(np.asarray(perm), dtype=torch.long)
    return {k: v[idx] for k, v in pb.items()}


def _split(pb, perm, size):
    """tianshou-0.5 Batch.split(size, shuffle=True, merge_last=True) as FSRL calls it (cpo.py:358, trpo_lag.py:178): the rows
    in `perm` order (None: stored order), cut every `size` rows, a remainder merged into the last chunk."""
    n = len(next(iter(pb.values())))
    idx = np.arange(n) if perm is None else np.asarray(perm)
    merge = (n % size) > 0
    for i in range(0, n, size):
        if merge and i + 2 * size >= n:
            yield _permuted(pb, idx[i:]) if (perm is not None or i > 0) else pb
            return
        chunk = idx[i:i + size]
        yield _permuted(pb, chunk) if (perm is not None or len(chunk) < n) else pb


@dataclass
class CPOConfig:
    obs_dim: int
    act_dim: int
    hidden: Tuple[int, int] = (128, 128)
    max_action: float = 1.0
    gamma: float = 0.99
    gae_lambda: float = 0.95
    target_kl: float = 0.01
    backtrack_coeff: float = 0.8
    damping_coeff: float = 0.1
    max_backtracks: int = 10
    optim_critic_iters: int = 20
    l2_reg: float = 0.001
    advantage_normalization: bool = True
    cost_limit: float = 10.0
    lr: float = 1e-3           # Adam over the CRITIC parameters only (cpo_agent.py:147-148)
    unbounded: bool = False              # ActorProb(unbounded=True)
    reward_normalization: bool = False   # base_policy.py:430-444 (shared compute_gae_returns)


@dataclass
class TRPOConfig:
    obs_dim: int
    act_dim: int
    hidden: Tuple[int, int] = (128, 128)
    max_action: float = 1.0
    gamma: float = 0.99
    gae_lambda: float = 0.95
    target_kl: float = 0.001
    backtrack_coeff: float = 0.8
    max_backtracks: int = 10
    optim_critic_iters: int = 5
    advantage_normalization: bool = True
    use_lagrangian: bool = True
    lr: float = 5e-4           # Adam over all params; only the critics ever receive gradients
    damping: float = 0.1       # trpo_lag.py:115
    unbounded: bool = False
    reward_normalization: bool = False


class _TrustRegionBase(PPOLagOracle):
    """Shares nets / layout / GAE with the PPO oracle; replaces the optimiser wiring."""

    def __init__(self, cfg, dtype=torch.float32):
        base = PPOLagConfig(obs_dim=cfg.obs_dim, act_dim=cfg.act_dim, hidden=cfg.hidden,
                            max_action=cfg.max_action, gamma=cfg.gamma, gae_lambda=cfg.gae_lambda, lr=cfg.lr,
                            unbounded=cfg.unbounded, reward_normalization=cfg.reward_normalization)
        self.tcfg = cfg
        super().__init__(base, dtype)

    def set_params(self, flat):
        super().set_params(flat)
        critic_leaves = [t for net in self.nets[1:] for t in net.values()]
        self.critic_optim = torch.optim.Adam(critic_leaves, lr=self.tcfg.lr)

    # ---- flat views of the ACTOR parameters (torch parameters() order)
    @property
    def actor_leaves(self) -> List[torch.Tensor]:
        return list(self.nets[0].values())

    def actor_flat(self):
        return torch.cat([p.detach().reshape(-1) for p in self.actor_leaves])

    def set_actor_flat(self, flat):
        off = 0
        with torch.no_grad():
            for p in self.actor_leaves:
                n = p.numel()
                p.copy_(flat[off:off + n].view_as(p))
                off += n

    def flat_grad(self, y, retain_graph=False, create_graph=False):
        grads = torch.autograd.grad(y, self.actor_leaves, retain_graph=retain_graph or create_graph,
                                    create_graph=create_graph)
        return torch.cat([g.reshape(-1) for g in grads])

    def hvp(self, v, flat_kl_grad, damping):
        """H v + damping v with H = Hessian of the mean KL (grad of (grad KL . v))."""
        return self.flat_grad(torch.dot(flat_kl_grad, v), retain_graph=True) + v * damping

    def conjugate_gradients(self, g, flat_kl_grad, damping, nsteps=10, residual_tol=1e-8):
        x = torch.zeros_like(g)
        r, p = g.clone(), g.clone()
        rs_old = torch.sum(r * r)
        for _ in range(nsteps):
            z = self.hvp(p, flat_kl_grad, damping)
            alpha = rs_old / torch.sum(p * z)
            x += alpha * p
            r -= alpha * z
            rs_new = torch.sum(r * r)
            if rs_new < residual_tol:
                break
            p = r + (rs_new / rs_old) * p
            rs_old = rs_new
        return x

    def normalise_full_batch(self, pb):
        for i in range(self.cfg.n_critics):
            a = pb["advs"][..., i]
            pb["advs"][..., i] = (a - a.mean()) / a.std()


class CPOOracle(_TrustRegionBase):
    def process(self, data: OnPolicyData):
        pb = super().process(data)
        if self.tcfg.advantage_normalization:
            self.normalise_full_batch(pb)
        with torch.no_grad():
            d = self.actor_dist(pb["obs"])
            pb["mean_old"], pb["std_old"] = d.base_dist.loc.clone(), d.base_dist.scale.clone()
        return pb

    def critics_step(self, pb):
        """One Adam step on both critics: mean squared error + l2_reg * sum(theta^2)."""
        total, stats = torch.zeros(1), {}
        for i in range(self.cfg.n_critics):
            v = self.value(i, pb["obs"])
            vf = (pb["rets"][..., i] - v).pow(2).mean()
            for p in self.nets[1 + i].values():
                vf = vf + p.pow(2).sum() * self.tcfg.l2_reg
            total = total + vf
            stats["loss/vf" + str(i)] = vf.item()
        self.critic_optim.zero_grad()
        total.backward()
        self.critic_optim.step()
        stats["loss/vf_total"] = total.item()
        return stats

    def policy_step(self, pb, ave_cost_return):
        c = self.tcfg
        obs, act = pb["obs"], pb["act"]
        adv_r, adv_c, logp_old = pb["advs"][..., 0], pb["advs"][..., 1], pb["logp_old"]
        dist_old = Independent(Normal(pb["mean_old"], pb["std_old"]), 1)

        def surrogates(dist):
            logp = dist.log_prob(act)
            ratio = torch.exp(logp - logp_old)
            objective = torch.mean(ratio * adv_r)
            cost_sur = ave_cost_return + torch.mean(ratio * adv_c) - torch.mean(adv_c)
            return objective, cost_sur

        dist = self.actor_dist(obs)
        ent = dist.entropy().mean()
        kl = kl_divergence(dist_old, dist).mean()
        objective, cost_sur = surrogates(dist)
        g = self.flat_grad(objective, retain_graph=True)
        b = self.flat_grad(-cost_sur, retain_graph=True)
        kl_grad = self.flat_grad(kl, create_graph=True)
        H_inv_g = self.conjugate_gradients(g, kl_grad, c.damping_coeff)
        approx_g = self.hvp(H_inv_g, kl_grad, c.damping_coeff)
        c_value = cost_sur - c.cost_limit
        if torch.dot(b, b) <= EPS and c_value < 0:
            H_inv_b = s_r = s_s = A = B = torch.zeros(1)
            s_q = torch.dot(approx_g, H_inv_g)
            case = 4
        else:
            H_inv_b = self.conjugate_gradients(b, kl_grad, c.damping_coeff)
            approx_b = self.hvp(H_inv_b, kl_grad, c.damping_coeff)
            s_q, s_r, s_s = torch.dot(approx_g, H_inv_g), torch.dot(approx_g, H_inv_b), torch.dot(approx_b, H_inv_b)
            A = s_q - s_r**2 / s_s
            B = 2 * c.target_kl - c_value**2 / s_s
            if c_value < 0 and B < 0:
                case = 3
            elif c_value < 0 and B >= 0:
                case = 2
            elif c_value >= 0 and B >= 0:
                case = 1
            else:
                case = 0
        if case in (3, 4):
            lam = torch.sqrt(s_q / (2 * c.target_kl))
            nu = torch.zeros_like(lam)
        elif case in (1, 2):
            LA, LB = [0, s_r / c_value], [s_r / c_value, np.inf]
            LA, LB = (LA, LB) if c_value < 0 else (LB, LA)
            proj = lambda x, L: max(L[0], min(L[1], x))  # noqa: E731
            lam_a = proj(torch.sqrt(A / B), LA)
            lam_b = proj(torch.sqrt(s_q / (2 * c.target_kl)), LB)
            f_a = lambda lam: -0.5 * (A / (lam + EPS) + B * lam) - s_r * c_value / (s_s + EPS)  # noqa: E731
            f_b = lambda lam: -0.5 * (s_q / (lam + EPS) + 2 * c.target_kl * lam)  # noqa: E731
            lam = lam_a if f_a(lam_a) >= f_b(lam_b) else lam_b
            lam = torch.as_tensor(lam)
            nu = max(0, (lam * c_value - s_r).item()) / (s_s + EPS)
        else:
            nu = torch.sqrt(2 * c.target_kl / (s_s + EPS))
            lam = torch.zeros_like(nu)
        with torch.no_grad():
            step_dir = (1. / (lam + EPS)) * (H_inv_g + nu * H_inv_b) if case > 0 else nu * H_inv_b
            step_dir = step_dir / torch.norm(step_dir)
            beta = 1.0
            if not torch.isnan(lam):
                theta0 = self.actor_flat().clone()
                obj0, cs0 = objective.detach().clone(), cost_sur.detach().clone()
                for _ in range(c.max_backtracks):
                    self.set_actor_flat(beta * step_dir + theta0)
                    d_new = self.actor_dist(obs)
                    new_kl = kl_divergence(dist_old, d_new).mean().item()
                    obj_n, cs_n = surrogates(d_new)
                    if new_kl <= c.target_kl and (obj_n > obj0 if case > 1 else True) and \
                            cs_n - cs0 <= max(-c_value.item(), 0):
                        break
                    beta *= c.backtrack_coeff
        f = lambda t: float(torch.as_tensor(t).detach().reshape(-1)[0])  # noqa: E731
        stats = {"loss/kl": kl.item(), "loss/entropy": ent.item(), "loss/rew_loss": objective.item(),
                 "loss/cost_loss": cost_sur.item(), "loss/optim_A": f(A), "loss/optim_B": f(B),
                 "loss/optim_C": c_value.item(), "loss/optim_Q": f(s_q), "loss/optim_R": f(s_r),
                 "loss/optim_S": f(s_s), "loss/optim_lam": f(lam), "loss/optim_nu": f(nu),
                 "loss/optim_case": case, "loss/step_size": beta}
        return stats, H_inv_g.detach(), (H_inv_b.detach() if case != 4 else None)

    def update(self, data: OnPolicyData, ave_cost_return, repeat, perms=None, batch_size=99999):
        """`perms[k]`: the shuffle Batch.split(batch_size, merge_last=True) applies to the batch in repeat k (the reference
        draws np.random.permutation; None = keep the stored order).  One row per minibatch of every repeat (cpo.py:357-366)."""
        pb = self.process(data)
        rows = []
        for k in range(repeat):
            for mb in _split(pb, None if perms is None else perms[k], batch_size):
                for _ in range(self.tcfg.optim_critic_iters):
                    sc = self.critics_step(mb)
                sa, hg, hb = self.policy_step(mb, ave_cost_return)
                rows.append((sa, sc, hg, hb))
        return pb, rows


class TRPOLagOracle(_TrustRegionBase):
    def process(self, data: OnPolicyData):
        pb = super().process(data)
        if self.tcfg.advantage_normalization:
            self.normalise_full_batch(pb)
        return pb

    def _surrogate(self, dist, pb, lagrangians, rescaling):
        logp = dist.log_prob(pb["act"])
        ratio = (logp - pb["logp_old"]).exp().float()
        ratio = ratio.reshape(ratio.size(0), -1).transpose(0, 1)
        loss_rew = -(ratio * pb["advs"][..., 0]).mean()
        loss_safety = 0.0
        stats = {"loss/rescaling": rescaling}
        if self.tcfg.use_lagrangian:
            for i in range(1, self.cfg.n_critics):
                li = torch.mean(ratio * pb["advs"][..., i] * lagrangians[i - 1])
                loss_safety = loss_safety + li
                stats["loss/lagrangian"] = lagrangians[i - 1]
                stats["loss/actor_safety"] = li.item()
        total = rescaling * (loss_rew + loss_safety)
        stats.update({"loss/actor_rew": loss_rew.item(), "loss/actor_total": total.item()})
        return total, stats

    def learn_step(self, pb, lagrangians, rescaling):
        c = self.tcfg
        dist = self.actor_dist(pb["obs"])
        loss_actor, stats = self._surrogate(dist, pb, lagrangians, rescaling)
        g = self.flat_grad(loss_actor, retain_graph=True).detach()
        with torch.no_grad():
            old = self.actor_dist(pb["obs"])
            old_dist = Independent(Normal(old.base_dist.loc.clone(), old.base_dist.scale.clone()), 1)
        kl = kl_divergence(old_dist, dist).mean()
        kl_grad = self.flat_grad(kl, create_graph=True)
        hv = lambda v: self.flat_grad((kl_grad * v).sum(), retain_graph=True).detach() + v * c.damping  # noqa: E731
        # CG with TRPO's tolerance (1e-10)
        x = torch.zeros_like(g)
        r, p = g.clone(), g.clone()
        rdotr = r.dot(r)
        for _ in range(10):
            z = hv(p)
            alpha = rdotr / p.dot(z)
            x += alpha * p
            r -= alpha * z
            new_rdotr = r.dot(r)
            if new_rdotr < 1e-10:
                break
            p = r + new_rdotr / rdotr * p
            rdotr = new_rdotr
        direction = -x
        step_size = torch.sqrt(2 * c.target_kl / (direction * hv(direction)).sum(0, keepdim=True))
        with torch.no_grad():
            theta0 = self.actor_flat().clone()
            for i in range(c.max_backtracks):
                new_theta = theta0 + step_size * direction
                self.set_actor_flat(new_theta)
                new_dist = self.actor_dist(pb["obs"])
                loss_new, _ = self._surrogate(new_dist, pb, lagrangians, rescaling)
                kl = kl_divergence(old_dist, new_dist).mean()
                if kl < c.target_kl and loss_new < loss_actor:
                    break
                elif i < c.max_backtracks - 1:
                    step_size = step_size * c.backtrack_coeff
                else:
                    self.set_actor_flat(new_theta)
                    step_size = torch.tensor([0.0])
        for _ in range(c.optim_critic_iters):
            loss_vf, sc = 0, {}
            for i in range(self.cfg.n_critics):
                vf = (pb["rets"][..., i] - self.value(i, pb["obs"])).pow(2).mean()
                loss_vf = loss_vf + vf
                sc["loss/vf" + str(i)] = vf.item()
            sc["loss/vf_total"] = loss_vf.item()
            self.critic_optim.zero_grad()
            loss_vf.backward()
            self.critic_optim.step()
            self.gradient_steps += 1
        stats.update(sc)
        stats.update({"loss/kl": kl.item(), "loss/step_size": step_size.item(),
                      "loss/entropy": dist.entropy().mean().item()})
        return stats, x.detach()

    def update(self, data: OnPolicyData, lagrangians, rescaling, repeat, perms=None, batch_size=99999):
        """One row per minibatch of every repeat (trpo_lag.py:177-178: batch.split(batch_size, merge_last=True))."""
        pb = self.process(data)
        rows = []
        for k in range(repeat):
            for mb in _split(pb, None if perms is None else perms[k], batch_size):
                rows.append(self.learn_step(mb, lagrangians, rescaling))
        return pb, rows
