#!/usr/bin/env python
"""Golden vectors G11: CVPO.update (fsrl/policy/cvpo.py) from the UNMODIFIED reference, build container
only.  Records the replay store, parameters, the sampled indices and every N(0,1) draw the update makes
(torch.normal(mean, std) == randn(shape) * std + mean bit for bit on CPU, checked below), and the per-update
stats / parameters / duals after a few collect cycles (pre_update_fn .. updates .. post_update_fn).

    python tests/golden/gen_golden_cvpo.py
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
from fsrl.policy import CVPO  # noqa: E402
from fsrl.utils.net.common import ActorCritic  # noqa: E402
from fsrl.utils.net.continuous import DoubleCritic, SingleCritic  # noqa: E402
from torch import nn  # noqa: E402
from torch.distributions import Independent, Normal  # noqa: E402

from gen_golden import CaptureLogger, fill_buffer, seed_all  # noqa: E402
from ref_shim import ActorProb, Net, _Box  # noqa: E402


def flat(mods):
    return torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()]).numpy().copy()


def gen(name, obs_dim, act_dim, hidden, env_num, ep_lens, batch_size, cycles, updates_per_cycle, seed, n_step=2,
        max_action=1.0, cost_limit=10.0, max_episode_steps=100, tau=0.05, actor_lr=5e-4, critic_lr=1e-3,
        gamma=0.98, double_critic=False, sample_act_num=16, estep_iter_num=1, mstep_iter_num=1, estep_kl=0.02,
        estep_dual_max=20.0, estep_dual_lr=0.02, mstep_kl_mu=0.005, mstep_kl_std=0.0005, mstep_dual_max=0.5,
        mstep_dual_lr=0.1):
    seed_all(seed)
    actor = ActorProb(Net((obs_dim, ), hidden_sizes=hidden), (act_dim, ), max_action=max_action,
                      conditioned_sigma=True, unbounded=False)
    actor_optim = torch.optim.Adam(actor.parameters(), lr=actor_lr)
    critics = []
    for _ in range(2):
        if double_critic:
            critics.append(DoubleCritic(Net((obs_dim, ), (act_dim, ), hidden_sizes=hidden, concat=True),
                                        Net((obs_dim, ), (act_dim, ), hidden_sizes=hidden, concat=True)))
        else:
            critics.append(SingleCritic(Net((obs_dim, ), (act_dim, ), hidden_sizes=hidden, concat=True)))
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
        # a moderate exploration width (sigma ~ e^-1) like a partly trained policy
        sig = [m for m in actor.sigma.modules() if isinstance(m, torch.nn.Linear)][-1]
        sig.bias.add_(-1.0)
        sig.weight.mul_(0.3)

    def dist(*logits):
        return Independent(Normal(*logits), 1)

    logger = CaptureLogger()
    policy = CVPO(actor=actor, critics=critics, actor_optim=actor_optim, critic_optim=critic_optim, logger=logger,
                  action_space=_Box(-max_action, max_action, (act_dim, )), dist_fn=dist,
                  max_episode_steps=max_episode_steps, cost_limit=cost_limit, tau=tau, gamma=gamma, n_step=n_step,
                  estep_iter_num=estep_iter_num, estep_kl=estep_kl, estep_dual_max=estep_dual_max,
                  estep_dual_lr=estep_dual_lr, sample_act_num=sample_act_num, mstep_iter_num=mstep_iter_num,
                  mstep_kl_mu=mstep_kl_mu, mstep_kl_std=mstep_kl_std, mstep_dual_max=mstep_dual_max,
                  mstep_dual_lr=mstep_dual_lr)
    policy.train()
    rng = np.random.default_rng(seed + 1000)
    buf = fill_buffer(rng, env_num, ep_lens, obs_dim, act_dim, buffer_size=100000)
    buf._meta["act"][:] = np.clip(2.0 * buf._meta["act"], -max_action, max_action)
    out = {"theta_actor0": flat([actor]), "theta_critics0": flat(critics)}
    meta = buf._meta
    used = np.concatenate([np.arange(o, o + len(b)) for o, b in zip(buf._offset, buf.buffers)])
    out["slots"] = used
    for k in ("obs", "act", "rew", "terminated", "truncated", "obs_next"):
        out["st_" + k] = meta[k][used]
    out["st_cost"] = meta["info.cost"][used]
    out["env_rows"] = np.array([len(b) for b in buf.buffers])
    out["sub_size"] = np.array(buf.buffers[0].maxsize)
    out["qc_thres"] = np.array(policy.qc_thres, np.float64)

    idx_log, eps_log = [], []
    orig_sample, orig_normal = buf.sample, torch.normal

    def rec_sample(bs):
        b, idx = orig_sample(bs)
        idx_log.append(np.asarray(idx).copy())
        return b, idx

    def rec_normal(mean, std, *a, **k):
        assert not a and not k and torch.is_tensor(mean) and torch.is_tensor(std)
        eps = torch.randn(mean.shape)          # same stream consumption as torch.normal(mean, std)
        eps_log.append(eps.numpy().copy())
        return eps * std + mean                # == at::normal's output.mul_(std).add_(mean)

    # the decomposition above is the identity this generator relies on
    m_, s_ = torch.randn(16, 7, 3), torch.rand(16, 7, 3) + 0.1
    torch.manual_seed(1)
    ref_draw = orig_normal(m_, s_)
    torch.manual_seed(1)
    assert torch.equal(ref_draw, torch.randn(16, 7, 3) * s_ + m_)

    buf.sample = rec_sample
    n_updates = cycles * updates_per_cycle
    n_draws = 3 + mstep_iter_num
    duals, mduals, actor_old_log = [], [], []
    torch.normal = rec_normal
    try:
        seed_all(seed + 7)
        for _ in range(cycles):
            policy.pre_update_fn(stats_train={"cost": 0.0})
            for _ in range(updates_per_cycle):
                policy.update(batch_size, buf)
                duals.append(policy.estep_dual.detach().numpy().copy())
                mduals.append([policy.mstep_dual_mu.item(), policy.mstep_dual_std.item()])
            policy.post_update_fn(stats_train={"cost": 0.0})
            actor_old_log.append(flat([policy.actor_old]))
    finally:
        torch.normal = orig_normal
    assert len(idx_log) == n_updates and len(eps_log) == n_draws * n_updates, (len(idx_log), len(eps_log))
    out["indices"] = np.stack(idx_log)                                 # [U][B]
    out["eps_target"] = np.stack(eps_log[0::n_draws])                  # [U][B][Da]   a' ~ pi(s_{t+n})   (used)
    out["eps_old_fwd"] = np.stack(eps_log[1::n_draws])                 # [U][B][Da]   forward(actor_old).act (unused)
    out["eps_particles"] = np.stack(eps_log[2::n_draws])               # [U][K][B][Da] the K particles   (used)
    out["eps_mstep"] = np.stack([np.stack(eps_log[u * n_draws + 3:(u + 1) * n_draws]) for u in range(n_updates)])
    # ---- one merged stats row per update (a row ends with the critic stats)
    merged, cur = [], {}
    for r in logger.rows:
        for k, v in r.items():
            if k.endswith("_time"):
                continue
            if k in cur and k.startswith(("mstep/", "loss/estep_loss")):
                continue            # mstep_iter_num/estep_iter_num > 1: keep the first iteration's row
            cur[k] = v
        if "loss/q_total" in r:
            merged.append(cur)
            cur = {}
    assert len(merged) == n_updates
    keys = list(merged[0].keys())
    out["stats_keys"] = np.array(keys)
    out["stats"] = np.array([[m[k] for k in keys] for m in merged], np.float64)
    out["estep_dual"] = np.array(duals, np.float32)                    # after each update
    out["mstep_dual"] = np.array(mduals, np.float32)
    out["theta_actor_final"] = flat([actor])
    out["theta_actor_old_cycles"] = np.stack(actor_old_log)
    out["theta_critics_final"] = flat(critics)
    out["theta_critics_old_final"] = flat(list(policy.critics_old))
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, batch_size=batch_size,
               cycles=cycles, updates_per_cycle=updates_per_cycle, seed=seed, n_step=n_step, max_action=max_action,
               cost_limit=cost_limit, max_episode_steps=max_episode_steps, tau=tau, actor_lr=actor_lr,
               critic_lr=critic_lr, gamma=gamma, double_critic=double_critic, sample_act_num=sample_act_num,
               estep_iter_num=estep_iter_num, mstep_iter_num=mstep_iter_num, estep_kl=estep_kl,
               estep_dual_max=estep_dual_max, estep_dual_lr=estep_dual_lr, mstep_kl_mu=mstep_kl_mu,
               mstep_kl_std=mstep_kl_std, mstep_dual_max=mstep_dual_max, mstep_dual_lr=mstep_dual_lr,
               buffer_size=100000)
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"cvpo_{name}.npz"), **out)
    print(f"G11 cvpo_{name}.npz rows={len(used)} updates={n_updates} keys={keys}\n    last={merged[-1]}\n"
          f"    estep_dual={duals[-1]} mstep_dual={mduals[-1]}")


if __name__ == "__main__":
    torch.set_num_threads(4)
    eps = [[60, 50, -17], [70, 55], [40, 40, 40, -9]]
    if sys.argv[1:] == ["depths"]:
        # hidden_sizes the fused kernels do not hold (cvpo_agent.py: any tuple): layered contexts on the HIP side
        gen("deep3", 6, 3, (48, 64, 40), 3, eps, batch_size=64, cycles=2, updates_per_cycle=4, seed=53, cost_limit=0.3,
            mstep_kl_mu=2e-4, mstep_kl_std=2e-6, actor_lr=2e-3)
        gen("wide1_double", 8, 2, (272, ), 3, eps, batch_size=100, cycles=2, updates_per_cycle=3, seed=54, n_step=3, max_action=2.0,
            cost_limit=0.5, mstep_kl_mu=1e-3, double_critic=True, mstep_iter_num=2, estep_iter_num=2, sample_act_num=8)
        sys.exit(0)
    # small: cost threshold below Qc (lambda grows) and tight KL bounds (the M-step duals turn positive)
    gen("small", 6, 3, (64, 64), 3, eps, batch_size=64, cycles=2, updates_per_cycle=6, seed=50, cost_limit=0.3,
        mstep_kl_mu=2e-4, mstep_kl_std=2e-6, actor_lr=2e-3)
    gen("default", 17, 6, (128, 128), 4, [[200, 150], [200, -120], [250, 100], [300]], batch_size=256, cycles=2,
        updates_per_cycle=3, seed=51, max_action=1.0)
    gen("double", 8, 2, (64, 64), 3, eps, batch_size=100, cycles=2, updates_per_cycle=3, seed=52, n_step=3,
        max_action=2.0, cost_limit=0.5, mstep_kl_mu=1e-3, double_critic=True, mstep_iter_num=2, estep_iter_num=2, sample_act_num=8)
