#!/usr/bin/env python
"""Golden vectors G10: FOCOPS.update (fsrl/policy/focops.py:126-251) from the UNMODIFIED reference, build
container only.  Records the on-policy store, parameters and the minibatch permutations; outputs are the
processed batch, the per-minibatch stats rows and the parameters after the update.

    python tests/golden/gen_golden_focops.py
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
from fsrl.policy import FOCOPS  # noqa: E402
from torch import nn  # noqa: E402

from gen_golden import CaptureLogger, fill_buffer, flat_params  # noqa: E402
from gen_golden_trust import PermRecorder, build_nets, dist, record_batch  # noqa: E402
from ref_shim import _Box  # noqa: E402


def gen(name, obs_dim, act_dim, hidden, env_num, ep_lens, batch_size, repeat, seed, cost_stat, cost_limit=10.0,
        actor_lr=5e-4, critic_lr=1e-3, auto_nu=True, nu=0.01, nu_max=2.0, nu_lr=1e-2, prior_updates=0, unbounded=False, **kw):
    actor, critic, ac = build_nets(obs_dim, act_dim, hidden, seed, unbounded)
    g = torch.Generator().manual_seed(seed + 3)
    with torch.no_grad():
        for p in ac.parameters():
            if p.ndim == 1 and p.numel() > act_dim:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    actor_optim = torch.optim.Adam(actor.parameters(), lr=actor_lr)
    critic_optim = torch.optim.Adam(nn.ModuleList(critic).parameters(), lr=critic_lr)
    nu_arg = (nu_max, nu_lr, torch.zeros(1) + nu) if auto_nu else nu
    logger = CaptureLogger()
    policy = FOCOPS(actor, critic, actor_optim, critic_optim, dist, logger=logger, cost_limit=cost_limit, nu=nu_arg,
                    observation_space=_Box(-np.inf, np.inf, (obs_dim, )), action_space=_Box(-1, 1, (act_dim, )), **kw)
    policy.train()
    buf = fill_buffer(np.random.default_rng(seed + 1000), env_num, ep_lens, obs_dim, act_dim)
    out = {"theta0": flat_params(ac)}
    record_batch(out, buf)
    policy.pre_update_fn(stats_train={"cost": cost_stat})
    batch, indices = buf.sample(0)
    pb = policy.process_fn(batch, buf, indices)
    for k in ("advs", "rets", "logp_old", "mean_old", "std_old"):
        out[k] = getattr(pb, k).numpy().copy()
    out["nu0"] = np.array(float(policy._nu))
    with PermRecorder() as pr:
        policy.update(0, buf, batch_size=batch_size, repeat=repeat)
    out["perms"] = np.stack(pr.perms)
    rows = [r for r in logger.rows if "update/gradient_steps" not in r]
    assert len(rows) % 3 == 0
    steps = len(rows) // 3
    kn, ka, kc = list(rows[0].keys()), list(rows[1].keys()), list(rows[2].keys())
    out["stats_nu_keys"], out["stats_actor_keys"], out["stats_critic_keys"] = np.array(kn), np.array(ka), np.array(kc)
    out["stats_nu"] = np.array([[float(rows[3 * i][k]) for k in kn] for i in range(steps)], np.float64)
    out["stats_actor"] = np.array([[float(rows[3 * i + 1][k]) for k in ka] for i in range(steps)], np.float64)
    out["stats_critic"] = np.array([[float(rows[3 * i + 2][k]) for k in kc] for i in range(steps)], np.float64)
    out["early_stop"] = np.array(len(logger.prints) if hasattr(logger, "prints") else -1)
    out["theta_final"] = flat_params(ac)
    out["nu_final"] = np.array(float(policy._nu))
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, batch_size=batch_size, repeat=repeat,
               seed=seed, cost_stat=cost_stat, cost_limit=cost_limit, actor_lr=actor_lr, critic_lr=critic_lr, auto_nu=auto_nu,
               nu=nu, nu_max=nu_max, nu_lr=nu_lr, max_action=1.0, n_perms=len(pr.perms))
    defaults = dict(l2_reg=1e-3, delta=0.02, eta=0.02, tem_lambda=0.95, gae_lambda=0.95, max_grad_norm=0.5,
                    advantage_normalization=True, gamma=0.99)
    defaults.update(kw)
    cfg.update(defaults)
    if unbounded:
        cfg["unbounded"] = True
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"focops_{name}.npz"), **out)
    print(f"G10 focops_{name}.npz N={len(out['indices'])} steps={steps} perms={len(pr.perms)} keys={kn}{ka}{kc} "
          f"nu {float(out['nu0']):.4f}->{float(out['nu_final']):.4f} kl last={out['stats_actor'][-1][ka.index('loss/kl')]:.5f}")


if __name__ == "__main__":
    torch.set_num_threads(4)
    eps = [[60, 50, -17], [70, 55], [40, 40, 40, -9]]
    if sys.argv[1:] == ["options"]:
        gen("unbounded", 6, 2, (64, 64), 3, eps, batch_size=64, repeat=3, seed=53, cost_stat=25.0, unbounded=True)
        # recompute_advantage (focops.py:223-226): GAE from the current critics before passes 2 and 3
        gen("recompute", 6, 2, (64, 64), 3, eps, batch_size=64, repeat=3, seed=54, cost_stat=25.0, recompute_advantage=True)
        sys.exit(0)
    if sys.argv[1:] == ["depths"]:
        # hidden_sizes the fused kernels do not hold (focops_agent.py: any tuple): layered contexts on the HIP side
        gen("deep3", 6, 2, (48, 64, 32), 3, eps, batch_size=64, repeat=3, seed=55, cost_stat=25.0)
        gen("wide1", 7, 3, (272, ), 3, eps, batch_size=64, repeat=2, seed=56, cost_stat=12.0, nu=0.4, unbounded=True)
        sys.exit(0)
    gen("small", 6, 2, (64, 64), 3, eps, batch_size=64, repeat=3, seed=50, cost_stat=25.0)
    gen("c1", 8, 2, (128, 128), 4, [[150, 150], [300], [200, -60], [120, 120, -40]], batch_size=256, repeat=4, seed=51,
        cost_stat=4.0, nu=0.3)
    gen("earlystop", 8, 2, (64, 64), 3, eps, batch_size=32, repeat=6, seed=52, cost_stat=30.0, delta=0.004, eta=0.5,
        actor_lr=3e-3, max_grad_norm=None, nu=0.5)
