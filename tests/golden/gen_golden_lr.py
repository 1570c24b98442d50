#!/usr/bin/env python
"""G4b: PPOLagrangian.update with an lr_scheduler (fsrl/policy/base_policy.py:352-354: stepped once at the END of every
update()), from the UNMODIFIED reference.  Two consecutive updates on one buffer with LambdaLR(optim, 0.5 ** epoch): the
first runs at lr, the second at lr / 2 (Adam state carried over).  Build container only.

    python tests/golden/gen_golden_lr.py        # writes tests/golden/ppo_lrsched.npz
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import CaptureLogger, build_ppo, fill_buffer, flat_params, seed_all  # noqa: E402  (installs ref_shim)

KEYS = ["loss/rescaling", "loss/lagrangian", "loss/actor_safety", "loss/actor_rew", "loss/actor_total", "loss/kl",
        "loss/vf0", "loss/vf1", "loss/vf_total", "loss/total", "loss/entropy"]


def main():
    torch.set_num_threads(4)
    cfg = dict(obs_dim=6, act_dim=2, hidden=[64, 64], env_num=3, batch_size=64, repeat=2, seed=11, cost_stat=25.0,
               cost_limit=10.0, max_action=1.0, lr=1e-3, max_grad_norm=0.5, target_kl=1e9, gamma_lr=0.5)
    logger = CaptureLogger()
    policy, actor_critic, optim = build_ppo(cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"]), cfg["seed"], logger=logger,
                                            lr=cfg["lr"], cost_limit=cfg["cost_limit"], max_grad_norm=cfg["max_grad_norm"],
                                            target_kl=cfg["target_kl"])
    policy.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(optim, lambda e: cfg["gamma_lr"] ** e)
    policy.train()
    rng = np.random.default_rng(cfg["seed"] + 1000)
    buf = fill_buffer(rng, cfg["env_num"], [[60, 50, -20], [70, 70], [45, 45, 45]], cfg["obs_dim"], cfg["act_dim"])
    out = {"theta0": flat_params(actor_critic)}
    policy.pre_update_fn(stats_train={"cost": cfg["cost_stat"]})
    out["lagrangian"] = np.array([o.get_lag() for o in policy.lag_optims], np.float64)
    batch, indices = buf.sample(0)
    out["indices"] = indices
    for k in ("obs", "act", "rew", "terminated", "truncated", "obs_next"):
        out["buf_" + k] = getattr(batch, k)
    out["buf_cost"] = batch.info.cost
    out["unfinished_index"] = buf.unfinished_index()
    out["env_rows"] = np.array([len(b) for b in buf.buffers])
    perms = []
    orig_perm = np.random.permutation

    def rec_perm(n):
        p = orig_perm(n)
        perms.append(np.asarray(p).copy())
        return p

    np.random.permutation = rec_perm
    lrs = [optim.param_groups[0]["lr"]]
    try:
        seed_all(cfg["seed"] + 7)
        for u in range(2):
            n0 = len(logger.rows)
            policy.update(0, buf, batch_size=cfg["batch_size"], repeat=cfg["repeat"])
            rows = [r for r in logger.rows[n0:] if "update/gradient_steps" not in r]
            stats = []
            for i in range(0, len(rows), 3):
                merged = {}
                for r in rows[i:i + 3]:
                    merged.update(r)
                stats.append([merged[k] for k in KEYS])
            out[f"stats{u}"] = np.array(stats, np.float64)
            out[f"theta_after{u}"] = flat_params(actor_critic)
            lrs.append(optim.param_groups[0]["lr"])
    finally:
        np.random.permutation = orig_perm
    out["perms"] = np.stack(perms)            # [2 updates * repeat][N]
    out["lrs"] = np.array(lrs, np.float64)    # before update 0, after update 0, after update 1
    out["stats_keys"] = np.array(KEYS)
    cfg.update(dict(vf_coef=0.25, gae_lambda=0.95, eps_clip=0.2, dual_clip=None, gamma=0.99, advantage_normalization=True,
                    use_lagrangian=True))
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, "ppo_lrsched.npz"), **out)
    print("G4b ppo_lrsched.npz  N=%d  lrs=%s  steps/update=%d" % (len(indices), lrs, len(out["stats0"])))


if __name__ == "__main__":
    main()
