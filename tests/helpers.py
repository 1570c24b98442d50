"""Shared helpers for the parity tests (golden-case loading)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def ppo_case(name):
    """-> (cfg dict, golden arrays) for tests/golden/ppo_<name>.npz"""
    g = load_npz(f"ppo_{name}.npz")
    cfg = json.loads(str(g["cfg_json"]))
    return cfg, g


def end_flag_of(g):
    """done | unfinished tail, in sample(0) order (fsrl/policy/base_policy.py:409-411)."""
    end = np.logical_or(g["buf_terminated"], g["buf_truncated"])
    end[np.isin(g["indices"], g["unfinished_index"])] = True
    return end


def oracle_cfg_and_data(cfg, g):
    from oracle.ppo_lag import OnPolicyData, PPOLagConfig
    ocfg = PPOLagConfig(obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"], hidden=tuple(cfg["hidden"]),
                        max_action=cfg["max_action"], gamma=cfg["gamma"],
                        gae_lambda=cfg["gae_lambda"], eps_clip=cfg["eps_clip"],
                        dual_clip=cfg["dual_clip"], vf_coef=cfg["vf_coef"],
                        max_grad_norm=cfg["max_grad_norm"], target_kl=cfg["target_kl"],
                        advantage_normalization=cfg["advantage_normalization"],
                        use_lagrangian=cfg["use_lagrangian"], lr=cfg["lr"],
                        recompute_advantage=bool(cfg.get("recompute_advantage", False)),
                        unbounded=bool(cfg.get("unbounded", False)),
                        reward_normalization=bool(cfg.get("reward_normalization", False)),
                        value_clip=bool(cfg.get("value_clip", False)))
    data = OnPolicyData(obs=g["buf_obs"], act=g["buf_act"], rew=g["buf_rew"], cost=g["buf_cost"],
                        terminated=g["buf_terminated"], truncated=g["buf_truncated"],
                        obs_next=g["buf_obs_next"], end_flag=end_flag_of(g))
    return ocfg, data
