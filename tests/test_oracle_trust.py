"""Oracle CPO / TRPO-Lagrangian vs golden vectors recorded from the unmodified reference."""
import json

import numpy as np
import pytest
import torch

from helpers import end_flag_of, load_npz, trust_full_case
from oracle.ppo_lag import OnPolicyData
from oracle.trust_region import CPOConfig, CPOOracle, TRPOConfig, TRPOLagOracle


def _data(g):
    return OnPolicyData(obs=g["buf_obs"], act=g["buf_act"], rew=g["buf_rew"], cost=g["buf_cost"],
                        terminated=g["buf_terminated"], truncated=g["buf_truncated"],
                        obs_next=g["buf_obs_next"], end_flag=end_flag_of(g))


def cpo_cfg(cfg):
    return CPOConfig(obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"], hidden=tuple(cfg["hidden"]),
                     gamma=cfg["gamma"], gae_lambda=cfg["gae_lambda"], target_kl=cfg["target_kl"],
                     backtrack_coeff=cfg["backtrack_coeff"], damping_coeff=cfg["damping_coeff"],
                     max_backtracks=cfg["max_backtracks"], optim_critic_iters=cfg["optim_critic_iters"],
                     l2_reg=cfg["l2_reg"], advantage_normalization=cfg["advantage_normalization"],
                     cost_limit=cfg["cost_limit"], lr=cfg["lr"], unbounded=bool(cfg.get("unbounded", False)),
                     reward_normalization=bool(cfg.get("reward_normalization", False)))


def trpo_cfg(cfg):
    return TRPOConfig(obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"], hidden=tuple(cfg["hidden"]),
                      gamma=cfg["gamma"], gae_lambda=cfg["gae_lambda"], target_kl=cfg["target_kl"],
                      backtrack_coeff=cfg["backtrack_coeff"], max_backtracks=cfg["max_backtracks"],
                      optim_critic_iters=cfg["optim_critic_iters"],
                      advantage_normalization=cfg["advantage_normalization"],
                      use_lagrangian=cfg["use_lagrangian"], lr=cfg["lr"], unbounded=bool(cfg.get("unbounded", False)),
                      reward_normalization=bool(cfg.get("reward_normalization", False)))


def cpo_case(name):
    """c3full: BASELINE configs[2] at full size (rollout and theta0 regenerated from seeds, helpers.trust_full_case)"""
    return trust_full_case(name) if name.endswith("full") else load_npz(f"cpo_{name}.npz")


@pytest.mark.parametrize("name", ["infeasible", "feasible", "edge", "case1", "case2", "case4", "options", "minibatch", "widths", "wideobs", "deep3", "wide1", "c3full"])
def test_cpo_update(name):
    torch.set_num_threads(4)
    g = cpo_case(name)
    cfg = json.loads(str(g["cfg_json"]))
    o = CPOOracle(cpo_cfg(cfg))
    o.set_params(g["theta0"])
    if "ret_rms0" in g:
        o.ret_rms[:] = g["ret_rms0"]
    pb, rows = o.update(_data(g), cfg["cost_stat"], cfg["repeat"], perms=g["perms"], batch_size=cfg.get("batch_size", 99999))
    if "ret_rms0" in g:
        np.testing.assert_allclose(o.ret_rms, g["ret_rms_final"], rtol=1e-10)
    np.testing.assert_allclose(pb["advs"].numpy(), g["advs_norm"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pb["mean_old"].numpy(), g["mean_old"], rtol=0, atol=1e-6)
    ka = [str(k) for k in g["stats_actor_keys"]]
    kc = [str(k) for k in g["stats_critic_keys"]]
    got_a = np.array([[r[0][k] for k in ka] for r in rows])
    got_c = np.array([[r[1][k] for k in kc] for r in rows])
    np.testing.assert_allclose(got_a, g["stats_actor"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got_c, g["stats_critic"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rows[0][2].numpy(), g["H_inv_g_first"], rtol=1e-4, atol=1e-6)
    if "H_inv_b_first" in g:
        np.testing.assert_allclose(rows[0][3].numpy(), g["H_inv_b_first"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(o.get_params(), g["theta_final"], rtol=0, atol=2e-6)


def test_cpo_policy_loss_away_from_theta_old():
    """theta != theta_old: the Hessian of the KL is NOT the Fisher matrix here."""
    torch.set_num_threads(4)
    g = load_npz("cpo_perturbed.npz")
    cfg = json.loads(str(g["cfg_json"]))
    o = CPOOracle(cpo_cfg(cfg))
    o.set_params(g["theta0"])
    pb = o.process(_data(g))
    o.set_params(g["theta0_perturbed"])     # fresh Adam state, like the generator (unused so far)
    for _ in range(cfg["optim_critic_iters"]):
        o.critics_step(pb)
    st, hg, hb = o.policy_step(pb, cfg["cost_stat"])
    keys = [str(k) for k in g["pl_stats_keys"]]
    np.testing.assert_allclose([st[k] for k in keys], g["pl_stats"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(hg.numpy(), g["pl_H_inv_g"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(hb.numpy(), g["pl_H_inv_b"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(o.get_params(), g["theta_after_pl"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", ["small", "c1", "options", "minibatch", "widths", "deep3"])
def test_trpo_update(name):
    torch.set_num_threads(4)
    g = load_npz(f"trpo_{name}.npz")
    cfg = json.loads(str(g["cfg_json"]))
    o = TRPOLagOracle(trpo_cfg(cfg))
    o.set_params(g["theta0"])
    if "ret_rms0" in g:
        o.ret_rms[:] = g["ret_rms0"]
    lag = g["lagrangian"]
    pb, rows = o.update(_data(g), lag, 1.0 / (lag.sum() + 1.0), cfg["repeat"], perms=g["perms"],
                        batch_size=cfg.get("batch_size", 99999))
    keys = [str(k) for k in g["stats_keys"]]
    got = np.array([[r[0][k] for k in keys] for r in rows])
    np.testing.assert_allclose(got, g["stats"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(rows[0][1].numpy(), g["cg_first"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(o.get_params(), g["theta_final"], rtol=0, atol=2e-6)
    assert o.gradient_steps == int(g["gradient_steps"])
