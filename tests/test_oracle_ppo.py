"""Oracle PPO-Lagrangian update vs golden vectors recorded from the unmodified reference.

Same torch CPU ops in the same order => agreement to rounding of BLAS blocking only.
Tolerances: per-step stats 2e-5 abs+rel, parameters 2e-6 abs (fp32)."""
import numpy as np
import pytest
import torch

from helpers import oracle_cfg_and_data, ppo_case, ppo_full_case, rollout_env_major
from oracle.pid import rescaling_factor
from oracle.ppo_lag import PPOLagOracle, split_chunks

CASES = ["tiny", "c1", "c2", "bigbatch", "earlystop", "dualclip", "recompute", "rewnorm", "rewnorm_first", "rewnorm_recompute",
         "unbounded", "widths", "widths_wide",
         "deep3", "wide", "one_layer", "deep4_options"]     # hidden_sizes of other depths / widths above 256


def test_split_chunks_matches_tianshou_semantics():
    # 292 rows, size 64: 3 full chunks then 64+36 merged
    ch = split_chunks(292, 64, np.arange(292))
    assert [len(c) for c in ch] == [64, 64, 64, 100]
    assert [len(c) for c in split_chunks(256, 64, None)] == [64] * 4
    assert [len(c) for c in split_chunks(10, 64, None)] == [10]
    assert [len(c) for c in split_chunks(100, 64, None, merge_last=False)] == [64, 36]


@pytest.mark.parametrize("name", CASES)
def test_process_fn(name):
    torch.set_num_threads(4)
    cfg, g = ppo_case(name)
    ocfg, data = oracle_cfg_and_data(cfg, g)
    o = PPOLagOracle(ocfg)
    o.set_params(g["theta0"])
    o.ret_rms[:] = g["ret_rms0"]
    pb = o.process(data)
    for k in ("values", "rets", "advs", "logp_old"):
        np.testing.assert_allclose(pb[k].numpy(), g[k], rtol=1e-5, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("name", CASES)
def test_full_update(name):
    torch.set_num_threads(4)
    cfg, g = ppo_case(name)
    ocfg, data = oracle_cfg_and_data(cfg, g)
    o = PPOLagOracle(ocfg)
    o.set_params(g["theta0"])
    o.ret_rms[:] = g["ret_rms0"]
    lag = g["lagrangian"]
    pb, stats, stopped = o.update(data, lag, rescaling_factor(lag), cfg["batch_size"],
                                  cfg["repeat"], perms=g["perms"])
    np.testing.assert_allclose(o.ret_rms, g["ret_rms_final"], rtol=1e-12, atol=0)      # float64 host arithmetic both sides
    assert stats.shape == g["stats"].shape
    assert o.gradient_steps == int(g["gradient_steps"])
    assert (stopped >= 0) == bool(g["early_stop_msgs"])
    np.testing.assert_allclose(stats, g["stats"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(o.get_params(), g["theta_final"], rtol=0, atol=2e-6)


def test_two_updates_under_an_lr_schedule():
    """ppo_lrsched.npz (tests/golden/gen_golden_lr.py): the reference steps LambdaLR(0.5 ** epoch) at the end of every
    update() (base_policy.py:352-354) -- update 0 at lr, update 1 at lr / 2 with the Adam state carried over."""
    torch.set_num_threads(4)
    cfg, g = ppo_case("lrsched")
    ocfg, data = oracle_cfg_and_data(cfg, g)
    o = PPOLagOracle(ocfg)
    o.set_params(g["theta0"])
    lag = g["lagrangian"]
    R = cfg["repeat"]
    for u in range(2):
        for pg in o.optim.param_groups:
            pg["lr"] = float(g["lrs"][u])
        _, stats, _ = o.update(data, lag, rescaling_factor(lag), cfg["batch_size"], R, perms=g["perms"][u * R:(u + 1) * R])
        np.testing.assert_allclose(stats, g[f"stats{u}"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(o.get_params(), g[f"theta_after{u}"], rtol=0, atol=2e-6)
    assert g["lrs"][1] == 0.5 * g["lrs"][0] and g["lrs"][2] == 0.25 * g["lrs"][0]


@pytest.mark.parametrize("name", ["c2full", "c5rank", "c0full"])
def test_full_size_update_vs_reference(name):
    """The headline workload itself (BASELINE configs[1] / one rank of configs[4]: obs 8, act 2, 256x256, N = 20 000, batch 256,
    4 passes = 312 steps, max_grad_norm 0.5 as in ppol_cfg.py:21; c0full: configs[0] at its full size, 4 envs x 5000 rows, 128x128)
    recorded from the unmodified reference (ppo_lag.py:214-257):
    process_fn, every logged statistic of the first pass at the fixture tolerances, theta after pass 1 and after pass 4."""
    from oracle.ppo_lag import OnPolicyData, PPOLagConfig
    torch.set_num_threads(4)
    cfg, g, steps = ppo_full_case(name)
    d = rollout_env_major(steps, cfg["env_num"])
    assert len(d["obs"]) == int(g["n_rows"]) == 20000
    ocfg = PPOLagConfig(obs_dim=8, act_dim=2, hidden=tuple(cfg["hidden"]), max_grad_norm=0.5, target_kl=1e9)
    o = PPOLagOracle(ocfg)
    o.set_params(g["theta0"])
    lag = g["lagrangian"]
    pb, stats, _ = o.update(OnPolicyData(**d), lag, rescaling_factor(lag), 256, 4, perms=g["perms"])
    np.testing.assert_allclose(pb["advs"].numpy(), g["advs"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(pb["logp_old"].numpy(), g["logp_old"], rtol=1e-5, atol=1e-5)
    assert stats.shape == g["stats"].shape == (312, 11)
    np.testing.assert_allclose(stats, g["stats"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(o.get_params(), g["theta_final"], rtol=0, atol=2e-6)


def test_full_size_kl_early_stop_vs_reference():
    """BASELINE configs[1] with the KL early stop ON at full size (target_kl 0.02 = the reference default, ppo_lag_agent.py:95; lr
    1.5e-4): the pass-mean KL stays below 1.5 x target_kl in pass 1 (0.0262) and crosses it in pass 2 (0.0366), so the
    unmodified reference runs 2 of its 4 passes (ppo_lag.py:251-255).  Fixture: tests/golden/gen_golden.py full_klstop."""
    from oracle.ppo_lag import OnPolicyData, PPOLagConfig
    torch.set_num_threads(4)
    cfg, g, steps = ppo_full_case("c2full_klstop")
    assert int(g["passes_run"]) == 2 and int(g["early_stop_msgs"]) == 1 and cfg["target_kl"] == 0.02
    o = PPOLagOracle(PPOLagConfig(obs_dim=8, act_dim=2, hidden=(256, 256), max_grad_norm=0.5, target_kl=cfg["target_kl"], lr=cfg["lr"]))
    o.set_params(g["theta0"])
    lag = g["lagrangian"]
    perms = list(g["perms"]) + [np.arange(20000)] * 2                  # passes 3, 4 must never be drawn on
    _, stats, stopped = o.update(OnPolicyData(**rollout_env_major(steps, cfg["env_num"])), lag, rescaling_factor(lag), 256, 4, perms=perms)
    assert stopped == 1 and stats.shape == g["stats"].shape == (156, 11)
    np.testing.assert_allclose(stats, g["stats"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(o.get_params(), g["theta_final"], rtol=0, atol=2e-6)
