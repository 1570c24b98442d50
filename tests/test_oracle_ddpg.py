"""Oracle DDPG-Lagrangian vs golden vectors recorded from the unmodified reference."""
import json

import numpy as np
import pytest
import torch

from helpers import load_npz
from oracle.ddpg_lag import DDPGConfig, DDPGLagOracle
from oracle.sac_lag import ReplayIndex


def ddpg_setup(name):
    g = load_npz(f"ddpg_{name}.npz")
    cfg = json.loads(str(g["cfg_json"]))
    ocfg = DDPGConfig(obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"], hidden=tuple(cfg["hidden"]),
                      max_action=cfg["max_action"], gamma=cfg["gamma"], n_step=cfg["n_step"], tau=cfg["tau"],
                      actor_lr=cfg["actor_lr"], critic_lr=cfg["critic_lr"], use_lagrangian=cfg["use_lagrangian"])
    sub = int(g["sub_size"])
    store = {}
    for k in ("obs", "act", "rew", "cost", "terminated", "truncated", "obs_next"):
        src = g["st_" + k]
        full = np.zeros((sub * cfg["env_num"], ) + src.shape[1:], src.dtype)
        full[g["slots"]] = src
        store[k] = full
    return g, cfg, ocfg, store, ReplayIndex(g["env_rows"], sub, store["terminated"] | store["truncated"])


@pytest.mark.parametrize("name", ["small", "scaled", "nolag", "deep3"])
def test_ddpg_updates(name):
    torch.set_num_threads(4)
    g, cfg, ocfg, store, index = ddpg_setup(name)
    o = DDPGLagOracle(ocfg)
    o.set_params(g["theta_actor0"], g["theta_critics0"])
    lag = g["lagrangian"] if cfg["use_lagrangian"] else np.zeros(0)
    resc = 1.0 / (lag.sum() + 1.0)
    ka = [str(k) for k in g["stats_actor_keys"]]; kc = [str(k) for k in g["stats_critic_keys"]]
    for u in range(cfg["n_updates"]):
        sa, sc, _ = o.update(store, index, g["indices"][u], lag, resc)
        np.testing.assert_allclose([sa[k] for k in ka], g["stats_actor"][u], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose([sc[k] for k in kc], g["stats_critic"][u], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(o.actor_flat(), g["theta_actor_final"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(o.actor_flat(old=True), g["theta_actor_old_final"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(o.critics_flat(), g["theta_critics_final"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(o.critics_flat(old=True), g["theta_critics_old_final"], rtol=0, atol=2e-6)
