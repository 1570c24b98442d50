"""Oracle SAC-Lagrangian vs golden vectors recorded from the unmodified reference."""
import json

import numpy as np
import pytest
import torch

from helpers import load_npz, sac_full_case
from oracle.sac_lag import ReplayIndex, SACConfig, SACLagOracle


def sac_setup(name):
    # c4full: BASELINE configs[3]'s shape at batch 1024 over a 97 000-row store (rollout + parameters regenerated from seeds)
    g = sac_full_case(name) if name.endswith("full") else load_npz(f"sac_{name}.npz")
    cfg = json.loads(str(g["cfg_json"]))
    ocfg = SACConfig(obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"], hidden=tuple(cfg["hidden"]),
                     gamma=cfg["gamma"], n_step=cfg["n_step"], tau=cfg["tau"], alpha=cfg["alpha"],
                     auto_alpha=cfg["auto_alpha"], actor_lr=cfg["actor_lr"], critic_lr=cfg["critic_lr"],
                     alpha_lr=cfg["alpha_lr"])
    sub = int(g["sub_size"])
    nslots = sub * cfg["env_num"]
    store = {}
    for k in ("obs", "act", "rew", "cost", "terminated", "truncated", "obs_next"):
        src = g["st_" + k]
        full = np.zeros((nslots, ) + src.shape[1:], src.dtype)
        full[g["slots"]] = src
        store[k] = full
    done = store["terminated"] | store["truncated"]
    return g, cfg, ocfg, store, ReplayIndex(g["env_rows"], sub, done)


def old_final(g, flat_old):
    """the full-size fixture keeps every 8th element of the target critics"""
    return flat_old[::8] if flat_old.size != g["theta_critics_old_final"].size else flat_old


@pytest.mark.parametrize("name", ["small", "nstep3", "c4", "widths", "deep3", "wide1", "c4full"])
def test_sac_updates(name):
    torch.set_num_threads(4)
    g, cfg, ocfg, store, index = sac_setup(name)
    o = SACLagOracle(ocfg)
    o.set_params(g["theta_actor0"], g["theta_critics0"])
    lag = g["lagrangian"]
    resc = 1.0 / (lag.sum() + 1.0)
    ka = [str(k) for k in g["stats_actor_keys"]]
    kc = [str(k) for k in g["stats_critic_keys"]]
    for u in range(cfg["n_updates"]):
        sa, sc, _ = o.update(store, index, g["indices"][u], g["eps_target"][u], g["eps_pi"][u], lag, resc)
        np.testing.assert_allclose([sa[k] for k in ka], g["stats_actor"][u], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose([sc[k] for k in kc], g["stats_critic"][u], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(o.actor_flat(), g["theta_actor_final"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(o.critics_flat(), g["theta_critics_final"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(old_final(g, o.critics_flat(old=True)), g["theta_critics_old_final"], rtol=0, atol=2e-6)
    assert abs(float(o.alpha) - float(g["alpha_final"])) < 1e-6
