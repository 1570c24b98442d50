"""Oracle CVPO vs golden vectors recorded from the unmodified reference (tests/golden/gen_golden_cvpo.py)."""
import json

import numpy as np
import pytest
import torch

from helpers import load_npz
from oracle.cvpo import CVPOConfig, CVPOOracle
from oracle.sac_lag import ReplayIndex

CFG_KEYS = ("obs_dim", "act_dim", "max_action", "gamma", "n_step", "tau", "actor_lr", "critic_lr", "double_critic",
            "sample_act_num", "estep_iter_num", "estep_kl", "estep_dual_max", "estep_dual_lr", "mstep_iter_num",
            "mstep_kl_mu", "mstep_kl_std", "mstep_dual_max", "mstep_dual_lr", "cost_limit", "max_episode_steps")


def cvpo_setup(name):
    g = load_npz(f"cvpo_{name}.npz")
    cfg = json.loads(str(g["cfg_json"]))
    ocfg = CVPOConfig(hidden=tuple(cfg["hidden"]), **{k: cfg[k] for k in CFG_KEYS})
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


@pytest.mark.parametrize("name", ["small", "default", "double", "deep3", "wide1_double"])
def test_cvpo_updates(name):
    torch.set_num_threads(4)
    g, cfg, ocfg, store, index = cvpo_setup(name)
    assert abs(ocfg.qc_thres - float(g["qc_thres"][0])) < 1e-12
    o = CVPOOracle(ocfg)
    o.set_params(g["theta_actor0"], g["theta_critics0"])
    keys = [str(k) for k in g["stats_keys"]]
    u = 0
    for c in range(cfg["cycles"]):
        o.pre_update()
        for _ in range(cfg["updates_per_cycle"]):
            st, _, w = o.update(store, index, g["indices"][u], g["eps_target"][u], g["eps_particles"][u])
            np.testing.assert_allclose(w.sum(0).numpy(), 1.0, atol=1e-5)
            np.testing.assert_allclose([st[k] for k in keys], g["stats"][u], rtol=3e-5, atol=3e-6, err_msg=f"u={u}")
            np.testing.assert_allclose(o.estep_dual.detach().numpy(), g["estep_dual"][u], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose([o.mstep_dual_mu.item(), o.mstep_dual_std.item()], g["mstep_dual"][u],
                                       rtol=1e-5, atol=1e-6)
            u += 1
        o.post_update()
        np.testing.assert_allclose(o.actor_flat(old=True), g["theta_actor_old_cycles"][c], rtol=0, atol=3e-6)
    np.testing.assert_allclose(o.actor_flat(), g["theta_actor_final"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(o.critics_flat(), g["theta_critics_final"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(o.critics_flat(old=True), g["theta_critics_old_final"], rtol=0, atol=3e-6)
