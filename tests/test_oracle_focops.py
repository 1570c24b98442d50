"""Oracle FOCOPS vs golden vectors recorded from the unmodified reference."""
import json

import numpy as np
import pytest
import torch

from helpers import end_flag_of, load_npz
from oracle.focops import FOCOPSConfig, FOCOPSOracle
from test_oracle_trust import _data


def focops_cfg(cfg):
    return FOCOPSConfig(obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"], hidden=tuple(cfg["hidden"]), max_action=cfg["max_action"],
                        gamma=cfg["gamma"], gae_lambda=cfg["gae_lambda"], actor_lr=cfg["actor_lr"], critic_lr=cfg["critic_lr"],
                        l2_reg=cfg["l2_reg"], delta=cfg["delta"], eta=cfg["eta"], tem_lambda=cfg["tem_lambda"],
                        max_grad_norm=cfg["max_grad_norm"], advantage_normalization=cfg["advantage_normalization"],
                        nu_max=cfg["nu_max"], nu_lr=cfg["nu_lr"], cost_limit=cfg["cost_limit"],
                        unbounded=bool(cfg.get("unbounded", False)),
                        recompute_advantage=bool(cfg.get("recompute_advantage", False)))


@pytest.mark.parametrize("name", ["small", "c1", "earlystop", "unbounded", "recompute", "deep3", "wide1"])
def test_focops_update(name):
    torch.set_num_threads(4)
    g = load_npz(f"focops_{name}.npz")
    cfg = json.loads(str(g["cfg_json"]))
    o = FOCOPSOracle(focops_cfg(cfg))
    o.set_params(g["theta0"], nu=float(g["nu0"]))
    pb, rows, stopped = o.update(_data(g), cfg["cost_stat"], cfg["batch_size"], cfg["repeat"], list(g["perms"]) + [None] * 8)
    for k in ("advs", "rets", "logp_old", "mean_old", "std_old"):
        np.testing.assert_array_equal(pb[k].numpy(), g[k], err_msg=k)
    kn, ka, kc = ([str(k) for k in g[f"stats_{w}_keys"]] for w in ("nu", "actor", "critic"))
    assert len(rows) == len(g["stats_actor"]) and (stopped >= 0) == (len(g["perms"]) < cfg["repeat"])
    for i, (sn, sa, sc) in enumerate(rows):
        np.testing.assert_allclose([sn[k] for k in kn], g["stats_nu"][i], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose([sa[k] for k in ka], g["stats_actor"][i], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose([sc[k] for k in kc], g["stats_critic"][i], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(o.get_params(), g["theta_final"], rtol=0, atol=2e-6)
    assert abs(o.nu - float(g["nu_final"])) < 1e-6
