"""GPU parity of FOCOPS through the C ABI against the golden vectors recorded from the
unmodified reference.  Tolerances: process_fn products 5e-6 * scale, per-minibatch stats 3e-5 rel + 3e-5 abs,
parameters 99.9 % within 5e-6 (Adam on noise-level gradients may move single entries by ~lr per step)."""
import json

import numpy as np
import pytest

from helpers import load_npz

pytestmark = pytest.mark.gpu


def _engine(cfg, g):
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    eng = Engine(EngineConfig(algo=_lib.ALGO_FOCOPS, obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"], hidden=cfg["hidden"][0],
                              n_critics=2, env_num=cfg["env_num"], max_action=cfg["max_action"], gamma=cfg["gamma"],
                              gae_lambda=cfg["gae_lambda"], norm_adv=cfg["advantage_normalization"], target_kl=None))
    eng.focops_init(actor_lr=cfg["actor_lr"], critic_lr=cfg["critic_lr"], l2_reg=cfg["l2_reg"], delta=cfg["delta"],
                    eta=cfg["eta"], tem_lambda=cfg["tem_lambda"], max_grad_norm=cfg["max_grad_norm"])
    eng.set_params(g["theta0"])
    rows = g["env_rows"]; off = np.concatenate([[0], np.cumsum(rows)])
    for t in range(rows.max()):
        ids = [e for e in range(len(rows)) if t < rows[e]]
        sel = np.array([off[e] + t for e in ids])
        eng.push(ids, g["buf_obs"][sel], g["buf_act"][sel], g["buf_rew"][sel], g["buf_cost"][sel], g["buf_terminated"][sel],
                 g["buf_truncated"][sel], g["buf_obs_next"][sel])
    return eng


@pytest.mark.parametrize("name", ["small", "c1", "earlystop"])
def test_focops_update_vs_golden(name):
    g = load_npz(f"focops_{name}.npz")
    cfg = json.loads(str(g["cfg_json"]))
    eng = _engine(cfg, g)
    nu = float(g["stats_nu"][0][1]); nu_loss = float(g["stats_nu"][0][0])       # the host-side nu step (focops.py:154-159)
    perms = list(g["perms"]) + [np.arange(len(g["indices"]))] * (cfg["repeat"] - len(g["perms"]))
    stats, stopped = eng.focops_update(nu, nu_loss, cfg["batch_size"], cfg["repeat"], perms=perms)
    for k in ("rets", "advs", "logp_old"):
        scale = max(1.0, float(np.abs(g[k]).max()))
        np.testing.assert_allclose(eng.batch_get(k), g[k], rtol=0, atol=5e-6 * scale, err_msg=k)
    want = np.concatenate([g["stats_nu"], g["stats_actor"], g["stats_critic"]], 1)
    assert stats.shape == want.shape, (stats.shape, want.shape)
    assert (stopped >= 0) == (len(g["perms"]) < cfg["repeat"])
    np.testing.assert_allclose(stats, want, rtol=3e-5, atol=3e-5)
    d = np.abs(eng.get_params() - g["theta_final"])
    assert np.quantile(d, 0.999) <= 5e-6 and d.max() <= 2e-3, (np.quantile(d, 0.999), d.max())
    eng.close()
