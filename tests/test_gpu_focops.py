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
    eng = Engine(EngineConfig(algo=_lib.ALGO_FOCOPS, obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"], hidden_sizes=tuple(cfg["hidden"]),
                              n_critics=2, env_num=cfg["env_num"], max_action=cfg["max_action"], gamma=cfg["gamma"],
                              gae_lambda=cfg["gae_lambda"], norm_adv=cfg["advantage_normalization"], target_kl=None,
                              unbounded=bool(cfg.get("unbounded", False)),
                              recompute_adv=bool(cfg.get("recompute_advantage", False))))
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


@pytest.mark.parametrize("four_launch", [0, 1])     # 0: three launches per step (ppo_wgrad_kernel), 1: split-K weight gradients + their sum
@pytest.mark.parametrize("name", ["small", "c1", "earlystop", "unbounded", "recompute", "deep3", "wide1"])
def test_focops_update_vs_golden(name, four_launch):
    g = load_npz(f"focops_{name}.npz")
    cfg = json.loads(str(g["cfg_json"]))
    eng = _engine(cfg, g)
    eng.focops_set_plan(four_launch)
    nu = float(g["stats_nu"][0][1]); nu_loss = float(g["stats_nu"][0][0])       # the host-side nu step (focops.py:154-159)
    perms = list(g["perms"]) + [np.arange(len(g["indices"]))] * (cfg["repeat"] - len(g["perms"]))
    stats, stopped = eng.focops_update(nu, nu_loss, cfg["batch_size"], cfg["repeat"], perms=perms)
    for k in ("logp_old", ) if cfg.get("recompute_advantage") else ("rets", "advs", "logp_old"):   # recompute overwrites rets / advs
        scale = max(1.0, float(np.abs(g[k]).max()))
        np.testing.assert_allclose(eng.batch_get(k), g[k], rtol=0, atol=5e-6 * scale, err_msg=k)
    want = np.concatenate([g["stats_nu"], g["stats_actor"], g["stats_critic"]], 1)
    assert stats.shape == want.shape, (stats.shape, want.shape)
    assert (stopped >= 0) == (len(g["perms"]) < cfg["repeat"])
    np.testing.assert_allclose(stats, want, rtol=3e-5, atol=3e-5)
    d = np.abs(eng.get_params() - g["theta_final"])
    assert np.quantile(d, 0.999) <= 5e-6 and d.max() <= 2e-3, (np.quantile(d, 0.999), d.max())
    eng.close()


def test_focops_facade_matches_reference_and_agent_learns(tmp_path):
    import random
    import torch
    from torch.distributions import Independent, Normal
    from fsrl_amd.agent import FOCOPSAgent
    from fsrl_amd.data import Batch, HipVectorReplayBuffer
    from fsrl_amd.env import Box, SyntheticSafetyVectorEnv
    from fsrl_amd.policy import FOCOPS
    from fsrl_amd.utils import BaseLogger
    from fsrl_amd.utils.net import ActorProb, Critic, Net
    g = load_npz("focops_small.npz"); cfg = json.loads(str(g["cfg_json"]))
    Do, Da, h = cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"])
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), max_action=1.0)
    critics = [Critic(Net((Do, ), hidden_sizes=h)) for _ in range(2)]

    class Cap:
        def __init__(self): self.rows = []
        def store(self, tab=None, **kw): self.rows.append(dict(kw))
        def print(self, *a, **k): pass
    log = Cap()
    pol = FOCOPS(actor, critics, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                 torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=cfg["critic_lr"]),
                 lambda *l: Independent(Normal(*l), 1), logger=log, cost_limit=cfg["cost_limit"],
                 nu=(cfg["nu_max"], cfg["nu_lr"], torch.zeros(1) + cfg["nu"]), observation_space=Box(-np.inf, np.inf, (Do, )),
                 action_space=Box(-1, 1, (Da, )), device=0, env_num=cfg["env_num"])
    pol.engine.set_params(g["theta0"]); pol._pull_params()
    pol.train()
    buf = HipVectorReplayBuffer(pol.engine, 100000, cfg["env_num"])
    rows = g["env_rows"]; off = np.concatenate([[0], np.cumsum(rows)])
    for t in range(rows.max()):
        ids = np.array([e for e in range(len(rows)) if t < rows[e]])
        sel = np.array([off[e] + t for e in ids])
        buf.add(Batch(obs=g["buf_obs"][sel], act=g["buf_act"][sel], rew=g["buf_rew"][sel], info={"cost": g["buf_cost"][sel]},
                      terminated=g["buf_terminated"][sel], truncated=g["buf_truncated"][sel], obs_next=g["buf_obs_next"][sel]),
                buffer_ids=ids)
    pol.pre_update_fn(stats_train={"cost": cfg["cost_stat"]})
    orig = np.random.permutation
    it = iter(g["perms"])
    np.random.permutation = lambda n: next(it)                 # the reference's recorded Batch.split shuffles
    try:
        pol.update(0, buf, batch_size=cfg["batch_size"], repeat=cfg["repeat"])
    finally:
        np.random.permutation = orig
    rows_l = [r for r in log.rows if "gradient_steps" not in r]
    kn, ka, kc = ([str(k) for k in g[f"stats_{w}_keys"]] for w in ("nu", "actor", "critic"))
    for i in range(len(g["stats_actor"])):
        np.testing.assert_allclose([rows_l[3 * i][k] for k in kn], g["stats_nu"][i], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose([rows_l[3 * i + 1][k] for k in ka], g["stats_actor"][i], rtol=3e-5, atol=3e-5)
        np.testing.assert_allclose([rows_l[3 * i + 2][k] for k in kc], g["stats_critic"][i], rtol=3e-5, atol=3e-5)
    assert abs(float(pol._nu) - float(g["nu_final"])) < 1e-6
    env = SyntheticSafetyVectorEnv(env_num=4, episode_len=30, seed=2)
    agent = FOCOPSAgent(env, BaseLogger(str(tmp_path), name="f"), cost_limit=10, device="cuda:0", seed=1,
                        hidden_sizes=(64, 64), training_num=4)
    ep, stat, info = agent.learn(env, None, epoch=2, episode_per_collect=4, step_per_epoch=240, repeat_per_collect=2,
                                 batch_size=64, verbose=False, save_ckpt=False)
    assert ep == 2 and np.isfinite(list(stat.values())).all() and "loss/nu_value" in stat
