"""GPU parity of the DDPG-Lagrangian update through the C ABI against the golden vectors
recorded from the unmodified reference.  Tolerances as for SAC: stats 5e-5 rel + 5e-6 abs, parameters
99 % within 5e-6 (Adam on noise-level gradients may move single entries by ~lr per step)."""
import numpy as np
import pytest

from test_oracle_ddpg import ddpg_setup

pytestmark = pytest.mark.gpu

KEYS = ["loss/rescaling", "loss/lagrangian", "loss/actor_safety", "loss/alpha_loss", "loss/alpha_value",
        "loss/actor_rew", "loss/actor_total", "loss/q0", "loss/q1", "loss/q_total"]


def _engine(cfg, g):
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    eng = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"],
                              hidden_sizes=tuple(cfg["hidden"]), n_critics=2, env_num=cfg["env_num"], buffer_size=cfg["buffer_size"],
                              gamma=cfg["gamma"], max_action=cfg["max_action"], target_kl=None))
    eng.sac_init(actor_lr=cfg["actor_lr"], critic_lr=cfg["critic_lr"], tau=cfg["tau"], n_step=cfg["n_step"],
                 use_lagrangian=cfg["use_lagrangian"], deterministic=True)
    eng.sac_set_params(g["theta_actor0"], g["theta_critics0"], 0.0)
    rows = g["env_rows"]; off = np.concatenate([[0], np.cumsum(rows)])
    for t in range(rows.max()):
        ids = [e for e in range(len(rows)) if t < rows[e]]
        sel = np.array([off[e] + t for e in ids])
        eng.push(ids, g["st_obs"][sel], g["st_act"][sel], g["st_rew"][sel], g["st_cost"][sel], g["st_terminated"][sel],
                 g["st_truncated"][sel], g["st_obs_next"][sel])
    return eng


@pytest.mark.parametrize("name", ["small", "scaled", "nolag", "deep3"])
def test_ddpg_updates_vs_golden(name):
    g, cfg, ocfg, store, index = ddpg_setup(name)
    eng = _engine(cfg, g)
    assert np.array_equal(eng.sac_get_params(0)[0], g["theta_actor0"]) and np.array_equal(eng.sac_get_params(3)[0], g["theta_actor0"])
    assert np.array_equal(eng.sac_get_params(1)[0], g["theta_critics0"])
    lag = g["lagrangian"] if cfg["use_lagrangian"] else np.zeros(0)
    resc = 1.0 / (lag.sum() + 1.0)
    ka = [str(k) for k in g["stats_actor_keys"]]; kc = [str(k) for k in g["stats_critic_keys"]]
    B, Da = cfg["batch_size"], cfg["act_dim"]
    zero = np.zeros((B, Da), np.float32)
    for u in range(cfg["n_updates"]):
        st = eng.sac_update(B, lag, resc, indices=g["indices"][u], eps_target=zero, eps_pi=zero)
        want = {**dict(zip(ka, g["stats_actor"][u])), **dict(zip(kc, g["stats_critic"][u]))}
        for j, k in enumerate(KEYS):
            if k in want:
                assert abs(st[j] - want[k]) <= 5e-5 * abs(want[k]) + 5e-6, (u, k, st[j], want[k])
    for which, key in ((0, "theta_actor_final"), (3, "theta_actor_old_final"), (1, "theta_critics_final"),
                       (2, "theta_critics_old_final")):
        d = np.abs(eng.sac_get_params(which)[0] - g[key])
        assert np.quantile(d, 0.99) <= 5e-6 and d.max() <= 2e-3, (key, np.quantile(d, 0.99), d.max())
    # collector-time action: deterministic = max_action * tanh(actor(s)); exploration adds N(0, 0.1^2)
    obs = g["st_obs"][:5]
    a_det = eng.actor_sample(obs, deterministic=True)
    assert np.abs(a_det).max() <= cfg["max_action"] + 1e-6
    a = np.stack([eng.actor_sample(obs, seed=3 if i == 0 else 0) for i in range(400)])
    assert abs((a - a_det).std() - 0.1) < 0.01
    eng.close()


def test_ddpg_facade_matches_reference_and_agent_learns(tmp_path):
    import json
    from fsrl_amd.agent import DDPGLagAgent
    from fsrl_amd.data import Batch, HipVectorReplayBuffer
    from fsrl_amd.env import Box, SyntheticSafetyVectorEnv
    from fsrl_amd.policy import DDPGLagrangian, SACLagrangian
    from fsrl_amd.utils import BaseLogger
    from fsrl_amd.utils.net import Actor, Critic, Net
    import torch
    g, cfg, ocfg, store, index = ddpg_setup("small")
    Do, Da, h = cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"])
    actor = Actor(Net((Do, ), hidden_sizes=h), (Da, ), max_action=cfg["max_action"])
    critics = [Critic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True)) for _ in range(2)]
    SACLagrangian._unflat([actor], g["theta_actor0"]); SACLagrangian._unflat(critics, g["theta_critics0"])

    class Cap:
        def __init__(self): self.rows = []
        def store(self, tab=None, **kw): self.rows.append(dict(kw))
        def print(self, *a, **k): pass
    log = Cap()
    pol = DDPGLagrangian(actor, critics, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                         torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=cfg["critic_lr"]), logger=log,
                         tau=cfg["tau"], n_step=cfg["n_step"], cost_limit=cfg["cost_limit"], gamma=cfg["gamma"],
                         observation_space=Box(-np.inf, np.inf, (Do, )), action_space=Box(-1, 1, (Da, )), device=0,
                         env_num=cfg["env_num"], reference_rng=True)
    pol.train()
    buf = HipVectorReplayBuffer(pol.engine, cfg["buffer_size"], cfg["env_num"])
    rows = g["env_rows"]; off = np.concatenate([[0], np.cumsum(rows)])
    for t in range(rows.max()):
        ids = np.array([e for e in range(len(rows)) if t < rows[e]])
        sel = np.array([off[e] + t for e in ids])
        buf.add(Batch(obs=g["st_obs"][sel], act=g["st_act"][sel], rew=g["st_rew"][sel], info={"cost": g["st_cost"][sel]},
                      terminated=g["st_terminated"][sel], truncated=g["st_truncated"][sel], obs_next=g["st_obs_next"][sel]),
                buffer_ids=ids)
    pol.pre_update_fn(stats_train={"cost": cfg["cost_stat"]})
    import random
    seed = cfg["seed"] + 7
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    ka = [str(k) for k in g["stats_actor_keys"]]; kc = [str(k) for k in g["stats_critic_keys"]]
    for u in range(cfg["n_updates"]):
        pol.update(cfg["batch_size"], buf)
        np.testing.assert_allclose([log.rows[2 * u][k] for k in ka], g["stats_actor"][u], rtol=5e-5, atol=5e-6)
        np.testing.assert_allclose([log.rows[2 * u + 1][k] for k in kc], g["stats_critic"][u], rtol=5e-5, atol=5e-6)
    sd = pol.state_dict()
    assert "actor_old.last.model.0.weight" in sd and "critics_old.1.preprocess.model.model.0.weight" in sd
    d = np.abs(SACLagrangian._flat([pol.actor_old]) - g["theta_actor_old_final"])
    assert np.quantile(d, 0.99) <= 5e-6
    env = SyntheticSafetyVectorEnv(env_num=4, episode_len=30, seed=2)
    agent = DDPGLagAgent(env, BaseLogger(str(tmp_path), name="g"), cost_limit=10, device="cuda:0", seed=1,
                         hidden_sizes=(64, 64), training_num=4, buffer_size=2000)
    ep, stat, info = agent.learn(env, None, epoch=2, episode_per_collect=4, step_per_epoch=240, update_per_step=0.2,
                                 batch_size=32, verbose=False, save_ckpt=False)
    assert ep == 2 and np.isfinite(list(stat.values())).all() and "loss/q_total" in stat
