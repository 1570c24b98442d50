"""GPU parity of the CVPO update through the C ABI against the golden vectors recorded from the
unmodified reference (tests/golden/gen_golden_cvpo.py).  Tolerances: logged stats 1e-4 rel + 1e-5 abs (the
E-step's logsumexp / softmax over K particles and the weighted likelihood sum in a different fp32 order than
torch's), duals 1e-4, parameters 99 % within 1e-5 (Adam on noise-level gradients may move single entries by
~lr per step)."""
import numpy as np
import pytest

from test_oracle_cvpo import cvpo_setup

pytestmark = pytest.mark.gpu


def _engine(cfg, g, ocfg):
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    eng = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"],
                              hidden=cfg["hidden"][0], n_critics=2, env_num=cfg["env_num"], buffer_size=cfg["buffer_size"],
                              gamma=cfg["gamma"], max_action=cfg["max_action"], target_kl=None))
    eng.cvpo_init(ocfg.qc_thres, **{k: cfg[k] for k in (
        "actor_lr", "critic_lr", "tau", "n_step", "double_critic", "sample_act_num", "estep_iter_num", "mstep_iter_num",
        "estep_kl", "estep_dual_max", "estep_dual_lr", "mstep_kl_mu", "mstep_kl_std", "mstep_dual_max", "mstep_dual_lr")})
    eng.sac_set_params(g["theta_actor0"], g["theta_critics0"], 0.0)
    rows = g["env_rows"]; off = np.concatenate([[0], np.cumsum(rows)])
    for t in range(rows.max()):
        ids = [e for e in range(len(rows)) if t < rows[e]]
        sel = np.array([off[e] + t for e in ids])
        eng.push(ids, g["st_obs"][sel], g["st_act"][sel], g["st_rew"][sel], g["st_cost"][sel], g["st_terminated"][sel],
                 g["st_truncated"][sel], g["st_obs_next"][sel])
    return eng


def _close(d, q99, mx):
    return np.quantile(d, 0.99) <= q99 and d.max() <= mx


@pytest.mark.parametrize("name", ["small", "default", "double"])
def test_cvpo_updates_vs_golden(name):
    g, cfg, ocfg, store, index = cvpo_setup(name)
    eng = _engine(cfg, g, ocfg)
    assert np.array_equal(eng.sac_get_params(0)[0], g["theta_actor0"]) and np.array_equal(eng.sac_get_params(3)[0], g["theta_actor0"])
    assert np.array_equal(eng.sac_get_params(1)[0], g["theta_critics0"]) and np.array_equal(eng.sac_get_params(2)[0], g["theta_critics0"])
    keys = [str(k) for k in g["stats_keys"]]
    B = cfg["batch_size"]
    u = 0
    for c in range(cfg["cycles"]):
        eng.cvpo_pre_update()
        for _ in range(cfg["updates_per_cycle"]):
            st = eng.cvpo_update(B, indices=g["indices"][u], eps_target=g["eps_target"][u], eps_particles=g["eps_particles"][u])
            want = g["stats"][u]
            for j, k in enumerate(keys):
                assert abs(st[j] - want[j]) <= 1e-4 * abs(want[j]) + 1e-5, (u, k, st[j], want[j])
            d = eng.cvpo_duals()
            np.testing.assert_allclose(d[:2], g["estep_dual"][u], rtol=1e-4, atol=1e-6, err_msg=f"u={u}")
            np.testing.assert_allclose(d[2:], g["mstep_dual"][u], rtol=1e-4, atol=1e-5, err_msg=f"u={u}")
            u += 1
        eng.cvpo_post_update()
        dd = np.abs(eng.sac_get_params(3)[0] - g["theta_actor_old_cycles"][c])
        assert _close(dd, 1e-5, 4e-3), (c, np.quantile(dd, 0.99), dd.max())
    for which, key in ((0, "theta_actor_final"), (1, "theta_critics_final"), (2, "theta_critics_old_final")):
        dd = np.abs(eng.sac_get_params(which)[0] - g[key])
        assert _close(dd, 1e-5, 4e-3), (key, np.quantile(dd, 0.99), dd.max())
    # collector-time action: mu = max_action * tanh(head) (deterministic eval), samples are NOT squashed
    obs = g["st_obs"][:6]
    mu, sigma = eng.sac_actor_forward(obs)
    assert np.abs(mu).max() <= cfg["max_action"] + 1e-6 and (sigma > 0).all()
    assert np.allclose(eng.actor_sample(obs, deterministic=True), mu, atol=1e-6)
    a = np.stack([eng.actor_sample(obs, seed=5 if i == 0 else 0) for i in range(600)])
    assert np.abs(a.mean(0) - mu).max() < 0.2 * sigma.max() + 0.02
    assert np.abs(a.std(0) / sigma - 1.0).max() < 0.15
    eng.close()


def test_cvpo_device_rng_replays_through_caller_rng():
    """Library-RNG mode (Philox indices / target noise / particles on the device) = the same update fed back
    through the caller-RNG arguments."""
    g, cfg, ocfg, store, index = cvpo_setup("small")
    B = cfg["batch_size"]
    a, b = _engine(cfg, g, ocfg), _engine(cfg, g, ocfg)
    for eng in (a, b):
        eng.cvpo_pre_update()
    for u in range(3):
        st_a = a.cvpo_update(B, seed=11 if u == 0 else 0)
        idx, et, _ = a.sac_last_sample(B)
        ek = a.cvpo_last_particles(B)
        assert idx.min() >= 0 and len(np.unique(idx)) > B // 2
        assert abs(ek.mean()) < 0.05 and abs(ek.std() - 1.0) < 0.05 and abs(et.std() - 1.0) < 0.2
        st_b = b.cvpo_update(B, indices=idx, eps_target=et, eps_particles=ek)
        np.testing.assert_array_equal(st_a, st_b)
    for which in (0, 1, 2):
        np.testing.assert_array_equal(a.sac_get_params(which)[0], b.sac_get_params(which)[0])
    # asynchronous updates land in the ring with CVPO's row width
    for _ in range(4):
        a.cvpo_update(B, sync=False)
    rows = a.sac_drain()
    assert rows.shape == (4, 17) and np.isfinite(rows).all()
    with pytest.raises(Exception):
        a.sac_update(B, np.zeros(1), 1.0)
    a.close(); b.close()


def test_cvpo_facade_matches_reference_and_agent_learns(tmp_path):
    """The façade with reference_rng=True draws the numpy / torch streams exactly as the reference does
    (cvpo.py:208, 331, 334, 382), so the golden run is reproduced from the SEED alone."""
    import random
    import torch
    from torch.distributions import Independent, Normal
    from fsrl_amd.agent import CVPOAgent
    from fsrl_amd.data import Batch, HipVectorReplayBuffer
    from fsrl_amd.env import Box, SyntheticSafetyVectorEnv
    from fsrl_amd.policy import CVPO, SACLagrangian
    from fsrl_amd.utils import BaseLogger
    from fsrl_amd.utils.net import ActorProb, Net, SingleCritic
    g, cfg, ocfg, store, index = cvpo_setup("small")
    Do, Da, h = cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"])
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), max_action=cfg["max_action"], conditioned_sigma=True, unbounded=False)
    critics = [SingleCritic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True)) for _ in range(2)]
    SACLagrangian._unflat([actor], g["theta_actor0"]); SACLagrangian._unflat(critics, g["theta_critics0"])

    class Cap:
        def __init__(self): self.rows = {}
        def store(self, tab=None, **kw):
            for k, v in kw.items():
                self.rows.setdefault((tab + "/" + k) if tab else k, []).append(float(v))
        def print(self, *a, **k): pass
    log = Cap()
    pol = CVPO(actor, critics, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
               torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=cfg["critic_lr"]),
               action_space=Box(-cfg["max_action"], cfg["max_action"], (Da, )),
               dist_fn=lambda *l: Independent(Normal(*l), 1), max_episode_steps=cfg["max_episode_steps"], logger=log,
               cost_limit=cfg["cost_limit"], tau=cfg["tau"], gamma=cfg["gamma"], n_step=cfg["n_step"],
               mstep_kl_mu=cfg["mstep_kl_mu"], mstep_kl_std=cfg["mstep_kl_std"], observation_space=Box(-np.inf, np.inf, (Do, )),
               device=0, env_num=cfg["env_num"], reference_rng=True)
    pol.train()
    assert abs(pol.qc_thres[0] - float(g["qc_thres"][0])) < 1e-12
    buf = HipVectorReplayBuffer(pol.engine, cfg["buffer_size"], cfg["env_num"])
    rows = g["env_rows"]; off = np.concatenate([[0], np.cumsum(rows)])
    for t in range(rows.max()):
        ids = np.array([e for e in range(len(rows)) if t < rows[e]])
        sel = np.array([off[e] + t for e in ids])
        buf.add(Batch(obs=g["st_obs"][sel], act=g["st_act"][sel], rew=g["st_rew"][sel], info={"cost": g["st_cost"][sel]},
                      terminated=g["st_terminated"][sel], truncated=g["st_truncated"][sel], obs_next=g["st_obs_next"][sel]),
                buffer_ids=ids)
    seed = cfg["seed"] + 7
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    keys = [str(k) for k in g["stats_keys"]]
    u = 0
    for c in range(cfg["cycles"]):
        pol.pre_update_fn(stats_train={"cost": 0.0})
        for _ in range(cfg["updates_per_cycle"]):
            pol.update(cfg["batch_size"], buf)
            np.testing.assert_allclose([log.rows[k][u] for k in keys], g["stats"][u], rtol=1e-4, atol=1e-5, err_msg=f"u={u}")
            u += 1
        pol.post_update_fn(stats_train={"cost": 0.0})
    sd = pol.state_dict()
    assert "actor_old.mu.model.0.weight" in sd and "critics_old.1.preprocess.model.model.0.weight" in sd
    d = np.abs(SACLagrangian._flat([pol.actor_old]) - g["theta_actor_old_cycles"][-1])
    assert np.quantile(d, 0.99) <= 1e-5, np.quantile(d, 0.99)
    # state_dict round trip through a fresh policy keeps acting identically
    obs = torch.as_tensor(g["st_obs"][:7])
    pol.eval()
    a0 = pol(Batch(obs=obs)).act
    pol.load_state_dict(sd)
    assert torch.equal(pol(Batch(obs=obs)).act, a0)
    # the agent learns end to end on the synthetic env (device RNG, asynchronous updates)
    env = SyntheticSafetyVectorEnv(env_num=4, episode_len=30, seed=2)
    agent = CVPOAgent(env, BaseLogger(str(tmp_path), name="g"), cost_limit=10, device="cuda:0", seed=1,
                      hidden_sizes=(64, 64), training_num=4, buffer_size=2000)
    ep, stat, info = agent.learn(env, None, epoch=2, episode_per_collect=4, step_per_epoch=240, update_per_step=0.2,
                                 batch_size=32, verbose=False, save_ckpt=False)
    assert ep == 2 and np.isfinite(list(stat.values())).all() and "loss/q_total" in stat and "mstep/mstep_kl_mu" in stat
