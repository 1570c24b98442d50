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
                              hidden_sizes=tuple(cfg["hidden"]), n_critics=2, env_num=cfg["env_num"], buffer_size=cfg["buffer_size"],
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


@pytest.mark.parametrize("name", ["small", "default", "double", "deep3", "wide1_double"])
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


CVPO_VARIANTS = [  # Do, Da, H, rows per env, batch, K, n_step, double_critic, estep_iters, mstep_iters, max_action
    (7, 3, 64, [40, 17], 100, 16, 1, False, 1, 1, 1.0),       # batch not a multiple of 16, 1-step targets
    (20, 8, 128, [64, 64, 64], 16, 64, 3, True, 1, 2, 2.0),   # widest action head, most particles, DoubleCritic, scaled actions
    (9, 2, 256, [33], 1, 1, 2, False, 3, 1, 1.0),             # one row, one particle (weights == 1), three E-step iterations
    (41, 1, 64, [90, 45], 333, 5, 2, False, 1, 1, 0.5),       # batch larger than the store, K not a power of two
    (7, 3, (40, 24, 56), [40, 17], 100, 16, 2, False, 1, 1, 1.0),    # layered context: three ragged hidden layers
    (12, 4, (300, ), [64, 64], 50, 8, 3, True, 2, 2, 2.0),            # layered context: one wide layer, DoubleCritic, scaled actions
]


@pytest.mark.parametrize("Do,Da,H,rows,B,K,n_step,double,eit,mit,amax", CVPO_VARIANTS)
def test_cvpo_variants_vs_oracle(Do, Da, H, rows, B, K, n_step, double, eit, mit, amax):
    """Shapes and options outside the golden set, checked against the (pinned) oracle on the same random problem."""
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    from oracle.cvpo import CVPOConfig, CVPOOracle
    from oracle.sac_lag import ReplayIndex
    rng = np.random.default_rng(Do + 10 * Da)
    E, sub = len(rows), 128
    hs = (H, H) if isinstance(H, int) else tuple(H)
    ocfg = CVPOConfig(obs_dim=Do, act_dim=Da, hidden=hs, max_action=amax, gamma=0.97, n_step=n_step, tau=0.1,
                      double_critic=double, sample_act_num=K, estep_iter_num=eit, mstep_iter_num=mit, cost_limit=0.5,
                      max_episode_steps=50, mstep_kl_mu=1e-4, mstep_kl_std=1e-5, actor_lr=1e-3)
    eng = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=Do, act_dim=Da, hidden_sizes=hs, n_critics=2, env_num=E,
                              buffer_size=E * sub, gamma=0.97, max_action=amax, target_kl=None))
    eng.cvpo_init(ocfg.qc_thres, actor_lr=1e-3, tau=0.1, n_step=n_step, double_critic=double, sample_act_num=K,
                  estep_iter_num=eit, mstep_iter_num=mit, mstep_kl_mu=1e-4, mstep_kl_std=1e-5)
    o = CVPOOracle(ocfg)
    n_a = sum(int(np.prod(s)) for s in o.aspec.values()); n_c = sum(int(np.prod(s)) for s in o.cspec.values())
    tha = (0.2 * rng.standard_normal(n_a)).astype(np.float32)
    thc = (0.2 * rng.standard_normal(2 * n_c)).astype(np.float32)
    o.set_params(tha, thc); eng.sac_set_params(tha, thc, 0.0)
    store = {k: np.zeros((E * sub, ) + s, d) for k, s, d in (("obs", (Do, ), np.float32), ("obs_next", (Do, ), np.float32),
             ("act", (Da, ), np.float32), ("rew", (), np.float64), ("cost", (), np.float64),
             ("terminated", (), bool), ("truncated", (), bool))}
    for t in range(max(rows)):
        ids = [e for e in range(E) if t < rows[e]]
        k = len(ids)
        row = dict(obs=rng.standard_normal((k, Do)).astype(np.float32),
                   act=np.clip(rng.standard_normal((k, Da)), -amax, amax).astype(np.float32), rew=rng.normal(0, 1, k),
                   cost=(rng.random(k) < 0.3).astype(np.float64), terminated=rng.random(k) < 0.1,
                   truncated=np.full(k, t % 11 == 10), obs_next=rng.standard_normal((k, Do)).astype(np.float32))
        eng.push(ids, row["obs"], row["act"], row["rew"], row["cost"], row["terminated"], row["truncated"], row["obs_next"])
        for e, j in zip(ids, range(k)):
            for key in store:
                store[key][e * sub + t] = row[key][j]
    index = ReplayIndex(rows, sub, store["terminated"] | store["truncated"])
    valid = np.concatenate([e * sub + np.arange(r) for e, r in enumerate(rows)])
    keys = ["loss/estep_loss", "estep/dual0", "estep/dual1", "mstep/mstep_kl_mu", "mstep/mstep_kl_std", "mstep/mstep_loss_kl",
            "mstep/mstep_loss_mle", "mstep/mstep_loss_total", "mstep/mstep_dual_mu", "mstep/mstep_dual_std", "mstep/entropy",
            "loss/loss_q0", "estep/val_q0", "loss/loss_q1", "estep/val_q1", "estep/thres_q1", "loss/q_total"]
    for cyc in range(2):
        o.pre_update(); eng.cvpo_pre_update()
        for u in range(2):
            idx = rng.choice(valid, B)
            et = rng.standard_normal((B, Da)).astype(np.float32); ek = rng.standard_normal((K, B, Da)).astype(np.float32)
            want, _, _ = o.update(store, index, idx, et, ek)
            st = eng.cvpo_update(B, indices=idx, eps_target=et, eps_particles=ek)
            for j, kname in enumerate(keys):
                w = float(want[kname])
                assert abs(st[j] - w) <= 2e-4 * abs(w) + 2e-5, (cyc, u, kname, st[j], w)
            d = eng.cvpo_duals()
            np.testing.assert_allclose(d, [o.estep_dual[0].item(), o.estep_dual[1].item(), o.mstep_dual_mu.item(),
                                           o.mstep_dual_std.item()], rtol=2e-4, atol=2e-5)
        o.post_update(); eng.cvpo_post_update()
    for got, ref in ((eng.sac_get_params(0)[0], o.actor_flat()), (eng.sac_get_params(3)[0], o.actor_flat(old=True)),
                     (eng.sac_get_params(1)[0], o.critics_flat()), (eng.sac_get_params(2)[0], o.critics_flat(old=True))):
        d = np.abs(got - ref)      # Adam: an entry whose gradient is rounding noise moves by +-lr per step either way
        assert np.quantile(d, 0.99) <= 1e-5 and d.max() <= 5e-3, (np.quantile(d, 0.99), d.max())
    eng.close()


def test_cvpo_on_a_wrapped_store_device_and_host_chains_agree():
    """Sub-buffers overwritten 2.5 times: the device sampler's n-step chains (sac_sample_kernel) and the caller-RNG
    mode's (host) give bit-identical CVPO updates from the same indices and noise."""
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    rng = np.random.default_rng(4)
    Do, Da, E, sub, B = 5, 3, 3, 40, 48

    def make():
        eng = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=Do, act_dim=Da, hidden=64, n_critics=2, env_num=E,
                                  buffer_size=E * sub, gamma=0.97, target_kl=None))
        eng.cvpo_init(0.2, n_step=3, sample_act_num=8)
        r = np.random.default_rng(0)
        eng.sac_set_params((0.2 * r.standard_normal(eng.n_sac_actor)).astype(np.float32),
                           (0.2 * r.standard_normal(eng.n_sac_critics)).astype(np.float32), 0.0)
        eng.cvpo_pre_update()
        return eng
    dev, twin = make(), make()
    T = 100
    obs = rng.standard_normal((T + 1, E, Do)).astype(np.float32)
    act = np.clip(rng.standard_normal((T, E, Da)), -1, 1).astype(np.float32)
    rew = rng.normal(0, 1, (T, E)); cost = (rng.random((T, E)) < 0.3).astype(np.float64)
    term = rng.random((T, E)) < 0.05; trunc = np.zeros((T, E), bool); trunc[12::13] = True
    for t in range(T):
        ids = [0, 1, 2] if t % 7 else [0, 2]
        for e_ in (dev, twin):
            e_.push(ids, obs[t, ids], act[t, ids], rew[t, ids], cost[t, ids], term[t, ids], trunc[t, ids], obs[t + 1, ids])
    assert len(dev) == E * sub
    for u in range(5):
        dev.cvpo_update(B, seed=9 if u == 0 else 0, sync=False)
        idx, et, _ = dev.sac_last_sample(B)
        ek = dev.cvpo_last_particles(B)
        st = twin.cvpo_update(B, indices=idx, eps_target=et, eps_particles=ek)
        if u == 2:
            for e_ in (dev, twin):
                e_.cvpo_post_update(); e_.cvpo_pre_update()
    rows = dev.sac_drain()
    assert np.isfinite(rows).all() and np.array_equal(rows[-1], st)
    for which in (0, 1, 2, 3):
        assert np.array_equal(dev.sac_get_params(which)[0], twin.sac_get_params(which)[0])
    dev.close(); twin.close()


def test_cvpo_cost_limit_update_and_call_order_errors():
    """update_cost_limit (cvpo.py:165-176) reaches the device threshold; CVPO entry points on a context that was not
    initialised for CVPO fail with the state error instead of running something else."""
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    g, cfg, ocfg, store, index = cvpo_setup("small")
    eng = _engine(cfg, g, ocfg)
    eng.cvpo_pre_update()
    B = cfg["batch_size"]
    st0 = eng.cvpo_update(B, indices=g["indices"][0], eps_target=g["eps_target"][0], eps_particles=g["eps_particles"][0])
    assert abs(st0[15] - ocfg.qc_thres) < 1e-6
    eng.cvpo_set_thres(2.5)
    st1 = eng.cvpo_update(B, indices=g["indices"][1], eps_target=g["eps_target"][1], eps_particles=g["eps_particles"][1])
    assert abs(st1[15] - 2.5) < 1e-6
    with pytest.raises(AssertionError):          # caller-RNG arguments come together
        eng.lib.fsrl_cvpo_update  # noqa: B018  (symbol exists)
        _lib.check(eng.lib.fsrl_cvpo_update(eng._ctx, B, None, None, None, 0, None) or
                   eng.lib.fsrl_cvpo_update(eng._ctx, 0, None, None, None, 0, None))
    eng.close()
    sac = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=4, act_dim=2, hidden=64, n_critics=2, env_num=1, buffer_size=64,
                              target_kl=None))
    sac.sac_init()
    with pytest.raises(_lib.FsrlHipError):
        sac.cvpo_pre_update()
    with pytest.raises(_lib.FsrlHipError):
        sac.cvpo_update(8)
    sizes = sac.store_sizes()
    assert sizes.tolist() == [0]
    sac.close()


@pytest.mark.parametrize("name", ["default", "double"])
def test_cvpo_launch_plans_are_bit_identical(name):
    """r6: sample + gather + the K particles' noise in one launch and the float64 n-step targets inside the critics' tile launch (13
    launches per update) against the stand-alone launches (fsrl_sac_set_plan bits 1 and 2: sampler, gather and n-step kernel on their
    own, 15 launches): same Philox counters, same float64 operations -- statistics, parameters, duals and the drawn sample must agree
    bit for bit over a run of library-RNG updates."""
    g, cfg, ocfg, store, index = cvpo_setup(name)
    B = cfg["batch_size"]
    outs = []
    for plan in (0, 2, 4, 6):
        eng = _engine(cfg, g, ocfg)
        eng.sac_set_plan(plan)
        eng.cvpo_pre_update()
        rows = [eng.cvpo_update(B, seed=9 if u == 0 else 0).copy() for u in range(5)]
        idx, et, _ = eng.sac_last_sample(B)
        outs.append((np.stack(rows), idx.copy(), et.copy(), eng.cvpo_last_particles(B).copy(), np.asarray(eng.cvpo_duals()),
                     eng.sac_get_params(0)[0], eng.sac_get_params(1)[0], eng.sac_get_params(2)[0]))
        eng.close()
    assert np.isfinite(outs[0][0]).all()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)
