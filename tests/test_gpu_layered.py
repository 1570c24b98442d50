"""Layered PPO-Lagrangian contexts (include/fsrl_hip.h fsrl_config.n_hidden; fsrl_amd/csrc/host_layered.inc): `hidden_sizes` that
are not two layers of at most 256 units (fsrl/agent/ppo_lag_agent.py:91,136-145 takes any tuple).  The reference-generated
fixtures ppo_{deep3,wide,one_layer,deep4_options} run through the ordinary tests of test_gpu_ppo.py (process_fn, minibatch
gradient, full update); here: the layered kernels against the fused ones on the SAME two-layer network, a BASELINE-sized batch
against the oracle, the collector-side actor, determinism, and what such a context refuses."""
import numpy as np
import pytest
import torch

from helpers import oracle_cfg_and_data, ppo_case
from test_gpu_ppo import _engine, _push_golden, _rescale, _start

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["c1", "widths", "rewnorm_recompute", "dualclip"])
def test_two_layer_network_through_the_layered_kernels(name):
    """force_layered: the same network, fixture and update through one GEMM launch per Linear instead of the fused kernels --
    both sit within the fixture tolerances of the reference, and within twice that of each other."""
    cfg, g = ppo_case(name)
    res = []
    for force in (False, True):
        eng = _engine(cfg, force_layered=force)
        _start(eng, g)
        _push_golden(eng, g)
        lag = g["lagrangian"]
        stats, stopped = eng.ppo_update(lag, _rescale(lag), cfg["batch_size"], cfg["repeat"], perms=g["perms"])
        np.testing.assert_allclose(stats, g["stats"], rtol=2e-5, atol=2e-5)
        tol = 2e-6 * max(1.0, cfg["lr"] / 5e-4)
        err = np.abs(eng.get_params() - g["theta_final"])
        assert err.max() <= 2 * tol and (err > tol).mean() <= 1e-4, (force, err.max())
        res.append((stats, eng.get_params(), stopped))
        eng.close()
    assert res[0][2] == res[1][2]
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=4e-5, atol=4e-5)
    assert np.abs(res[0][1] - res[1][1]).max() <= 4 * 2e-6 * max(1.0, cfg["lr"] / 5e-4)


@pytest.mark.parametrize("hidden", [(256, 256, 256), (512, 512), (33, 1000, 7)])
def test_batch_of_baseline_size_vs_oracle(hidden):
    """obs 8, act 2, 20 envs x 200 rows (4 000 rows), batch 256 with the merged last minibatch, clip 0.5, two passes, on networks
    the fused kernels cannot hold: logged rows and parameters against the oracle (pinned to the reference by the depth
    fixtures).  Tolerances as in test_full_size_update_vs_oracle's first pass: rows 5e-4, parameters 2e-5."""
    from oracle.ppo_lag import OnPolicyData, PPOLagConfig, PPOLagOracle
    from fsrl_amd.engine import Engine, EngineConfig
    rng = np.random.default_rng(21)
    env_num, rows = 20, 200
    eng = Engine(EngineConfig(obs_dim=8, act_dim=2, hidden_sizes=hidden, env_num=env_num, max_grad_norm=0.5, target_kl=None))
    o = PPOLagOracle(PPOLagConfig(obs_dim=8, act_dim=2, hidden=hidden, max_grad_norm=0.5, target_kl=1e9))
    assert eng.n_params == o.n_params
    torch.manual_seed(6)
    theta = (0.08 * torch.randn(o.n_params)).numpy()
    o.set_params(theta); eng.set_params(theta)
    n = env_num * rows
    obs = rng.standard_normal((n, 8)).astype(np.float32); nxt = rng.standard_normal((n, 8)).astype(np.float32)
    act = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    rew = rng.normal(0.5, 0.5, n); cost = (rng.random(n) < 0.2).astype(np.float64)
    term = np.zeros(n, bool); trunc = np.zeros(n, bool)
    for e in range(env_num):                   # env-major rows; every env: episodes of 80, 80 and an unfinished tail of 40
        trunc[e * rows + 79] = True
        term[e * rows + 159] = True
    for t in range(rows):
        sel = np.arange(env_num) * rows + t
        eng.push(list(range(env_num)), obs[sel], act[sel], rew[sel], cost[sel], term[sel], trunc[sel], nxt[sel])
    end = term | trunc
    end[rows - 1::rows] = True                 # unfinished tails end at the last row of each sub-buffer
    data = OnPolicyData(obs=obs, act=act, rew=rew, cost=cost, terminated=term, truncated=trunc, obs_next=nxt, end_flag=end)
    lag = np.array([0.7])
    perms = np.stack([rng.permutation(n) for _ in range(2)])
    torch.set_num_threads(8)
    _, ostats, _ = o.update(data, lag, _rescale(lag), 256, 2, perms=perms)
    stats, stopped = eng.ppo_update(lag, _rescale(lag), 256, 2, perms=perms)
    assert stopped < 0 and stats.shape == ostats.shape == (2 * (n // 256), 11)
    np.testing.assert_allclose(stats[:10], ostats[:10], rtol=5e-5, atol=5e-5)
    np.testing.assert_allclose(stats, ostats, rtol=5e-4, atol=5e-4)
    err = np.abs(eng.get_params() - o.get_params())
    assert err.max() <= 2e-5 and (err > 4e-6).mean() <= 1e-3, (err.max(), float((err > 4e-6).mean()))
    eng.close()


def test_layered_update_is_deterministic():
    cfg, g = ppo_case("deep3")
    outs = []
    for _ in range(2):
        eng = _engine(cfg)
        _start(eng, g); _push_golden(eng, g)
        lag = g["lagrangian"]
        stats, _ = eng.ppo_update(lag, _rescale(lag), cfg["batch_size"], cfg["repeat"], perms=g["perms"])
        outs.append((stats.copy(), eng.get_params()))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("hidden,k", [((64, 48, 32), 20), ((300, ), 1), ((40, 72, 72, 24), 33)])
def test_collector_actor_of_a_layered_context(hidden, k):
    """fsrl_actor_forward / fsrl_actor_sample on a layered context: mean and std of the policy's Gaussian against the torch
    mirror of the same parameters (tianshou ActorProb: max_action * tanh(mu head), exp(sigma_param))."""
    from fsrl_amd.engine import Engine, EngineConfig
    from fsrl_amd.utils.net import ActorProb, Net
    torch.manual_seed(3)
    Do, Da = 11, 3
    actor = ActorProb(Net((Do, ), hidden_sizes=hidden), (Da, ), max_action=1.5)
    torch.nn.init.constant_(actor.sigma_param, -0.3)
    eng = Engine(EngineConfig(obs_dim=Do, act_dim=Da, hidden_sizes=hidden, env_num=4, max_action=1.5))
    flat = eng.get_params()
    na = sum(p.numel() for p in actor.parameters())
    flat[:na] = torch.cat([p.detach().reshape(-1) for p in actor.parameters()]).numpy()
    eng.set_params(flat)
    obs = np.random.default_rng(1).standard_normal((k, Do)).astype(np.float32)
    mu, sigma = eng.actor_forward(obs)
    with torch.no_grad():
        (tmu, tsig), _ = actor(torch.from_numpy(obs))
    np.testing.assert_allclose(mu, tmu.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(sigma, tsig.numpy(), rtol=1e-6, atol=1e-6)
    a = eng.actor_sample(obs, deterministic=True)
    np.testing.assert_allclose(a, mu, rtol=0, atol=0)
    eng.close()


def test_what_a_layered_context_refuses():
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    eng = Engine(EngineConfig(obs_dim=6, act_dim=2, hidden_sizes=(64, 64, 64), env_num=2))
    with pytest.raises(AssertionError, match="fused"):
        eng.launch_floors(256, 10)
    eng.close()
    for bad in ((), (1, ) * 9):
        with pytest.raises(ValueError, match="1 to 8 hidden layers"):
            Engine(EngineConfig(obs_dim=6, act_dim=2, hidden_sizes=bad, env_num=2))
    with pytest.raises(AssertionError, match="hidden_sizes"):
        Engine(EngineConfig(obs_dim=6, act_dim=2, hidden_sizes=(64, 5000), env_num=2))


@pytest.mark.parametrize("kind", ["ppol", "focops", "cpo", "trpol", "cpo_minibatch", "sacl", "ddpgl", "cvpo"])
def test_agent_with_three_hidden_layers_trains_and_round_trips(kind, tmp_path):
    """The on-policy agents with hidden_sizes=(64, 64, 32): collect with the device actor, update (CPO also with Batch.split
    minibatches inside learn), checkpoint shapes, evaluate."""
    import copy
    from fsrl_amd import agent as A
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    from fsrl_amd.utils import BaseLogger
    env = SyntheticSafetyVectorEnv(env_num=4, episode_len=40, seed=1)
    cls = {"ppol": A.PPOLagAgent, "focops": A.FOCOPSAgent, "cpo": A.CPOAgent, "trpol": A.TRPOLagAgent, "cpo_minibatch": A.CPOAgent,
           "sacl": A.SACLagAgent, "ddpgl": A.DDPGLagAgent, "cvpo": A.CVPOAgent}[kind]
    replay = kind in ("sacl", "ddpgl", "cvpo")
    kw = dict(max_grad_norm=0.5) if kind == "ppol" else dict(buffer_size=4000) if replay else {}
    agent = cls(env, BaseLogger(str(tmp_path), name="t"), cost_limit=10, device="cuda:0", seed=3,
                hidden_sizes=(64, 64, 32), training_num=4, **kw)
    pol = agent.policy
    theta0 = (pol.engine.sac_get_params(0)[0] if replay else pol.engine.get_params()).copy()
    bs = {"ppol": 64, "focops": 64, "cpo": 99999, "trpol": 99999, "cpo_minibatch": 100, "sacl": 64, "ddpgl": 64, "cvpo": 64}[kind]
    if replay:
        agent.learn(env, None, epoch=2, episode_per_collect=4, step_per_epoch=320, update_per_step=0.2, batch_size=bs, verbose=False,
                    save_ckpt=False)
    else:
        agent.learn(env, None, epoch=2, episode_per_collect=4, step_per_epoch=320, repeat_per_collect=2, batch_size=bs, verbose=False,
                    save_ckpt=False)
    sd = copy.deepcopy(pol.state_dict())
    assert tuple(sd["actor.preprocess.model.model.4.weight"].shape) == (32, 64)
    assert all(torch.isfinite(v).all() for v in sd.values() if torch.is_tensor(v) and v.is_floating_point())
    theta1 = pol.engine.sac_get_params(0)[0] if replay else pol.engine.get_params()
    assert np.abs(theta1 - theta0).max() > 1e-4          # it did learn something
    pol.load_state_dict(sd)
    if replay:
        pol._dirty = pol._rest_dirty = True              # the next state_dict() reads the device back
    else:
        pol._mark_stale()
    for kk, v in pol.state_dict().items():
        if torch.is_tensor(v):
            assert torch.equal(v, sd[kk]), kk
    rew, length, cost = agent.evaluate(env, eval_episodes=2)
    assert length == 40.0 and np.isfinite(rew)
