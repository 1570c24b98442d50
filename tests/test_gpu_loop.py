"""Closed-loop parity: collect with the policy -> update on the MI355X -> collect with the updated policy ..., against
the learning curve the UNMODIFIED reference PPOLagrangian produced on the same synthetic env with the same random
streams (tests/golden/gen_golden_loop.py).  This is the results-parity target of BASELINE.md (returns, costs) in
miniature: acting, storing, process_fn + learn and the PID multiplier across 12 collect/update cycles.

Random streams: numpy's for the minibatch permutations, torch's for the sampled actions -- including the draws the
reference's forward() wastes inside update() (PPOLagrangian(reference_rng=True) burns the same amount).
Tolerances (observed: rewards within 1.5e-4 of values ~50, episode costs identical in every cycle, theta mean
diff 1.4e-7): rewards 5e-3 abs, costs equal, PID multiplier 1e-5, parameters mean 2e-6 / max 2e-3."""
import json
import random

import numpy as np
import pytest
import torch
from torch.distributions import Independent, Normal

from helpers import load_npz

pytestmark = pytest.mark.gpu


def _rollout(policy, env, buf, noise=False):
    from fsrl_amd.data import Batch
    obs, _ = env.reset()
    E = len(env)
    ids = np.arange(E)
    ep_rew, ep_cost, steps = np.zeros(E), np.zeros(E), 0
    while True:
        with torch.no_grad():
            res = policy(Batch(obs=obs, info={}), None)
        act = res.act.numpy()
        if noise:
            act = policy.exploration_noise(act, None)
        obs_next, rew, term, trunc, info = env.step(policy.map_action(act), ids)
        buf.add(Batch(obs=obs, act=act, rew=rew, info={"cost": info["cost"]}, terminated=term, truncated=trunc,
                      obs_next=obs_next), buffer_ids=ids)
        ep_rew += rew; ep_cost += info["cost"]; steps += E
        obs = obs_next
        if (term | trunc).all():
            break
    return dict(reward=float(ep_rew.mean()), cost=float(ep_cost.mean()), steps=steps)


def test_closed_training_loop_tracks_the_reference_learning_curve():
    from fsrl_amd.data import HipVectorReplayBuffer
    from fsrl_amd.env import Box, SyntheticSafetyVectorEnv
    from fsrl_amd.policy import PPOLagrangian
    from fsrl_amd.utils.net import ActorProb, Critic, Net
    g = load_npz("loop_ppo.npz"); cfg = json.loads(str(g["cfg_json"]))
    Do, Da, h, E = cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"]), cfg["env_num"]
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), max_action=1.0)
    critics = [Critic(Net((Do, ), hidden_sizes=h)) for _ in range(2)]

    class Cap:
        def __init__(self): self.rows = []
        def store(self, tab=None, **kw): self.rows.append({(tab + "/" + k if tab else k): float(v) for k, v in kw.items()})
        def print(self, *a, **k): pass
    log = Cap()
    from fsrl_amd.utils.net import ActorCritic
    optim = torch.optim.Adam(ActorCritic(actor, critics).parameters(), lr=cfg["lr"])
    pol = PPOLagrangian(actor, critics, optim, lambda *l: Independent(Normal(*l), 1), logger=log, cost_limit=cfg["cost_limit"],
                        target_kl=cfg["target_kl"], max_grad_norm=cfg["max_grad_norm"],
                        observation_space=Box(-np.inf, np.inf, (Do, )), action_space=Box(-1, 1, (Da, )), device=0, env_num=E,
                        buffer_size=E * cfg["ep_len"] * 2, reference_rng=True)
    pol.engine.set_params(g["theta0"]); pol._pull_params()
    pol.train()
    env = SyntheticSafetyVectorEnv(env_num=E, obs_dim=Do, act_dim=Da, episode_len=cfg["ep_len"], seed=cfg["seed"] + 11)
    buf = HipVectorReplayBuffer(pol.engine, E * cfg["ep_len"] * 2, E)
    seed = cfg["seed"] + 7
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    keys = [str(k) for k in g["stat_keys"]]
    curve, lags = [], []
    for c in range(cfg["cycles"]):
        buf.reset()
        st = _rollout(pol, env, buf)
        pol.pre_update_fn(stats_train={"cost": st["cost"]})
        lags.append(pol.lag_optims[0].get_lag())
        n0 = len(log.rows)
        pol.update(0, buf, batch_size=cfg["batch_size"], repeat=cfg["repeat"])
        rows = [r for r in log.rows[n0:] if "update/gradient_steps" not in r]
        assert len(rows) // 2 == int(g["steps_per_update"][c])
        first = {**rows[0], **rows[1]}
        last = {**rows[-2], **rows[-1]}
        curve.append([st["reward"], st["cost"], st["steps"]])
        assert abs(st["reward"] - g["curve"][c][0]) <= 5e-3 and st["cost"] == g["curve"][c][1], (c, st, g["curve"][c])
        np.testing.assert_allclose([first[k] for k in keys], g["first_rows"][c], rtol=2e-4, atol=1e-4)
        np.testing.assert_allclose([last[k] for k in keys], g["last_rows"][c], rtol=2e-4, atol=1e-4)
    curve, want = np.array(curve), g["curve"]
    assert np.array_equal(curve[:, 2], want[:, 2])
    np.testing.assert_allclose(lags, g["lagrangians"][:, 0], rtol=1e-5, atol=1e-7)
    d = np.abs(pol.engine.get_params() - g["theta_final"])
    assert d.mean() <= 2e-6 and d.max() <= 2e-3, (d.mean(), d.max())
    print("closed loop: max |reward diff|", np.abs(curve[:, 0] - want[:, 0]).max(), "max |cost diff|",
          np.abs(curve[:, 1] - want[:, 1]).max(), "theta mean/max diff", d.mean(), d.max())
    pol.engine.close()


def test_closed_offpolicy_loop_tracks_the_reference_learning_curve():
    """SAC-Lagrangian: the store accumulates over 10 collects, 30 updates after each (300 updates in total), same
    numpy (buffer.sample) and torch (rsample in acting and in both update forwards) streams as the reference."""
    from fsrl_amd.data import HipVectorReplayBuffer
    from fsrl_amd.env import Box, SyntheticSafetyVectorEnv
    from fsrl_amd.policy import SACLagrangian
    from fsrl_amd.utils.net import ActorProb, DoubleCritic, Net
    g = load_npz("loop_sac.npz"); cfg = json.loads(str(g["cfg_json"]))
    Do, Da, h, E = cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"]), cfg["env_num"]
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), conditioned_sigma=True, unbounded=True)
    critics = [DoubleCritic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True), Net((Do, ), (Da, ), hidden_sizes=h, concat=True))
               for _ in range(2)]
    SACLagrangian._unflat([actor], g["theta_actor0"]); SACLagrangian._unflat(critics, g["theta_critics0"])

    class Cap:
        def __init__(self): self.rows = []
        def store(self, tab=None, **kw): self.rows.append({(tab + "/" + k if tab else k): float(v) for k, v in kw.items()})
        def print(self, *a, **k): pass
    log = Cap()
    la = torch.zeros(1, requires_grad=True)
    pol = SACLagrangian(actor, critics, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                        torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=cfg["critic_lr"]), logger=log,
                        alpha=(-float(Da), la, torch.optim.Adam([la], lr=cfg["alpha_lr"])), tau=cfg["tau"], n_step=cfg["n_step"],
                        cost_limit=cfg["cost_limit"], gamma=cfg["gamma"], observation_space=Box(-np.inf, np.inf, (Do, )),
                        action_space=Box(-1, 1, (Da, )), device=0, env_num=E, buffer_size=E * cfg["ep_len"] * cfg["cycles"],
                        reference_rng=True)
    pol.train()
    env = SyntheticSafetyVectorEnv(env_num=E, obs_dim=Do, act_dim=Da, episode_len=cfg["ep_len"], seed=cfg["seed"] + 11)
    buf = HipVectorReplayBuffer(pol.engine, E * cfg["ep_len"] * cfg["cycles"], E)
    seed = cfg["seed"] + 7
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    keys = [str(k) for k in g["stat_keys"]]
    worst_r, worst_c = 0.0, 0.0
    for c in range(cfg["cycles"]):
        st = _rollout(pol, env, buf)
        pol.pre_update_fn(stats_train={"cost": st["cost"]})
        assert abs(pol.lag_optims[0].get_lag() - g["lagrangians"][c][0]) <= 1e-4 * max(1.0, g["lagrangians"][c][0])
        n0 = len(log.rows)
        for _ in range(cfg["updates_per_cycle"]):
            pol.update(cfg["batch_size"], buf)
        last = {**log.rows[-2], **log.rows[-1]}
        worst_r = max(worst_r, abs(st["reward"] - g["curve"][c][0])); worst_c = max(worst_c, abs(st["cost"] - g["curve"][c][1]))
        assert abs(st["reward"] - g["curve"][c][0]) <= 0.05 and abs(st["cost"] - g["curve"][c][1]) <= 0.5, (c, st, g["curve"][c])
        # 300 dependent updates: fp32 rounding differences grow (Q-learning feeds its own targets): tight early, 2 % late
        tol = 2e-3 if c < 5 else 2e-2
        np.testing.assert_allclose([last[k] for k in keys], g["last_rows"][c], rtol=tol, atol=tol)
        assert abs(float(pol.engine.sac_get_params(0)[1]) - g["alphas"][c]) <= 1e-5
    d = np.abs(pol.engine.sac_get_params(0)[0] - g["theta_actor_final"])
    print("closed SAC loop: max |reward diff|", worst_r, "max |cost diff|", worst_c, "actor theta mean/max diff", d.mean(), d.max())
    assert d.mean() <= 1e-4, (d.mean(), d.max())
    pol.engine.close()


class _Cap:
    def __init__(self): self.rows = []
    def store(self, tab=None, **kw): self.rows.append({(tab + "/" + k if tab else k): float(v) for k, v in kw.items()})
    def print(self, *a, **k): pass


def test_closed_cvpo_loop_tracks_the_reference_learning_curve():
    """CVPO: 10 collects x 30 updates with actor_old refreshed after every cycle; same numpy (buffer.sample) and torch
    (Normal.sample in acting, target action, the unused forward draws and the K particles) streams as the reference."""
    from torch.distributions import Independent, Normal
    from fsrl_amd.data import HipVectorReplayBuffer
    from fsrl_amd.env import Box, SyntheticSafetyVectorEnv
    from fsrl_amd.policy import CVPO, SACLagrangian
    from fsrl_amd.utils.net import ActorProb, Net, SingleCritic
    g = load_npz("loop_cvpo.npz"); cfg = json.loads(str(g["cfg_json"]))
    Do, Da, h, E = cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"]), cfg["env_num"]
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), conditioned_sigma=True, unbounded=False)
    critics = [SingleCritic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True)) for _ in range(2)]
    SACLagrangian._unflat([actor], g["theta_actor0"]); SACLagrangian._unflat(critics, g["theta_critics0"])
    log = _Cap()
    pol = CVPO(actor, critics, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
               torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=cfg["critic_lr"]), action_space=Box(-1, 1, (Da, )),
               dist_fn=lambda *l: Independent(Normal(*l), 1), max_episode_steps=cfg["ep_len"], logger=log,
               cost_limit=cfg["cost_limit"], tau=cfg["tau"], gamma=cfg["gamma"], n_step=cfg["n_step"],
               observation_space=Box(-np.inf, np.inf, (Do, )), device=0, env_num=E,
               buffer_size=E * cfg["ep_len"] * cfg["cycles"], reference_rng=True)
    pol.train()
    env = SyntheticSafetyVectorEnv(env_num=E, obs_dim=Do, act_dim=Da, episode_len=cfg["ep_len"], seed=cfg["seed"] + 11)
    buf = HipVectorReplayBuffer(pol.engine, E * cfg["ep_len"] * cfg["cycles"], E)
    seed = cfg["seed"] + 7
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    keys = [str(k) for k in g["stat_keys"]]
    worst_r, worst_c = 0.0, 0.0
    for c in range(cfg["cycles"]):
        st = _rollout(pol, env, buf)
        pol.pre_update_fn(stats_train={"cost": st["cost"]})
        n0 = len(log.rows)
        for _ in range(cfg["updates_per_cycle"]):
            pol.update(cfg["batch_size"], buf)
        duals = pol.engine.cvpo_duals()
        pol.post_update_fn(stats_train={"cost": st["cost"]})
        last = {}
        for r in log.rows[-5:]:                       # one update = five store() calls in the façade
            last.update(r)
        worst_r = max(worst_r, abs(st["reward"] - g["curve"][c][0])); worst_c = max(worst_c, abs(st["cost"] - g["curve"][c][1]))
        assert abs(st["reward"] - g["curve"][c][0]) <= 0.05 and abs(st["cost"] - g["curve"][c][1]) <= 0.5, (c, st, g["curve"][c])
        tol = 2e-3 if c < 5 else 2e-2
        np.testing.assert_allclose([last[k] for k in keys], g["last_rows"][c], rtol=tol, atol=tol, err_msg=f"cycle {c}")
        np.testing.assert_allclose(duals, g["duals"][c], rtol=tol, atol=tol, err_msg=f"cycle {c}")
    d = np.abs(pol.engine.sac_get_params(0)[0] - g["theta_actor_final"])
    print("closed CVPO loop: max |reward diff|", worst_r, "max |cost diff|", worst_c, "actor theta mean/max diff", d.mean(), d.max())
    assert d.mean() <= 1e-4, (d.mean(), d.max())
    pol.engine.close()


def test_closed_focops_loop_tracks_the_reference_learning_curve():
    """FOCOPS incl. the cycles in which the reference stops a pass early on the KL threshold."""
    from fsrl_amd.data import HipVectorReplayBuffer
    from fsrl_amd.env import Box, SyntheticSafetyVectorEnv
    from fsrl_amd.policy import FOCOPS
    from fsrl_amd.utils.net import ActorProb, Critic, Net
    g = load_npz("loop_focops.npz"); cfg = json.loads(str(g["cfg_json"]))
    Do, Da, h, E = cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"]), cfg["env_num"]
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), max_action=1.0)
    critics = [Critic(Net((Do, ), hidden_sizes=h)) for _ in range(2)]
    log = _Cap()
    pol = FOCOPS(actor, critics, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                 torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=cfg["critic_lr"]),
                 lambda *l: Independent(Normal(*l), 1), logger=log, cost_limit=cfg["cost_limit"],
                 nu=(cfg["nu_max"], cfg["nu_lr"], torch.zeros(1)), observation_space=Box(-np.inf, np.inf, (Do, )),
                 action_space=Box(-1, 1, (Da, )), device=0, env_num=E, buffer_size=E * cfg["ep_len"] * 2, reference_rng=True)
    pol.engine.set_params(g["theta0"]); pol._pull_params()
    pol.train()
    env = SyntheticSafetyVectorEnv(env_num=E, obs_dim=Do, act_dim=Da, episode_len=cfg["ep_len"], seed=cfg["seed"] + 11)
    buf = HipVectorReplayBuffer(pol.engine, E * cfg["ep_len"] * 2, E)
    seed = cfg["seed"] + 7
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    keys = [str(k) for k in g["stat_keys"]]
    worst = 0.0
    for c in range(cfg["cycles"]):
        buf.reset()
        st = _rollout(pol, env, buf)
        pol.pre_update_fn(stats_train={"cost": st["cost"]})
        n0 = len(log.rows)
        pol.update(0, buf, batch_size=cfg["batch_size"], repeat=cfg["repeat"])
        rows = [r for r in log.rows[n0:] if "update/gradient_steps" not in r]
        assert len(rows) // 3 == int(g["steps_per_update"][c]), (c, len(rows) // 3, g["steps_per_update"][c])   # same early stops
        last = {**rows[-3], **rows[-2], **rows[-1]}
        worst = max(worst, abs(st["reward"] - g["curve"][c][0]))
        assert abs(st["reward"] - g["curve"][c][0]) <= 5e-3 and st["cost"] == g["curve"][c][1], (c, st, g["curve"][c])
        np.testing.assert_allclose([last[k] for k in keys], g["last_rows"][c], rtol=3e-4, atol=1e-4)
        assert abs(float(pol._nu) - g["nus"][c]) <= 1e-6
    d = np.abs(pol.engine.get_params() - g["theta_final"])
    print("closed FOCOPS loop: max |reward diff|", worst, "theta mean/max diff", d.mean(), d.max())
    assert d.mean() <= 2e-6 and d.max() <= 2e-3, (d.mean(), d.max())
    pol.engine.close()


def test_closed_ddpg_loop_tracks_the_reference_learning_curve():
    """DDPG-Lagrangian: deterministic actor + Gaussian exploration noise from numpy's stream, 300 updates."""
    from fsrl_amd.data import HipVectorReplayBuffer
    from fsrl_amd.env import Box, SyntheticSafetyVectorEnv
    from fsrl_amd.policy import DDPGLagrangian, SACLagrangian
    from fsrl_amd.utils.net import Actor, Critic, GaussianNoise, Net
    g = load_npz("loop_ddpg.npz"); cfg = json.loads(str(g["cfg_json"]))
    Do, Da, h, E = cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"]), cfg["env_num"]
    actor = Actor(Net((Do, ), hidden_sizes=h), (Da, ), max_action=1.0)
    critics = [Critic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True)) for _ in range(2)]
    SACLagrangian._unflat([actor], g["theta_actor0"]); SACLagrangian._unflat(critics, g["theta_critics0"])
    log = _Cap()
    pol = DDPGLagrangian(actor, critics, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                         torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=cfg["critic_lr"]), logger=log,
                         tau=cfg["tau"], exploration_noise=GaussianNoise(sigma=cfg["exploration_sigma"]), n_step=cfg["n_step"],
                         cost_limit=cfg["cost_limit"], gamma=cfg["gamma"], observation_space=Box(-np.inf, np.inf, (Do, )),
                         action_space=Box(-1, 1, (Da, )), device=0, env_num=E, buffer_size=E * cfg["ep_len"] * cfg["cycles"],
                         reference_rng=True)
    pol.train()
    env = SyntheticSafetyVectorEnv(env_num=E, obs_dim=Do, act_dim=Da, episode_len=cfg["ep_len"], seed=cfg["seed"] + 11)
    buf = HipVectorReplayBuffer(pol.engine, E * cfg["ep_len"] * cfg["cycles"], E)
    seed = cfg["seed"] + 7
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    keys = [str(k) for k in g["stat_keys"]]
    worst_r = worst_c = 0.0
    for c in range(cfg["cycles"]):
        st = _rollout(pol, env, buf, noise=True)
        pol.pre_update_fn(stats_train={"cost": st["cost"]})
        assert abs(pol.lag_optims[0].get_lag() - g["lagrangians"][c][0]) <= 1e-4 * max(1.0, g["lagrangians"][c][0])
        for _ in range(cfg["updates_per_cycle"]):
            pol.update(cfg["batch_size"], buf)
        last = {**log.rows[-2], **log.rows[-1]}
        worst_r = max(worst_r, abs(st["reward"] - g["curve"][c][0])); worst_c = max(worst_c, abs(st["cost"] - g["curve"][c][1]))
        assert abs(st["reward"] - g["curve"][c][0]) <= 0.05 and abs(st["cost"] - g["curve"][c][1]) <= 0.5, (c, st, g["curve"][c])
        tol = 2e-3 if c < 5 else 2e-2
        np.testing.assert_allclose([last[k] for k in keys], g["last_rows"][c], rtol=tol, atol=tol)
    d = np.abs(pol.engine.sac_get_params(0)[0] - g["theta_actor_final"])
    print("closed DDPG loop: max |reward diff|", worst_r, "max |cost diff|", worst_c, "actor theta mean/max diff", d.mean(), d.max())
    assert d.mean() <= 1e-4, (d.mean(), d.max())
    pol.engine.close()


@pytest.mark.parametrize("kind", ["cpo", "trpo"])
def test_closed_trust_region_loop_tracks_the_reference_learning_curve(kind):
    """CPO / TRPO-Lagrangian: full-batch updates with conjugate gradients and a line search in every cycle.  fp32 CG on
    the damped Hessian amplifies rounding differences (conditioning note: the reference itself moves by
    ~1e-3 in Q/R/S when its batch is merely re-ordered), so one update already leaves theta ~1e-4..1e-3 apart and a
    stochastic closed loop turns that into percent-level differences of the episode returns within a few cycles.
    What is asserted: the first cycle is exact (same acting, storing, random stream), the second within 5 %, and for
    TRPO-Lag the whole curve within 10 % of its range (observed 0.2 .. 4.1 on returns of ~50).  CPO's line search
    exhausts its backtracks in most cycles of this fixture (step 0.8^10 in the reference too) and the two stochastic runs
    decorrelate after the third cycle; asserted for CPO: the mean |reward difference| over the curve stays below a quarter
    of the curve's range (observed 0.15), episode costs within 12 of the reference's in every cycle and within 6 on average
    (a cycle's cost is the mean of 8 episodes, standard error ~3.5: two decorrelated runs of the SAME algorithm differ by ~5
    typically; observed max 5.5 .. 8.75, mean ~4 across builds whose kernels differ in the last bit), line-search step sizes
    equal in the first cycle and within four backtracks afterwards, and the cost falls by at least half of the reference's.
    The reference against ITSELF (tests/golden/loop_sensitivity.py: the same generator with 1 torch thread instead of 4, i.e. only
    the GEMM summation order changes): TRPO-Lag rewards drift 0.2 -> 2.8 over the eight cycles (costs up to 1.0), CPO rewards up to
    26 and costs up to 4.6 -- the bands here are the reference's own reproducibility, not slack for the port."""
    from fsrl_amd.data import HipVectorReplayBuffer
    from fsrl_amd.env import Box, SyntheticSafetyVectorEnv
    from fsrl_amd.policy import CPO, TRPOLagrangian
    from fsrl_amd.utils.net import ActorProb, Critic, Net
    g = load_npz(f"loop_{kind}.npz"); cfg = json.loads(str(g["cfg_json"]))
    Do, Da, h, E = cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"]), cfg["env_num"]
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), max_action=1.0)
    critics = [Critic(Net((Do, ), hidden_sizes=h)) for _ in range(2)]
    log = _Cap()
    cls = CPO if kind == "cpo" else TRPOLagrangian
    pol = cls(actor, critics, torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=cfg["lr"]),
              lambda *l: Independent(Normal(*l), 1), logger=log, cost_limit=cfg["cost_limit"],
              optim_critic_iters=cfg["optim_critic_iters"], observation_space=Box(-np.inf, np.inf, (Do, )),
              action_space=Box(-1, 1, (Da, )), device=0, env_num=E, buffer_size=E * cfg["ep_len"] * 2, reference_rng=True)
    pol.engine.set_params(g["theta0"]); pol._pull_params()
    pol.train()
    env = SyntheticSafetyVectorEnv(env_num=E, obs_dim=Do, act_dim=Da, episode_len=cfg["ep_len"], seed=cfg["seed"] + 11)
    buf = HipVectorReplayBuffer(pol.engine, E * cfg["ep_len"] * 2, E)
    seed = cfg["seed"] + 7
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    keys = [str(k) for k in g["stat_keys"]]
    si = keys.index("loss/step_size")
    diffs = []
    for c in range(cfg["cycles"]):
        buf.reset()
        st = _rollout(pol, env, buf)
        pol.pre_update_fn(stats_train={"cost": st["cost"]})
        n0 = len(log.rows)
        pol.update(0, buf, batch_size=99999, repeat=cfg["repeat"])
        rows = [r for r in log.rows[n0:] if "update/gradient_steps" not in r]
        per = len(rows) // cfg["repeat"]
        last = {}
        for r in rows[-per:]:
            last.update(r)
        diffs.append((abs(st["reward"] - g["curve"][c][0]), abs(st["cost"] - g["curve"][c][1]),
                      last["loss/step_size"], g["last_rows"][c][si], st["cost"]))
    print(f"closed {kind} loop (|reward diff|, |cost diff|, step got, step want):")
    for d in diffs:
        print("   ", d)
    th = np.abs(pol.engine.get_params() - g["theta_final"])
    print("    theta mean/max diff", th.mean(), th.max())
    assert diffs[0][0] <= 1e-3 and diffs[0][1] == 0
    span = float(g["curve"][:, 0].max() - g["curve"][:, 0].min())
    assert diffs[1][0] <= 0.05 * max(abs(float(g["curve"][1][0])), span), diffs[1]
    if kind == "trpo":
        assert max(d[0] for d in diffs) <= max(0.1 * span, 6.0), [d[0] for d in diffs]
    else:
        assert np.mean([d[0] for d in diffs]) <= 0.25 * span, ([d[0] for d in diffs], span)
        assert max(d[1] for d in diffs) <= 12.0 and np.mean([d[1] for d in diffs]) <= 6.0, [d[1] for d in diffs]
        # line search: the same outcome while the runs are still correlated (r5: cycles 0-3), and afterwards -- two decorrelated
        # trajectories of a chaotic loop -- never further apart than the range both runs cover: 5 to 10 backtracks in every cycle
        # (r4 build: 6 to 10, bar four; r5's Gauss-Newton first repeat and contraction-free loss head moved the device's
        # trajectory, not its distribution: equal in 4 of 8 cycles)
        backs = [abs(np.log(d[2] / d[3]) / np.log(0.8)) for d in diffs]
        assert backs[0] < 1e-3 and max(backs) <= 5.05, backs
        ref_fall = float(g["curve"][0][1] - g["curve"][-1][1])
        assert diffs[0][4] - diffs[-1][4] >= 0.5 * ref_fall, (diffs[0][4], diffs[-1][4], ref_fall)
    pol.engine.close()
