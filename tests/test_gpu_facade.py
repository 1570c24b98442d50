"""GPU tests of the drop-in surface: fsrl_amd.policy / agent / trainer / data over the engine."""
import numpy as np
import pytest
import torch

from helpers import ppo_case

pytestmark = pytest.mark.gpu


def _policy_from_case(cfg, g, logger=None):
    from torch.distributions import Independent, Normal
    from fsrl_amd.env import Box
    from fsrl_amd.policy import PPOLagrangian
    from fsrl_amd.utils.net import ActorCritic, ActorProb, Critic, Net
    h, Do, Da = tuple(cfg["hidden"]), cfg["obs_dim"], cfg["act_dim"]
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), max_action=cfg["max_action"],
                      unbounded=bool(cfg.get("unbounded", False)))
    critics = [Critic(Net((Do, ), hidden_sizes=h)) for _ in range(2)]
    ac = ActorCritic(actor, critics)
    flat, off = torch.from_numpy(g["theta0"]), 0
    with torch.no_grad():
        for p in ac.parameters():
            p.copy_(flat[off:off + p.numel()].view_as(p)); off += p.numel()
    optim = torch.optim.Adam(ac.parameters(), lr=cfg["lr"])
    return PPOLagrangian(actor, critics, optim, lambda *l: Independent(Normal(*l), 1), logger=logger,
                         target_kl=cfg["target_kl"], vf_coef=cfg["vf_coef"],
                         max_grad_norm=cfg["max_grad_norm"], gae_lambda=cfg["gae_lambda"],
                         eps_clip=cfg["eps_clip"], dual_clip=cfg["dual_clip"],
                         advantage_normalization=cfg["advantage_normalization"],
                         recompute_advantage=bool(cfg.get("recompute_advantage", False)),
                         value_clip=bool(cfg.get("value_clip", False)),
                         reward_normalization=bool(cfg.get("reward_normalization", False)),
                         use_lagrangian=cfg["use_lagrangian"], cost_limit=cfg["cost_limit"],
                         gamma=cfg["gamma"], observation_space=Box(-np.inf, np.inf, (Do, )),
                         action_space=Box(-1, 1, (Da, )), device=0, env_num=cfg["env_num"])


class _Capture:
    def __init__(self):
        self.rows, self.msgs = [], []

    def store(self, tab=None, **kw):
        self.rows.append({(tab + "/" + k if tab else k): v for k, v in kw.items()})

    def print(self, msg, *a):
        self.msgs.append(msg)


@pytest.mark.parametrize("name", ["tiny", "c1", "earlystop", "rewnorm_first", "rewnorm_recompute", "unbounded"])
def test_policy_update_through_facade_matches_reference(name):
    """Same call sequence as OnpolicyTrainer.policy_update_fn, numpy RNG seeded like the golden
    generator: the facade draws the SAME permutations as the reference's Batch.split."""
    import random
    from fsrl_amd.data import Batch, HipVectorReplayBuffer
    cfg, g = ppo_case(name)
    log = _Capture()
    pol = _policy_from_case(cfg, g, log)
    pol.train()
    buf = HipVectorReplayBuffer(pol.engine, 100000, cfg["env_num"])
    rows = g["env_rows"]; off = np.concatenate([[0], np.cumsum(rows)])
    for t in range(rows.max()):
        ids = np.array([e for e in range(len(rows)) if t < rows[e]])
        sel = np.array([off[e] + t for e in ids])
        buf.add(Batch(obs=g["buf_obs"][sel], act=g["buf_act"][sel], rew=g["buf_rew"][sel],
                      info={"cost": g["buf_cost"][sel]}, terminated=g["buf_terminated"][sel],
                      truncated=g["buf_truncated"][sel], obs_next=g["buf_obs_next"][sel]), buffer_ids=ids)
    pol.pre_update_fn(stats_train={"cost": cfg["cost_stat"]})
    assert pol.lag_optims[0].get_lag() == g["lagrangian"][0]
    rms = lambda: np.array([[r.mean, r.var, r.count] for r in pol.ret_rms])      # noqa: E731
    assert np.array_equal(rms(), [[0.0, 1.0, 0.0]] * 2)                # RunningMeanStd(): mean 0, var 1, count 0
    if cfg.get("reward_normalization"):
        pol.engine.ret_rms_set(g["ret_rms0"])
    seed = cfg["seed"] + 7
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    out = pol.update(0, buf, batch_size=cfg["batch_size"], repeat=cfg["repeat"])
    assert out["gradient_steps"] == int(g["gradient_steps"]) == pol.gradient_steps
    assert (out["early_stop_pass"] >= 0) == bool(g["early_stop_msgs"]) == bool(log.msgs)
    keys = [str(k) for k in g["stats_keys"]]
    rows_ = [r for r in log.rows if "update/gradient_steps" not in r]
    got = np.array([[{**rows_[i], **rows_[i + 1]}[k] for k in keys] for i in range(0, len(rows_), 2)])
    np.testing.assert_allclose(got, g["stats"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(pol._flat_params(), g["theta_final"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(rms(), g["ret_rms_final"], rtol=1e-5, atol=1e-7)
    # checkpoint round trip under the reference's key names
    sd = pol.state_dict()
    pol2 = _policy_from_case(cfg, g)
    pol2.load_state_dict(sd)
    assert np.array_equal(pol2.engine.get_params(), pol.engine.get_params())
    assert pol2.lag_optims[0].get_lag() == pol.lag_optims[0].get_lag()


def test_agent_learn_end_to_end_on_synthetic_env(tmp_path):
    from fsrl_amd.agent import PPOLagAgent
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    from fsrl_amd.utils import BaseLogger
    train = SyntheticSafetyVectorEnv(env_num=4, episode_len=50, seed=1)
    test = SyntheticSafetyVectorEnv(env_num=2, episode_len=50, seed=2)
    agent = PPOLagAgent(train, BaseLogger(str(tmp_path), name="t"), cost_limit=10, device="cuda:0", seed=3,
                        hidden_sizes=(64, 64), max_grad_norm=0.5, training_num=4)
    theta0 = agent.policy.engine.get_params()
    ep, stat, info = agent.learn(train, test, epoch=2, episode_per_collect=4, step_per_epoch=400,
                                 repeat_per_collect=2, batch_size=64, testing_num=2, save_interval=1,
                                 verbose=False)
    assert ep == 2 and info["train_speed"] > 0 and info["policy_update_time"] > 0
    assert "loss/kl" in stat and "train/reward" in stat and "test/cost" in stat
    assert np.isfinite(list(stat.values())).all()
    theta1 = agent.policy.engine.get_params()
    assert np.abs(theta1 - theta0).max() > 1e-4                      # it learned something
    assert np.array_equal(agent.policy._flat_params(), theta1)      # host mirror in sync
    rew, length, cost = agent.evaluate(test, eval_episodes=2)
    assert length == 50.0 and np.isfinite(rew) and cost >= 0
    assert (tmp_path / "t" / "checkpoint" / "model.pt").exists()


@pytest.mark.parametrize("kind", ["cpo", "trpo"])
def test_trust_region_agents_learn_end_to_end(kind, tmp_path):
    from fsrl_amd.agent import CPOAgent, TRPOLagAgent
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    from fsrl_amd.utils import BaseLogger
    train = SyntheticSafetyVectorEnv(env_num=4, episode_len=50, seed=1)
    Agent = CPOAgent if kind == "cpo" else TRPOLagAgent
    agent = Agent(train, BaseLogger(str(tmp_path), name="t"), cost_limit=10, device="cuda:0", seed=3,
                  hidden_sizes=(64, 64), optim_critic_iters=3, training_num=4)
    theta0 = agent.policy.engine.get_params()
    ep, stat, info = agent.learn(train, None, epoch=2, episode_per_collect=4, step_per_epoch=400,
                                 repeat_per_collect=2, batch_size=99999, verbose=False)
    assert ep == 2 and np.isfinite(list(stat.values())).all()
    assert "loss/step_size" in stat and "loss/vf_total" in stat and "loss/kl" in stat
    assert ("loss/optim_case" in stat) == (kind == "cpo")
    theta1 = agent.policy.engine.get_params()
    assert np.abs(theta1 - theta0).max() > 1e-5 and np.array_equal(agent.policy._flat_params(), theta1)


def test_sac_policy_update_through_facade_matches_reference():
    """The facade reproduces the reference's random streams: numpy RNG for buffer.sample (tianshou's
    sub-buffer-proportional rule), torch RNG for the two rsample draws."""
    import json, random
    from fsrl_amd.data import Batch, HipVectorReplayBuffer
    from fsrl_amd.env import Box
    from fsrl_amd.policy import SACLagrangian
    from fsrl_amd.utils.net import ActorProb, DoubleCritic, Net
    from helpers import load_npz
    g = load_npz("sac_small.npz"); cfg = json.loads(str(g["cfg_json"]))
    Do, Da, h = cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"])
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), conditioned_sigma=True, unbounded=True)
    critics = [DoubleCritic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True),
                            Net((Do, ), (Da, ), hidden_sizes=h, concat=True)) for _ in range(2)]
    SACLagrangian._unflat([actor], g["theta_actor0"]); SACLagrangian._unflat(critics, g["theta_critics0"])
    log_alpha = torch.zeros(1, requires_grad=True)
    alpha = (-float(Da), log_alpha, torch.optim.Adam([log_alpha], lr=cfg["alpha_lr"]))
    log = _Capture()
    pol = SACLagrangian(actor, critics, torch.optim.Adam(actor.parameters(), lr=cfg["actor_lr"]),
                        torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=cfg["critic_lr"]),
                        logger=log, alpha=alpha, tau=cfg["tau"], n_step=cfg["n_step"], cost_limit=cfg["cost_limit"],
                        gamma=cfg["gamma"], observation_space=Box(-np.inf, np.inf, (Do, )),
                        action_space=Box(-1, 1, (Da, )), device=0, env_num=cfg["env_num"], reference_rng=True)
    pol.train()
    buf = HipVectorReplayBuffer(pol.engine, cfg["buffer_size"], cfg["env_num"])
    rows = g["env_rows"]; off = np.concatenate([[0], np.cumsum(rows)])
    for t in range(rows.max()):
        ids = np.array([e for e in range(len(rows)) if t < rows[e]])
        sel = np.array([off[e] + t for e in ids])
        buf.add(Batch(obs=g["st_obs"][sel], act=g["st_act"][sel], rew=g["st_rew"][sel],
                      info={"cost": g["st_cost"][sel]}, terminated=g["st_terminated"][sel],
                      truncated=g["st_truncated"][sel], obs_next=g["st_obs_next"][sel]), buffer_ids=ids)
    pol.pre_update_fn(stats_train={"cost": cfg["cost_stat"]})
    seed = cfg["seed"] + 7
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    ka = [str(k) for k in g["stats_actor_keys"]]; kc = [str(k) for k in g["stats_critic_keys"]]
    for u in range(cfg["n_updates"]):
        pol.update(cfg["batch_size"], buf)
        ra, rc = log.rows[2 * u], log.rows[2 * u + 1]
        np.testing.assert_allclose([ra[k] for k in ka], g["stats_actor"][u], rtol=5e-5, atol=5e-6)
        np.testing.assert_allclose([rc[k] for k in kc], g["stats_critic"][u], rtol=5e-5, atol=5e-6)
    sd = pol.state_dict()
    assert "critics_old.0.preprocess1.model.model.0.weight" in sd and "actor.sigma.model.0.weight" in sd
    d = np.abs(SACLagrangian._flat([pol.actor]) - g["theta_actor_final"])
    assert np.quantile(d, 0.99) <= 5e-6 and d.max() <= 5e-4


def test_sac_agent_learns_end_to_end(tmp_path):
    from fsrl_amd.agent import SACLagAgent
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    from fsrl_amd.utils import BaseLogger
    train = SyntheticSafetyVectorEnv(env_num=4, episode_len=40, seed=1)
    agent = SACLagAgent(train, BaseLogger(str(tmp_path), name="s"), cost_limit=10, device="cuda:0", seed=3,
                        hidden_sizes=(64, 64), training_num=4, buffer_size=4000)
    a0, _ = agent.policy.engine.sac_get_params(0)
    ep, stat, info = agent.learn(train, None, epoch=2, episode_per_collect=4, step_per_epoch=320,
                                 update_per_step=0.2, batch_size=64, verbose=False)
    assert ep == 2 and np.isfinite(list(stat.values())).all() and "loss/q_total" in stat and "loss/alpha_value" in stat
    a1, alpha = agent.policy.engine.sac_get_params(0)
    assert np.abs(a1 - a0).max() > 1e-4 and 0 < alpha < 1.0


def test_device_actor_sampling_statistics_and_collector_path(tmp_path):
    """fsrl_actor_sample: deterministic == the actor mean; stochastic draws have the policy's mean and
    sigma; the collector's device_actor path trains end to end with it."""
    from fsrl_amd.agent import PPOLagAgent, SACLagAgent
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    from fsrl_amd.utils import BaseLogger
    env = SyntheticSafetyVectorEnv(env_num=4, episode_len=30, seed=2)
    agent = PPOLagAgent(env, BaseLogger(str(tmp_path), name="d"), cost_limit=10, device="cuda:0", seed=1,
                        hidden_sizes=(64, 64), training_num=4)
    eng = agent.policy.engine
    obs = np.random.default_rng(0).standard_normal((3, eng.cfg.obs_dim)).astype(np.float32)
    mu, sigma = eng.actor_forward(obs)
    assert np.array_equal(eng.actor_sample(obs, deterministic=True), mu)
    draws = np.stack([eng.actor_sample(obs, seed=7 if i == 0 else 0) for i in range(2000)])
    n = draws.shape[0]
    assert np.abs(draws.mean(0) - mu).max() < 5 * sigma.max() / np.sqrt(n)
    assert np.abs(draws.std(0) / sigma - 1).max() < 0.1
    ep, stat, info = agent.learn(env, None, epoch=2, episode_per_collect=4, step_per_epoch=240, repeat_per_collect=2,
                                 batch_size=64, verbose=False, save_ckpt=False, device_actor=True)
    assert ep == 2 and np.isfinite(list(stat.values())).all()
    sac = SACLagAgent(env, BaseLogger(str(tmp_path), name="e"), cost_limit=10, device="cuda:0", seed=1,
                      hidden_sizes=(64, 64), training_num=4, buffer_size=2000)
    a = sac.policy.engine.actor_sample(obs)
    assert a.shape == (3, sac.policy.engine.cfg.act_dim) and (np.abs(a) < 1).all()
    ep, stat, info = sac.learn(env, None, epoch=1, episode_per_collect=4, step_per_epoch=240, update_per_step=0.2,
                               batch_size=32, verbose=False, save_ckpt=False, device_actor=True)
    assert ep == 1 and np.isfinite(list(stat.values())).all()


def test_offpolicy_state_dict_keys_and_shapes_match_reference_manifest():
    """Checkpoints written by the facade load into the reference's SACLagrangian / DDPGLagrangian and back:
    same key order and shapes as the unmodified reference builds (tests/golden/gen_manifest_offpolicy.py)."""
    import json, os
    from fsrl_amd.env import Box
    from torch.distributions import Independent, Normal
    from fsrl_amd.policy import CVPO, DDPGLagrangian, SACLagrangian
    from fsrl_amd.utils.net import Actor, ActorProb, Critic, DoubleCritic, Net, SingleCritic
    man = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_manifest.json")))
    Do, Da, h = 6, 3, (64, 64)
    sp = dict(observation_space=Box(-np.inf, np.inf, (Do, )), action_space=Box(-1, 1, (Da, )), device=0, env_num=2,
              cost_limit=10.0, buffer_size=256)
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), conditioned_sigma=True, unbounded=True)
    critics = [DoubleCritic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True), Net((Do, ), (Da, ), hidden_sizes=h, concat=True))
               for _ in range(2)]
    la = torch.zeros(1, requires_grad=True)
    sac = SACLagrangian(actor, critics, torch.optim.Adam(actor.parameters(), lr=1e-3),
                        torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=1e-3),
                        alpha=(-3.0, la, torch.optim.Adam([la], lr=1e-3)), **sp)
    actor2 = Actor(Net((Do, ), hidden_sizes=h), (Da, ))
    critics2 = [Critic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True)) for _ in range(2)]
    ddpg = DDPGLagrangian(actor2, critics2, torch.optim.Adam(actor2.parameters(), lr=1e-3),
                          torch.optim.Adam(torch.nn.ModuleList(critics2).parameters(), lr=1e-3), **sp)
    actor3 = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), conditioned_sigma=True, unbounded=False)
    critics3 = [SingleCritic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True)) for _ in range(2)]
    cvpo = CVPO(actor3, critics3, torch.optim.Adam(actor3.parameters(), lr=1e-3),
                torch.optim.Adam(torch.nn.ModuleList(critics3).parameters(), lr=1e-3), action_space=sp["action_space"],
                dist_fn=lambda *l: Independent(Normal(*l), 1), max_episode_steps=100, cost_limit=10.0,
                observation_space=sp["observation_space"], device=0, env_num=2, buffer_size=256)
    for pol, key in ((sac, "sac_lag_64x64_obs6_act3"), (ddpg, "ddpg_lag_64x64_obs6_act3"), (cvpo, "cvpo_64x64_obs6_act3")):
        sd = pol.state_dict()
        assert [k for k, _ in man[key]] == list(sd.keys()), key
        for k, shape in man[key]:
            if shape is not None:
                assert list(sd[k].shape) == shape, (key, k)
        # round trip: perturb the targets in the checkpoint, load, and read them back from the device
        tk = [k for k in sd if k.startswith("critics_old.") and k.endswith("weight")][0]
        sd2 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in sd.items()}
        sd2[tk] = sd2[tk] + 0.25
        pol.load_state_dict(sd2)
        assert np.array_equal(pol.engine.sac_get_params(2)[0], SACLagrangian._flat(list(pol.critics_old)))
        assert np.array_equal(pol.engine.sac_get_params(1)[0], SACLagrangian._flat(list(pol.critics)))
        assert not np.array_equal(pol.engine.sac_get_params(1)[0], pol.engine.sac_get_params(2)[0])
        pol.engine.close()


def test_agents_learn_the_synthetic_task_under_the_cost_constraint(tmp_path):
    """End-to-end learning smoke test in the spirit of the reference's tests/test_all_agents.py (which trains on
    SafetyBallRun-v0; no simulator in this image): on the synthetic vector env reward = s0*a0 - 0.1|a|^2 + 0.5 and
    cost = [|s1| > 1] are both controllable.  Judged on the TRAINING statistics of the last epochs (hundreds of episodes of
    the stochastic policy; a 4-episode deterministic evaluation is a coin flip on this task): the episode cost is driven
    to the limit of 20 from an untrained ~50, and the reward rises meanwhile."""
    from fsrl_amd.agent import CPOAgent, PPOLagAgent
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    from fsrl_amd.utils import BaseLogger
    # PPO-Lag with the gradient-norm clip of the reference's training config (ppol_cfg.py:21)
    for name, cls, ak, lk in (("ppol", PPOLagAgent, dict(max_grad_norm=0.5), dict(repeat_per_collect=4, batch_size=256)),
                              ("cpo", CPOAgent, {}, dict(repeat_per_collect=2, batch_size=99999))):
        env = SyntheticSafetyVectorEnv(env_num=10, episode_len=100, seed=0)
        test = SyntheticSafetyVectorEnv(env_num=2, episode_len=100, seed=5)
        agent = cls(env, BaseLogger(None), cost_limit=20, device="cuda:0", seed=1, hidden_sizes=(64, 64), training_num=10, **ak)
        _, _, c0 = agent.evaluate(test, eval_episodes=4)
        hist = []
        for _ in range(4):                                           # 4 x 5 epochs, last epoch's training means each time
            _, stat, _ = agent.learn(env, None, epoch=5, episode_per_collect=10, step_per_epoch=2000, verbose=False,
                                     save_ckpt=False, device_actor=True, **lk)
            hist.append((stat["train/reward"], stat["train/cost"]))
        assert c0 > 30, (name, c0)                                   # the untrained policy violates the limit of 20
        if name == "cpo":       # trust-region projection: hovers at the limit within epochs (one epoch = 20 episodes, +-8 of noise)
            assert np.mean([h[1] for h in hist[-3:]]) <= 1.3 * 20 and max(h[1] for h in hist) < 0.75 * c0, (name, hist, c0)
        else:                                                        # PID multiplier: swings around the limit on its way down
            assert hist[-1][1] <= 0.75 * hist[0][1] and hist[-1][1] <= 1.6 * 20, (name, hist)
        assert hist[-1][0] > hist[0][0] + 30, (name, hist)           # and the reward keeps rising under it
        if name == "ppol":
            assert agent.policy.lag_optims[0].get_lag() > 0
        agent.policy.engine.close()


def test_fused_collect_step_equals_actor_sample_plus_push(tmp_path):
    """fsrl_collect_step (one library call per vector step: store the finished rows while the actor for the next
    observations is in flight, noise, map_action) against fsrl_actor_sample + fsrl_store_push + the Python map_action:
    same library random stream, so the collects -- statistics, stored rows, the update that consumes them -- are identical,
    including the surplus-env dropping at the end of a collect (n_episode not a multiple of env_num)."""
    from fsrl_amd.agent import PPOLagAgent, SACLagAgent
    from fsrl_amd.data import FastCollector, HipVectorReplayBuffer
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    from fsrl_amd.utils import BaseLogger
    for cls, kw in ((PPOLagAgent, {}), (SACLagAgent, {})):
        out = []
        for fused in (True, False):
            env = SyntheticSafetyVectorEnv(env_num=6, obs_dim=8, act_dim=2, episode_len=23, seed=4)
            agent = cls(env, BaseLogger(str(tmp_path), name=f"c{fused}"), cost_limit=10, device="cuda:0", seed=2,
                        hidden_sizes=(64, 64), training_num=6, **kw)
            agent.policy.train()
            eng = agent.policy.engine
            buf = HipVectorReplayBuffer(eng, 6 * 200, 6)
            col = FastCollector(agent.policy, env, buf, exploration_noise=True, device_actor=True, fused_step=fused)
            eng.actor_sample(np.zeros((1, 8), np.float32), seed=77)            # key the library stream identically
            st = [col.collect(n_episode=n) for n in (6, 10, 3)]
            rows = eng.sample0()
            sizes = buf._sizes.copy()
            if cls is PPOLagAgent:
                upd, _ = eng.ppo_update([0.3], 1 / 1.3, 64, 2, seed=5)
            else:
                upd = eng.sac_update(32, [0.3], 1 / 1.3, seed=5)
            out.append((st, rows, sizes, np.asarray(upd)))
            eng.close()
        (st_a, rows_a, sz_a, upd_a), (st_b, rows_b, sz_b, upd_b) = out
        assert st_a == st_b, (st_a, st_b)
        assert np.array_equal(rows_a, rows_b) and np.array_equal(sz_a, sz_b)
        assert np.array_equal(upd_a, upd_b)
        assert sum(s["n/ep"] for s in st_a) == 19


def test_bench_multi_rank_path_on_one_gpu():
    """bench.py's N > 1 path (one process per rank, barrier + max-over-ranks timing, rank 0 prints the one JSON line
    with the whole-job value) run as two ranks sharing this box's GPU over gloo: the launch line is the driver's."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--share-gpu"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]      # whole-job = ranks / step time
    assert "roofline" in d and "cpu_baseline" not in d                               # CPU leg: rank 0 at N = 1 only
    assert d["ranks_seen"] == 2 and len(d["per_rank_updates_per_s"]) == 2              # the 8(e) exchange saw both ranks
    assert d["value"] <= d["sum_of_rank_rates"] * (1 + 1e-9)                           # max-over-ranks time <= any rank's own
    job = d["end_to_end_job"]                                                          # env-steps/s of the whole job
    assert job["ranks"] == 2 and len(job["per_rank_env_steps_per_s"]) == 2
    assert abs(job["env_steps_per_s"] - sum(job["per_rank_env_steps_per_s"])) < 1.0 and job["env_steps_per_s"] > 1000


def test_bench_falls_back_to_gloo_when_rccl_cannot_start():
    """Two ranks on ONE GPU with the driver's default backend: RCCL refuses two ranks on one device, the same way on both ranks
    -- the failure mode of a node whose RCCL cannot start.  bench.py must still print its line: the barrier and the
    max-over-ranks time go over gloo (the ranks' work has no data-path collective) and the line says so."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", FSRL_BENCH_LEG_BUDGET_S="60")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--share-gpu"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["ranks_seen"] == 2
    assert d["timing_exchange"].startswith("gloo; nccl could not start here"), d["timing_exchange"]


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_bench_eight_ranks_dry_run_on_one_gpu(backend):
    """The driver's SCALE run is one shot at 8 ranks: this is its dry run on ONE device (--share-gpu) -- eight processes, eight
    engines and stores in this GPU's HBM, eight training loops on their core slices (the box may have fewer usable CPUs than
    ranks), the watchdog budgets, the gloo control plane.  `gloo`: the exchange over gloo as asked.  `nccl`: the driver's default
    -- RCCL refuses eight ranks on one device, every rank agrees on the fallback over the control plane (no second rendezvous on
    the same port) and the line says so.  Exactly ONE JSON line either way, with the whole job's env-steps/s in the tail key."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", FSRL_BENCH_LEG_BUDGET_S="150")
    port = {"gloo": "29551", "nccl": "29553"}[backend]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--share-gpu"]
    if backend == "gloo":
        cmd += ["--backend", "gloo"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 8 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]      # whole-job = ranks / max-over-ranks step time
    assert d["ranks_seen"] == 8 and len(d["per_rank_updates_per_s"]) == 8
    assert d["timing_exchange"].startswith("gloo"), d["timing_exchange"]
    if backend == "nccl":
        assert "nccl could not start here" in d["timing_exchange"], d["timing_exchange"]
    job = d["end_to_end_job"]
    assert job["ranks"] == 8 and job["ranks_ok"] == 8 and len(job["per_rank_env_steps_per_s"]) == 8, job
    assert job["host_cores_per_rank"] >= 1 and job["env_steps_per_s"] > 1000
    assert list(d)[-1] == "configs" and len(json.dumps(d["configs"])) <= 600
    tail = d["configs"]["job"]
    assert tail["ranks_ok"] == 8 and abs(tail["env_steps_s"] - job["env_steps_per_s"]) <= 1.0 and tail["updates_s"] > 0


def test_bench_headline_survives_a_failing_leg():
    """a secondary leg that raises on one code path must not cost the headline line.  Two ranks over
    gloo sharing this GPU, the per-rank training-loop leg forced to fail on every rank: the line is still printed, the
    headline fields are intact, the job figure says no rank contributed."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", FSRL_BENCH_FAIL_LEG="end_to_end_rank,grouped")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--share-gpu"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["ranks_seen"] == 2 and d["roofline"]["frac"] > 0
    assert d["end_to_end_job"]["ranks"] == 2 and d["end_to_end_job"]["ranks_ok"] == 0


def test_multi_gpu_launcher_two_ranks_share_this_gpu(tmp_path):
    """examples/train_multi_gpu.py under the driver's launch line, two ranks on this box's one GPU over gloo: two real
    (tiny) PPO-Lag training loops, seeds base + rank, the epoch vector all-reduced from BaseTrainer._close_epoch.  The
    pooled job reward must be the mean of the two ranks' own last-epoch rewards (same episode count per rank here)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "examples", "train_multi_gpu.py"), "--algo", "ppol", "--envs", "4",
           "--epoch", "2", "--step-per-epoch", "480", "--episode-len", "40", "--hidden", "64", "--backend", "gloo",
           "--share-gpu", "--json", "--seed", "5", "--logdir", str(tmp_path)]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["ranks"] == 2 and d["epochs"] == 2 and sorted(r["seed"] for r in d["per_seed"]) == [5.0, 6.0]
    job = d["job_last_epoch"]
    assert job["job/ranks"] == 2.0 and job["job/env_step"] == 2 * 480 and job["job/episodes"] == 2 * 12
    mean_of_ranks = sum(r["reward"] for r in d["per_seed"]) / 2
    assert abs(job["job/reward"] - mean_of_ranks) <= 1e-9 * max(1.0, abs(mean_of_ranks))
    assert abs(job["job/cost"] - sum(r["cost"] for r in d["per_seed"]) / 2) <= 1e-9 * 100
    assert d["per_seed"][0]["reward"] != d["per_seed"][1]["reward"]                  # two different seeds really ran
    assert job["job/env_steps_per_s"] > 0 and job["job/updates_per_s"] > 0
    for seed in (5, 6):                                                              # every rank kept its own curve
        assert os.path.exists(os.path.join(str(tmp_path), f"ppol-seed{seed}", "progress.txt"))


@pytest.mark.parametrize("which", ["sac", "ddpg", "cvpo"])
def test_offpolicy_state_dict_after_a_host_forward_holds_the_device_critics(which, tmp_path):
    """update -> policy(batch) (the host forward refreshes the ACTOR mirror only) -> state_dict(): the checkpoint must
    still carry the device's critics and targets, not the mirror's stale ones (what agent.learn() with a test collector
    and agent.evaluate() do between updates and logger.save_checkpoint())."""
    from fsrl_amd.agent import CVPOAgent, DDPGLagAgent, SACLagAgent
    from fsrl_amd.data import Batch, FastCollector, HipVectorReplayBuffer
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    from fsrl_amd.policy.sac_lag import SACLagrangian
    cls = {"sac": SACLagAgent, "ddpg": DDPGLagAgent, "cvpo": CVPOAgent}[which]
    env = SyntheticSafetyVectorEnv(env_num=4, episode_len=30, seed=2)
    agent = cls(env, None, cost_limit=10, device="cuda:0", seed=1, hidden_sizes=(64, 64), training_num=4, buffer_size=2000)
    pol, eng = agent.policy, agent.policy.engine
    pol.train()
    buf = HipVectorReplayBuffer(eng, 2000, 4)
    FastCollector(pol, env, buf, exploration_noise=True).collect(n_episode=8)
    c0 = eng.sac_get_params(1)[0].copy()
    pol.pre_update_fn(stats_train={"cost": 20.0})
    for _ in range(5):
        pol.update(64, buf)
    pol.post_update_fn(stats_train={"cost": 20.0})
    assert np.abs(eng.sac_get_params(1)[0] - c0).max() > 1e-5          # the critics did move on the device
    pol(Batch(obs=np.zeros((3, eng.cfg.obs_dim), np.float32), info={}))  # actor-only refresh of the host mirror
    sd = pol.state_dict()
    assert np.array_equal(SACLagrangian._flat(list(pol.critics)), eng.sac_get_params(1)[0])
    assert np.array_equal(SACLagrangian._flat(list(pol.critics_old)), eng.sac_get_params(2)[0])
    assert np.array_equal(SACLagrangian._flat([pol.actor]), eng.sac_get_params(0)[0])
    k = [k for k in sd if k.startswith("critics.") and k.endswith("weight")][0]
    assert np.array_equal(sd[k].numpy(), dict(pol.named_parameters())[k].detach().numpy())
    eng.close()


def test_lr_scheduler_rate_reaches_the_engine(tmp_path):
    """BasePolicy.update steps lr_scheduler last (base_policy.py:352-354); the engine must see the new rate."""
    from fsrl_amd.agent import PPOLagAgent, SACLagAgent
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    env = SyntheticSafetyVectorEnv(env_num=4, episode_len=30, seed=2)
    agent = PPOLagAgent(env, None, cost_limit=10, device="cuda:0", seed=1, hidden_sizes=(64, 64), training_num=4, lr=1e-3)
    pol = agent.policy
    pol.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(pol.optim, lambda e: 0.5 ** e)
    assert abs(pol.engine.get_lr(0) - 1e-3) < 1e-9                 # the engine keeps float32 rates
    agent.learn(env, None, epoch=1, episode_per_collect=4, step_per_epoch=240, repeat_per_collect=1, batch_size=64,
                verbose=False, save_ckpt=False, show_progress=False)
    n_updates = pol.lr_scheduler.last_epoch
    assert n_updates >= 1
    assert abs(pol.engine.get_lr(0) - 1e-3 * 0.5 ** n_updates) < 1e-10
    assert abs(pol.optim.param_groups[0]["lr"] - pol.engine.get_lr(0)) < 1e-10
    with pytest.raises(AssertionError):
        pol.engine.set_lr(1, 1e-3)                                      # one optimiser: group 0 only
    pol.engine.close()
    sac = SACLagAgent(env, None, cost_limit=10, device="cuda:0", seed=1, hidden_sizes=(64, 64), training_num=4,
                      buffer_size=2000, actor_lr=4e-4, critic_lr=2e-3)
    e = sac.policy.engine
    assert abs(e.get_lr(0) - 4e-4) < 1e-10 and abs(e.get_lr(1) - 2e-3) < 1e-10
    e.set_lr(1, 5e-4)
    assert abs(e.get_lr(1) - 5e-4) < 1e-10
    e.close()


def test_metrics_allreduce_through_the_c_abi():
    """fsrl_metrics_allreduce: the identity without a communicator; with the library's RCCL communicator of world size 1
    (ncclGetUniqueId -> ncclCommInitRank, reached through dlopen) the sum of one rank's vector is that vector.  World sizes
    above 1 need one GPU per rank (RCCL refuses two ranks on one device): the driver's multi-GPU node."""
    from fsrl_amd import parallel
    from fsrl_amd.engine import Engine, EngineConfig
    eng = Engine(EngineConfig(hidden=64, env_num=2))
    v = np.arange(17, dtype=np.float64) * 1.5
    assert np.array_equal(eng.metrics_allreduce(v), v) and eng.comm_info() == (0, 1)
    uid = eng.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    eng.comm_init(0, 1, uid)
    assert eng.comm_info() == (0, 1)
    assert np.array_equal(eng.metrics_allreduce(v), v)
    with pytest.raises(AssertionError):
        eng.metrics_allreduce(np.zeros(65))                         # at most 64 metrics
    with pytest.raises(AssertionError):
        eng.comm_init(2, 2, uid)                                    # rank outside the world
    job = parallel.reduce_epoch({"n_st": 100.0, "n_ep": 4.0, "sum_rew": 40.0, "sum_cost": 8.0, "duration": 2.0}, eng)
    assert job["job/ranks"] == 1.0 and job["job/reward"] == 10.0 and job["job/env_steps_per_s"] == 50.0
    eng.comm_destroy()
    eng.close()


@pytest.mark.parametrize("kind", ["ppol", "cpo", "trpol", "focops"])
def test_agents_accept_the_on_policy_options_of_the_reference(kind, tmp_path):
    """unbounded actor head + reward_normalization (+ value_clip and recompute_advantage where the reference has them) through
    Agent.__init__ -> learn(): the arguments reach the engine, the running return statistics advance once per
    compute_gae_returns call, training stays finite."""
    from fsrl_amd.agent import CPOAgent, FOCOPSAgent, PPOLagAgent, TRPOLagAgent
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    from fsrl_amd.utils import BaseLogger
    cls, kw, lk = {"ppol": (PPOLagAgent, dict(value_clip=True, recompute_advantage=True, max_grad_norm=0.5), dict(repeat_per_collect=3, batch_size=64)),
                   "cpo": (CPOAgent, {}, dict(repeat_per_collect=2, batch_size=99999)),
                   "trpol": (TRPOLagAgent, {}, dict(repeat_per_collect=2, batch_size=99999)),
                   "focops": (FOCOPSAgent, dict(recompute_advantage=True), dict(repeat_per_collect=3, batch_size=64))}[kind]
    env = SyntheticSafetyVectorEnv(env_num=4, episode_len=50, seed=1)
    agent = cls(env, BaseLogger(str(tmp_path), name="t"), cost_limit=10, device="cuda:0", seed=3, hidden_sizes=(64, 64),
                training_num=4, unbounded=True, reward_normalization=True, **kw)
    cfg = agent.policy.engine.cfg
    assert cfg.unbounded and cfg.rew_norm and agent.policy.actor._unbounded
    assert cfg.value_clip == (kind == "ppol") and cfg.recompute_adv == (kind in ("ppol", "focops"))
    ep, stat, info = agent.learn(env, None, epoch=2, episode_per_collect=4, step_per_epoch=400, verbose=False, save_ckpt=False, **lk)
    assert ep == 2 and np.isfinite([v for v in stat.values() if isinstance(v, (int, float))]).all()
    rms = agent.policy.ret_rms
    collects = 2 * 2                                  # 2 epochs x (400 steps / (4 episodes x 50 steps)) collects
    per_update = 3 if kind in ("ppol", "focops") else 1   # recompute_advantage: once more before passes 2 and 3 (fewer after
    assert len(rms) == 2 and rms[0].count == rms[1].count and rms[0].count % 200 == 0       # a KL early stop)
    assert collects * 200 <= rms[0].count <= collects * per_update * 200, [r.count for r in rms]
    assert all(r.var > 0 and np.isfinite(r.mean) for r in rms)
    mu = agent.policy.engine.actor_forward(np.full((1, 8), 3.0, np.float32))[0]
    assert np.isfinite(mu).all()
    agent.policy.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize("workers", [0, 3])
def test_agent_learns_over_env_factories(workers, tmp_path):
    """The reference's own construction (train_ppol_agent.py:120-123): a list of env factories wrapped in DummyVectorEnv /
    ShmemVectorEnv.  Per-instance gym-style envs (obs 6, act 2, info["cost"]) -- in process, and in 3 worker processes with the
    device actor (native collector loop over the shared block)."""
    from fsrl_amd.agent import PPOLagAgent
    from fsrl_amd.env import DummyVectorEnv, PointCircleEnv, ShmemVectorEnv
    from fsrl_amd.utils import BaseLogger
    fns = [lambda: PointCircleEnv(max_episode_steps=40) for _ in range(6)]
    train = ShmemVectorEnv(fns, workers=workers, seed=1) if workers else DummyVectorEnv(fns, seed=1)
    test = DummyVectorEnv(fns[:2], seed=100)
    try:
        agent = PPOLagAgent(train, BaseLogger(str(tmp_path), name="t"), cost_limit=5, device="cuda:0", seed=3,
                            hidden_sizes=(64, 64), max_grad_norm=0.5, training_num=6)
        theta0 = agent.policy.engine.get_params()
        ep, stat, info = agent.learn(train, test, epoch=2, episode_per_collect=6, step_per_epoch=480, repeat_per_collect=2,
                                     batch_size=64, testing_num=2, verbose=False, save_ckpt=False, device_actor=bool(workers))
        assert ep == 2 and stat["train/length"] <= 40 and np.isfinite(list(stat.values())).all()
        assert np.abs(agent.policy.engine.get_params() - theta0).max() > 1e-4
        rew, length, cost = agent.evaluate(PointCircleEnv(max_episode_steps=40), eval_episodes=2)      # a bare env is wrapped
        assert 0 < length <= 40 and np.isfinite(rew) and cost >= 0
    finally:
        train.close()


@pytest.mark.parametrize("kind", ["ppol", "cpo", "sacl", "ddpgl", "cvpo", "focops"])
def test_agents_with_two_hidden_layers_of_any_width(kind, tmp_path):
    """`hidden_sizes: Tuple[int, ...]` of the agents (fsrl/agent/ppo_lag_agent.py:91,136): any two widths up to 256 run on the
    HIP path (the kernels' width is the next of 64 / 128 / 256, narrower layers are zero-padded: include/fsrl_hip.h hidden1 /
    hidden2).  Train a little, then: the state_dict has the caller's shapes, the device parameters round-trip through it, and
    three hidden layers / a 300-wide layer make a layered context for every agent."""
    from fsrl_amd import agent as A
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    from fsrl_amd.utils import BaseLogger
    cls = {"ppol": A.PPOLagAgent, "cpo": A.CPOAgent, "sacl": A.SACLagAgent, "ddpgl": A.DDPGLagAgent, "cvpo": A.CVPOAgent,
           "focops": A.FOCOPSAgent}[kind]
    env = SyntheticSafetyVectorEnv(env_num=4, episode_len=40, seed=1)
    kw = dict(buffer_size=4000) if kind in ("sacl", "ddpgl", "cvpo") else {}
    agent = cls(env, BaseLogger(str(tmp_path), name="t"), cost_limit=10, device="cuda:0", seed=3, hidden_sizes=(100, 50),
                training_num=4, **kw)
    pol = agent.policy
    w = {k: tuple(v.shape) for k, v in pol.state_dict().items() if k.startswith("actor.preprocess")}
    assert w["actor.preprocess.model.model.0.weight"] == (100, 8) and w["actor.preprocess.model.model.2.weight"] == (50, 100)
    if kind in ("sacl", "ddpgl", "cvpo"):
        agent.learn(env, None, epoch=1, episode_per_collect=4, step_per_epoch=320, update_per_step=0.2, batch_size=64,
                    verbose=False, save_ckpt=False)
    else:
        agent.learn(env, None, epoch=1, episode_per_collect=4, step_per_epoch=320, repeat_per_collect=2,
                    batch_size=64 if kind in ("ppol", "focops") else 99999, verbose=False, save_ckpt=False)
    import copy
    sd = copy.deepcopy(pol.state_dict())
    assert all(torch.isfinite(v).all() for v in sd.values() if torch.is_tensor(v) and v.is_floating_point())
    pol.load_state_dict(sd)                                    # host -> device -> host: the padding is invisible
    if kind in ("sacl", "ddpgl", "cvpo"):
        pol._dirty = pol._rest_dirty = True                   # the next state_dict() reads the device back
    else:
        pol._mark_stale()
    for k, v in pol.state_dict().items():
        if torch.is_tensor(v):
            assert torch.equal(v, sd[k]), k
    rew, length, cost = agent.evaluate(env, eval_episodes=2)
    assert length == 40.0 and np.isfinite(rew)
    for other in ((64, 64, 64), (300, 64)):
        # every agent takes any tuple: a layered context (include/fsrl_hip.h fsrl_config.n_hidden)
        a2 = cls(env, None, cost_limit=10, device="cuda:0", seed=3, hidden_sizes=other, training_num=4, **kw)
        if kind in ("ppol", "focops", "cpo"):
            assert a2.policy.engine.n_params == sum(p.numel() for p in a2.policy._actor_critic.parameters())


@pytest.mark.parametrize("kind", ["ppo", "cpo"])
def test_process_fn_then_learn_is_update(kind):
    """fsrl/policy/base_policy.py:332-355: update() = process_fn -> learn.  On the HIP path process_fn returns a DeviceBatch (the
    processed batch stays in HBM: values / rets / advs / logp_old on demand) and learn takes it back; the two calls in a row leave
    exactly the parameters one update() call leaves, and learn refuses anything that is not the pending batch."""
    from torch.distributions import Independent, Normal
    from fsrl_amd.data import Batch, HipVectorReplayBuffer
    from fsrl_amd.env import Box, SyntheticSafetyVectorEnv
    from fsrl_amd.policy import CPO, PPOLagrangian
    from fsrl_amd.policy.base_policy import DeviceBatch
    from fsrl_amd.utils.net import ActorCritic, ActorProb, Critic, Net
    from test_gpu_loop import _Cap, _rollout
    Do, Da, E, T = 8, 2, 4, 60

    def make():
        torch.manual_seed(3)
        actor = ActorProb(Net((Do, ), hidden_sizes=(64, 64)), (Da, ), max_action=1.0)
        critics = [Critic(Net((Do, ), hidden_sizes=(64, 64))) for _ in range(2)]
        common = dict(logger=_Cap(), cost_limit=10.0, observation_space=Box(-np.inf, np.inf, (Do, )), action_space=Box(-1, 1, (Da, )),
                      device=0, env_num=E, buffer_size=E * T * 2)
        if kind == "ppo":
            pol = PPOLagrangian(actor, critics, torch.optim.Adam(ActorCritic(actor, critics).parameters(), lr=5e-4),
                                lambda *l: Independent(Normal(*l), 1), target_kl=None, max_grad_norm=0.5, **common)
        else:
            pol = CPO(actor, critics, torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=1e-3),
                      lambda *l: Independent(Normal(*l), 1), optim_critic_iters=3, **common)
        pol.train()
        env = SyntheticSafetyVectorEnv(env_num=E, obs_dim=Do, act_dim=Da, episode_len=T, seed=5)
        buf = HipVectorReplayBuffer(pol.engine, E * T * 2, E)
        torch.manual_seed(11); np.random.seed(11)
        st = _rollout(pol, env, buf)
        pol.pre_update_fn(stats_train={"cost": st["cost"]})
        np.random.seed(12)
        return pol, buf
    kw = dict(batch_size=64, repeat=2) if kind == "ppo" else dict(batch_size=99999, repeat=2)
    a, buf_a = make()
    res_a = a.update(0, buf_a, **kw)
    theta_a = a.engine.get_params().copy()
    b, buf_b = make()
    batch = b.process_fn(None, buf_b, None, **({"batch_size": 64} if kind == "ppo" else {}))
    assert isinstance(batch, DeviceBatch) and len(batch) == E * T and b.updating
    assert tuple(batch.advs.shape) == (E * T, 2) and tuple(batch.rets.shape) == (E * T, 2) and tuple(batch.logp_old.shape) == (E * T, )
    assert torch.isfinite(batch.advs).all() and torch.isfinite(batch.values).all()
    with pytest.raises(AssertionError):
        b.learn(Batch(obs=np.zeros((1, Do), np.float32)), **kw)          # not the pending device batch
    res_b = b.learn(batch, **kw)
    assert res_b["gradient_steps"] == res_a["gradient_steps"] > 0
    assert np.array_equal(b.engine.get_params(), theta_a)
    with pytest.raises(AssertionError):
        b.learn(batch, **kw)                                              # consumed: a second learn needs a new process_fn
    a.engine.close(); b.engine.close()


@pytest.mark.parametrize("ref_rng", [True, False])
@pytest.mark.parametrize("kind", ["sac", "ddpg", "cvpo"])
def test_replay_process_fn_then_learn_is_update(kind, ref_rng):
    """fsrl/policy/base_policy.py:332-355 for the replay agents: update() = buffer.sample -> process_fn -> learn.  process_fn
    returns a ReplayDeviceBatch (the sample: indices + the noise the reference's `_target_q` forward drew; the rows stay in the HBM
    store), learn takes it back and runs the fused update; the calls made by hand leave exactly the parameters update() leaves."""
    from torch.distributions import Independent, Normal
    from fsrl_amd.data import Batch, HipVectorReplayBuffer
    from fsrl_amd.env import Box, SyntheticSafetyVectorEnv
    from fsrl_amd.policy import CVPO, DDPGLagrangian, SACLagrangian
    from fsrl_amd.policy.base_policy import ReplayDeviceBatch
    from fsrl_amd.utils.net import Actor, ActorProb, Critic, DoubleCritic, Net, SingleCritic
    from test_gpu_loop import _Cap, _rollout
    Do, Da, h, E, T, B = 6, 3, (64, 64), 4, 50, 64

    def make():
        torch.manual_seed(3)
        sp = dict(observation_space=Box(-np.inf, np.inf, (Do, )), action_space=Box(-1, 1, (Da, )), device=0, env_num=E,
                  cost_limit=10.0, buffer_size=E * T * 2, reference_rng=ref_rng, logger=_Cap())
        if kind == "sac":
            actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), conditioned_sigma=True, unbounded=True)
            critics = [DoubleCritic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True), Net((Do, ), (Da, ), hidden_sizes=h, concat=True))
                       for _ in range(2)]
            la = torch.zeros(1, requires_grad=True)
            pol = SACLagrangian(actor, critics, torch.optim.Adam(actor.parameters(), lr=1e-3),
                                torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=1e-3),
                                alpha=(-3.0, la, torch.optim.Adam([la], lr=1e-3)), n_step=2, **sp)
        elif kind == "ddpg":
            actor = Actor(Net((Do, ), hidden_sizes=h), (Da, ))
            critics = [Critic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True)) for _ in range(2)]
            pol = DDPGLagrangian(actor, critics, torch.optim.Adam(actor.parameters(), lr=1e-3),
                                 torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=1e-3), n_step=2, **sp)
        else:
            actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), conditioned_sigma=True, unbounded=False)
            critics = [SingleCritic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True)) for _ in range(2)]
            pol = CVPO(actor, critics, torch.optim.Adam(actor.parameters(), lr=1e-3),
                       torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=1e-3),
                       dist_fn=lambda *l: Independent(Normal(*l), 1), max_episode_steps=T, **sp)
        pol.train()
        env = SyntheticSafetyVectorEnv(env_num=E, obs_dim=Do, act_dim=Da, episode_len=T, seed=5)
        buf = HipVectorReplayBuffer(pol.engine, E * T * 2, E)
        torch.manual_seed(11); np.random.seed(11)
        st = _rollout(pol, env, buf, noise=(kind == "ddpg"))
        pol.pre_update_fn(stats_train={"cost": st["cost"]})
        torch.manual_seed(12); np.random.seed(12)
        return pol, buf
    a, buf_a = make()
    for _ in range(3):
        a.update(B, buf_a)
    a.post_update_fn(stats_train={"cost": 0.0})
    want = [a.engine.sac_get_params(w)[0].copy() for w in (0, 1, 2)]
    b, buf_b = make()
    for _ in range(3):
        indices = buf_b.sample_indices(B) if ref_rng else None                 # BasePolicy.update's buffer.sample
        batch = b.process_fn(None, buf_b, indices, sample_size=B)
        assert isinstance(batch, ReplayDeviceBatch) and len(batch) == B and b.updating
        assert (batch.indices is None) == (not ref_rng)
        b.learn(batch)
        b._step_lr_scheduler(); b.updating = False
    b.post_update_fn(stats_train={"cost": 0.0})
    assert b.gradient_steps == a.gradient_steps == 3
    for w, t in zip((0, 1, 2), want):
        got = b.engine.sac_get_params(w)[0]
        assert np.isfinite(got).all() and np.array_equal(got, t), (kind, ref_rng, w)
    assert not np.array_equal(want[0], make()[0].engine.sac_get_params(0)[0])  # the updates moved the actor
    with pytest.raises(AssertionError):
        b.learn(Batch(obs=np.zeros((1, Do), np.float32)))                      # not a ReplayDeviceBatch
    a.engine.close(); b.engine.close()
