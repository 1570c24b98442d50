"""the collector's device-actor path pinned against the host torch mirror (the reference's own actor
modules, fsrl/data/fast_collector.py:263-300 -> policy.forward, base_policy.py:178-190 / sac_lag.py:155-183).

  * fsrl_actor_forward / fsrl_sac_actor_forward: (mu, sigma) of the device actor == the host nn.Module on the same
    observations, 1e-6 (fp32, different summation order), every hidden width, ragged row counts;
  * fsrl_collect_step(deterministic = 1): a whole collect with the actor on the device stores the same rows, in the same
    slots, and reports the same statistics as the collector that runs the host mirror (policy.eval(): act = mean, so no
    random stream is involved) -- rows compared through fsrl_store_read."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hidden,obs_dim,act_dim", [(64, 8, 2), (128, 8, 2), (256, 8, 2), (256, 60, 2), (128, 33, 8),
                                                    ((100, 50), 8, 2), ((24, 200), 17, 3)])      # zero-padded widths
def test_actor_forward_equals_host_mirror(hidden, obs_dim, act_dim, tmp_path):
    from fsrl_amd.agent import PPOLagAgent
    from fsrl_amd.data.batch import Batch
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    env = SyntheticSafetyVectorEnv(env_num=4, obs_dim=obs_dim, act_dim=act_dim, episode_len=20, seed=1)
    agent = PPOLagAgent(env, None, cost_limit=10, device="cuda:0", seed=3,
                        hidden_sizes=hidden if isinstance(hidden, tuple) else (hidden, hidden), training_num=4)
    pol, eng = agent.policy, agent.policy.engine
    with torch.no_grad():                       # move sigma_param and the head off their init so every term is exercised
        pol.actor.sigma_param.add_(0.1 * torch.randn_like(pol.actor.sigma_param))
        for p in pol.actor.mu.parameters():
            p.add_(0.05 * torch.randn_like(p))
    pol._push_params()
    rng = np.random.default_rng(0)
    for k in (1, 4, 7, 16, 17, 37, 200):
        obs = (2.0 * rng.standard_normal((k, obs_dim))).astype(np.float32)
        mu, sigma = eng.actor_forward(obs)
        with torch.no_grad():
            (hm, hs), _ = pol.actor(obs)
        np.testing.assert_allclose(mu, hm.numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(sigma, np.broadcast_to(hs.numpy(), sigma.shape), rtol=1e-6, atol=0)
        # deterministic sampling == the mean, bit for bit, and what policy.forward() returns in eval mode
        pol.eval()
        with torch.no_grad():
            act = pol(Batch(obs=obs, info={})).act.numpy()
        assert np.array_equal(eng.actor_sample(obs, deterministic=True), mu)
        np.testing.assert_allclose(mu, act, rtol=0, atol=1e-6)
    eng.close()


@pytest.mark.parametrize("hidden", [64, 256])
def test_sac_and_ddpg_actor_forward_equal_host_mirror(hidden, tmp_path):
    from fsrl_amd.agent import DDPGLagAgent, SACLagAgent
    from fsrl_amd.data.batch import Batch
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    env = SyntheticSafetyVectorEnv(env_num=4, obs_dim=33, act_dim=8, episode_len=20, seed=1)
    rng = np.random.default_rng(1)
    sac = SACLagAgent(env, None, cost_limit=10, device="cuda:0", seed=3, hidden_sizes=(hidden, hidden), training_num=4,
                      buffer_size=400, deterministic_eval=True)
    pol, eng = sac.policy, sac.policy.engine
    for k in (1, 5, 16, 33, 130):
        obs = (1.5 * rng.standard_normal((k, 33))).astype(np.float32)
        mu, sigma = eng.sac_actor_forward(obs)
        with torch.no_grad():
            (hm, hs), _ = pol.actor(obs)
        np.testing.assert_allclose(mu, hm.numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(sigma, hs.numpy(), rtol=2e-6, atol=1e-9)       # sigma = exp(clamp(head)): relative
        pol.eval()
        with torch.no_grad():
            act = pol(Batch(obs=obs, info={})).act.numpy()                       # tanh(mean)
        np.testing.assert_allclose(eng.actor_sample(obs, deterministic=True), act, rtol=0, atol=2e-6)
    eng.close()
    ddpg = DDPGLagAgent(env, None, cost_limit=10, device="cuda:0", seed=3, hidden_sizes=(hidden, hidden), training_num=4,
                        buffer_size=400)
    pol, eng = ddpg.policy, ddpg.policy.engine
    obs = (1.5 * rng.standard_normal((21, 33))).astype(np.float32)
    pol.eval()
    with torch.no_grad():
        act = pol(Batch(obs=obs, info={})).act.numpy()                           # max_action * tanh(MLP)
    np.testing.assert_allclose(eng.actor_sample(obs, deterministic=True), act, rtol=0, atol=2e-6)
    eng.close()


def _collect_and_read(device_actor, agent_cls, kw, n_list):
    from fsrl_amd.data import FastCollector, HipVectorReplayBuffer
    from fsrl_amd.env import SyntheticSafetyVectorEnv
    env = SyntheticSafetyVectorEnv(env_num=6, obs_dim=8, act_dim=2, episode_len=23, seed=4)
    agent = agent_cls(env, None, cost_limit=10, device="cuda:0", seed=2, hidden_sizes=(64, 64), training_num=6, **kw)
    agent.policy.eval()                                  # deterministic_eval: act = the actor's mean
    eng = agent.policy.engine
    buf = HipVectorReplayBuffer(eng, 6 * 200, 6)
    col = FastCollector(agent.policy, env, buf, exploration_noise=False, device_actor=device_actor)
    stats = [col.collect(n_episode=n) for n in n_list]
    idx = eng.sample0()
    rows = eng.store_read(idx)
    eng.close()
    return stats, idx, rows


@pytest.mark.parametrize("which", ["ppo", "sac"])
def test_deterministic_collect_step_rows_equal_host_mirror_collector(which):
    """Same env seed, policy.eval(): FastCollector over fsrl_collect_step vs FastCollector over the host mirror.  n_episode
    values that are and are not multiples of the env count (surplus-env dropping, fast_collector.py:352-366)."""
    from fsrl_amd.agent import PPOLagAgent, SACLagAgent
    cls, kw = (PPOLagAgent, {}) if which == "ppo" else (SACLagAgent, {"buffer_size": 1200, "deterministic_eval": True})
    st_d, idx_d, rows_d = _collect_and_read(True, cls, kw, (6, 10, 3))
    st_h, idx_h, rows_h = _collect_and_read(False, cls, kw, (6, 10, 3))
    assert np.array_equal(idx_d, idx_h)                                       # same slots in the same order
    for a, b in zip(st_d, st_h):
        assert a["n/ep"] == b["n/ep"] and a["n/st"] == b["n/st"] and a["total_cost"] == b["total_cost"]
        assert abs(a["rew"] - b["rew"]) <= 1e-4 and a["len"] == b["len"]
    for k in ("terminated", "truncated", "cost"):
        assert np.array_equal(rows_d[k], rows_h[k]), k
    # the device actor's mean differs from torch's by fp32 summation order (<= 1e-6); the env's linear dynamics are
    # contracting, so rows stay within 1e-5 over the 23-step episodes
    for k in ("obs", "act", "obs_next", "rew"):
        np.testing.assert_allclose(rows_d[k], rows_h[k], rtol=0, atol=1e-5, err_msg=k)


def test_split_phase_collect_stores_the_rows_of_the_plain_fused_loop():
    """FastCollector(split_phase=True) over the two lanes of the worker-process env against the plain fused loop over the
    same env (same seed, policy.eval() so that no noise stream is involved): every env's trajectory depends on its own
    worker's random stream and the actions it is given only, so where both loops keep the SAME envs they must store the
    SAME rows in the SAME slots.  They keep the same envs whenever nothing or everything is dropped at an episode boundary
    (n_episode a multiple of the env count, or below it); otherwise the plain loop drops the first finished envs of the whole
    vector and the split loop the first finished envs of the lane that reports last -- equally exact, different envs -- and
    only the episode accounting is compared."""
    from fsrl_amd.agent import PPOLagAgent
    from fsrl_amd.data import FastCollector, HipVectorReplayBuffer
    from fsrl_amd.env import ShmemVectorEnv

    def run(split, n_list):
        env = ShmemVectorEnv(env_num=6, workers=6, obs_dim=8, act_dim=2, episode_len=17, seed=4)
        try:
            agent = PPOLagAgent(env, None, cost_limit=10, device="cuda:0", seed=2, hidden_sizes=(64, 64), training_num=6)
            agent.policy.eval()
            eng = agent.policy.engine
            buf = HipVectorReplayBuffer(eng, 6 * 300, 6)
            col = FastCollector(agent.policy, env, buf, exploration_noise=False, device_actor=True, split_phase=split)
            assert col.split_phase == split
            stats = [col.collect(n_episode=n) for n in n_list]
            idx = eng.sample0()
            rows = eng.store_read(idx)
            sizes = buf._sizes.copy()
            eng.close()
            return stats, idx, rows, sizes
        finally:
            env.close()

    same = (6, 12, 3, 1, 18)
    st_s, idx_s, rows_s, _ = run(True, same)
    st_p, idx_p, rows_p, _ = run(False, same)
    assert np.array_equal(idx_s, idx_p)
    for a, b in zip(st_s, st_p):
        assert {k: v for k, v in a.items() if k != "rew"} == {k: v for k, v in b.items() if k != "rew"}, (a, b)
        assert abs(a["rew"] - b["rew"]) <= 1e-9 * max(1.0, abs(b["rew"]))          # the mean is taken in a different order
    for k in rows_p:
        assert np.array_equal(rows_s[k], rows_p[k]), k
    # episode-exact collection when envs are dropped mid-way: exactly n whole episodes, whole episodes per env
    ragged = (10, 7, 4, 11)
    st_s, _, rows_s, sizes = run(True, ragged)
    assert [s["n/ep"] for s in st_s] == list(ragged) and all(s["len"] == 17.0 and s["truncated"] == 1.0 for s in st_s)
    assert [s["n/st"] for s in st_s] == [17 * n for n in ragged]
    assert int(sizes.sum()) == 17 * sum(ragged) and (sizes % 17 == 0).all()
    assert int(rows_s["truncated"].sum()) == sum(ragged)


@pytest.mark.parametrize("workers", [1, 3, 6])
def test_native_collector_loop_stores_the_rows_of_the_interpreted_loop(workers):
    """fsrl_collect_episodes (a whole collect(n_episode) inside the library: handshake with the env workers, store, actor,
    resets, episode accounting, surplus envs) and fsrl_collect_run (Python keeps the episode boundaries) against the same loop
    driven step by step from Python: same library noise stream, same env seeds -> the same rows in the same slots, the same
    statistics.  Exploration noise ON: the paths must consume the stream identically."""
    from fsrl_amd.agent import PPOLagAgent
    from fsrl_amd.data import FastCollector, HipVectorReplayBuffer
    from fsrl_amd.env import ShmemVectorEnv

    def run(native):
        env = ShmemVectorEnv(env_num=6, workers=workers, obs_dim=8, act_dim=2, episode_len=19, seed=4)
        try:
            agent = PPOLagAgent(env, None, cost_limit=10, device="cuda:0", seed=2, hidden_sizes=(64, 64), training_num=6)
            agent.policy.train()
            eng = agent.policy.engine
            buf = HipVectorReplayBuffer(eng, 6 * 300, 6)
            col = FastCollector(agent.policy, env, buf, exploration_noise=True, device_actor=True, native_loop=native)
            eng.actor_sample(np.zeros((1, 8), np.float32), seed=77)            # key the library stream identically
            stats = [col.collect(n_episode=n) for n in (6, 10, 3, 1, 7)]
            idx = eng.sample0()
            rows = eng.store_read(idx)
            eng.close()
            return stats, idx, rows
        finally:
            env.close()

    st_p, idx_p, rows_p = run(False)
    for mode in (True, "run"):
        st_n, idx_n, rows_n = run(mode)
        assert np.array_equal(idx_n, idx_p), mode
        for a, b in zip(st_n, st_p):
            assert a == b, (mode, a, b)
        for k in rows_p:
            assert np.array_equal(rows_n[k], rows_p[k]), (mode, k)
    assert rows_p["act"].std() > 0.1                       # the noise was on


@pytest.mark.parametrize("workers", [2, 6])
def test_native_split_phase_collect_stores_the_rows_of_the_interpreted_split_loop(workers):
    """fsrl_collect_episodes_split (the two-lane split-phase collect as ONE library call) against FastCollector._collect_split
    driven from Python: the same library calls in the same order -> the same rows in the same slots, the same noise stream
    (exploration noise ON), the same statistics -- also for episode counts at which envs are dropped mid-way."""
    from fsrl_amd.agent import PPOLagAgent
    from fsrl_amd.data import FastCollector, HipVectorReplayBuffer
    from fsrl_amd.env import ShmemVectorEnv

    def run(native):
        env = ShmemVectorEnv(env_num=6, workers=workers, obs_dim=8, act_dim=2, episode_len=19, seed=4)
        try:
            agent = PPOLagAgent(env, None, cost_limit=10, device="cuda:0", seed=2, hidden_sizes=(64, 64), training_num=6)
            agent.policy.train()
            eng = agent.policy.engine
            buf = HipVectorReplayBuffer(eng, 6 * 300, 6)
            col = FastCollector(agent.policy, env, buf, exploration_noise=True, device_actor=True, split_phase=True, native_loop=native)
            assert col.split_phase
            eng.actor_sample(np.zeros((1, 8), np.float32), seed=77)            # key the library stream identically
            stats = [col.collect(n_episode=n) for n in (6, 10, 3, 1, 7, 12)]
            idx = eng.sample0()
            rows = eng.store_read(idx)
            eng.close()
            return stats, idx, rows
        finally:
            env.close()

    st_p, idx_p, rows_p = run(False)
    st_n, idx_n, rows_n = run(True)
    assert np.array_equal(idx_n, idx_p)
    for a, b in zip(st_n, st_p):
        assert a == b, (a, b)
    for k in rows_p:
        assert np.array_equal(rows_n[k], rows_p[k]), k
    assert rows_p["act"].std() > 0.1 and [s["n/ep"] for s in st_n] == [6, 10, 3, 1, 7, 12]


def test_native_collect_fails_fast_on_a_dead_worker_and_leaves_the_env_object_consistent():
    """A worker that dies mid-collect fails the native collect within seconds (the C wait looks at the workers between 0.5 s
    sleeps instead of waiting out the 60 s handshake timeout), and the env's generation counters are read back even though
    the call raised (FastCollector wraps it in try / finally)."""
    import time
    from fsrl_amd.agent import PPOLagAgent
    from fsrl_amd.data import FastCollector, HipVectorReplayBuffer
    from fsrl_amd.env import ShmemVectorEnv
    env = ShmemVectorEnv(env_num=4, workers=2, obs_dim=8, act_dim=2, episode_len=50, seed=1)
    try:
        agent = PPOLagAgent(env, None, cost_limit=10, device="cuda:0", seed=2, hidden_sizes=(64, 64), training_num=4)
        eng = agent.policy.engine
        buf = HipVectorReplayBuffer(eng, 4 * 300, 4)
        col = FastCollector(agent.policy, env, buf, exploration_noise=True, device_actor=True)
        col.collect(n_episode=4)
        env._procs[1].terminate(); env._procs[1].join(5)
        t0 = time.time()
        with pytest.raises(Exception, match="worker"):
            col.collect(n_episode=4)
        assert time.time() - t0 < 10.0
        assert env._gen[0] == int(env._desc.gen[0]) and env._gen[1] == int(env._desc.gen[1])
        eng.close()
    finally:
        env.close()


def _mk_engine(resident, hidden=256, obs_dim=8, act_dim=2, unbounded=False):
    from fsrl_amd.engine import Engine, EngineConfig
    eng = Engine(EngineConfig(obs_dim=obs_dim, act_dim=act_dim, hidden=hidden, env_num=64, max_grad_norm=0.5, target_kl=None,
                              unbounded=unbounded))
    eng.actor_set_resident(resident, idle_timeout_us=2.0e5)      # 0.2 s: the launch counts below do not depend on first-launch code loading
    rng = np.random.default_rng(3)
    eng.set_params((0.2 * rng.standard_normal(eng.n_params)).astype(np.float32))
    return eng


@pytest.mark.parametrize("hidden,obs_dim,act_dim,unbounded", [(256, 8, 2, False), (128, 27, 8, False), (64, 60, 2, True)])
def test_resident_actor_gives_the_launched_actors_actions_bit_for_bit(hidden, obs_dim, act_dim, unbounded):
    """fsrl_actor_set_resident: the collector's actor as one workgroup that stays on its CU between calls (doorbell in pinned
    memory) against one kernel launch per call -- same library RNG stream, same mean / sigma: identical actions for every row count
    (one tile, ragged second tile, the cap of four tiles, beyond it: the launch path), across a parameter upload (the resident kernel
    holds the weights in registers: the upload must end it) and across its idle timeout (it ends by itself, the next call
    launches another)."""
    import time
    a, b = _mk_engine(True, hidden, obs_dim, act_dim, unbounded), _mk_engine(False, hidden, obs_dim, act_dim, unbounded)
    rng = np.random.default_rng(5)
    ks = [1, 16, 20, 33, 64, 7, 65, 20, 20]
    for i, k in enumerate(ks):
        obs = rng.standard_normal((k, obs_dim)).astype(np.float32)
        xa, xb = a.actor_sample(obs, seed=9 if i == 0 else 0), b.actor_sample(obs, seed=9 if i == 0 else 0)
        assert np.isfinite(xa).all() and np.array_equal(xa, xb), (i, k)
        ma, mb = a.actor_forward(obs), b.actor_forward(obs)
        assert np.array_equal(ma[0], mb[0]) and np.array_equal(ma[1], mb[1]), (i, k)
    st = a.actor_resident_stats()
    assert st["requests"] == 2 * (len(ks) - 1) and st["live"], st           # k = 65 went down the launch path (and ended the resident kernel)
    assert 2 <= st["launches"] <= 12, st      # one before it, one after (+ idle timeouts while the other engine's first launches load code)
    assert b.actor_resident_stats() == dict(launches=0, requests=0, live=False)
    # a parameter upload between two calls: the next call sees the new weights
    th = (0.2 * np.random.default_rng(8).standard_normal(a.n_params)).astype(np.float32)
    a.set_params(th); b.set_params(th)
    assert not a.actor_resident_stats()["live"]
    obs = rng.standard_normal((20, obs_dim)).astype(np.float32)
    ma, mb = a.actor_forward(obs), b.actor_forward(obs)
    assert np.array_equal(ma[0], mb[0]) and a.actor_resident_stats()["launches"] == st["launches"] + 1
    # idle timeout: the kernel ends by itself, the host finds out at the next call
    a.actor_set_resident(True, idle_timeout_us=300.0)
    for rep in range(3):
        ma = a.actor_forward(obs)
        assert np.array_equal(ma[0], mb[0])
        time.sleep(0.02)
    assert a.actor_resident_stats()["launches"] == st["launches"] + 4, a.actor_resident_stats()
    a.sync()                                                                  # nothing is left running on the stream
    a.close(); b.close()


def test_resident_actor_survives_updates_between_collects():
    """collect (resident actor) -> PPO update -> collect: the update's launches are ordered behind the resident kernel's end, the
    second collect's actor runs the updated weights; identical to the one-launch-per-call actor throughout."""
    outs = []
    for resident in (True, False):
        eng = _mk_engine(resident, 64)
        rng = np.random.default_rng(1)
        acts = []
        for cyc in range(3):
            obs = rng.standard_normal((8, 8)).astype(np.float32)
            for t in range(40):
                act = eng.actor_sample(obs, seed=4 if (cyc == 0 and t == 0) else 0)
                nxt = (0.9 * obs + 0.1 * rng.standard_normal((8, 8))).astype(np.float32)
                done = np.full(8, t == 39)
                eng.push(np.arange(8), obs, act, rng.normal(0.5, 0.5, 8), (rng.random(8) < 0.1).astype(np.float64),
                         np.zeros(8, bool), done, nxt)
                acts.append(act.copy()); obs = nxt
            eng.ppo_update(np.array([0.5]), 1.0 / 1.5, 64, 2, perms=[rng.permutation(320) for _ in range(2)])
            eng.reset_store()
        outs.append((np.stack(acts), eng.get_params(), eng.actor_resident_stats()))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert 3 <= outs[0][2]["launches"] <= 12 and outs[0][2]["requests"] == 120, outs[0][2]


@pytest.mark.parametrize("kind", ["sac", "ddpg", "cvpo"])
def test_resident_actor_of_the_replay_agents_is_the_launched_one_bit_for_bit(kind):
    """The replay agents' collector actor (raw head outputs [mu | log sigma], sac_lag.py:155-183) through the resident kernel against
    one launch per call: identical (mu, sigma) and identical sampled actions for every row count, before and after updates (the
    update's launches end the resident kernel; the next call runs the updated actor)."""
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    E, sub, Do, Da, B = 24, 64, 11, 3, 64
    outs = []
    for resident in (True, False):
        eng = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=Do, act_dim=Da, hidden_sizes=(128, 128), n_critics=2, env_num=E,
                                  buffer_size=E * sub, gamma=0.99, target_kl=None))
        if kind == "cvpo":
            eng.cvpo_init(10.0, n_step=2)
        else:
            eng.sac_init(n_step=2, deterministic=(kind == "ddpg"))
        eng.actor_set_resident(resident, idle_timeout_us=2.0e5)
        rng = np.random.default_rng(4)
        got = []
        for cyc in range(2):
            for t in range(30):
                k = E if t % 3 else int(rng.integers(1, E + 1))
                obs = rng.standard_normal((k, Do)).astype(np.float32)
                mu, sg = eng.sac_actor_forward(obs)
                act = eng.actor_sample(obs, seed=6 if (cyc == 0 and t == 0) else 0)
                assert np.isfinite(mu).all() and np.isfinite(sg).all() and np.isfinite(act).all()
                got += [mu.copy(), sg.copy(), act.copy()]
                eng.push(np.arange(k), obs, np.clip(act, -1, 1), rng.standard_normal(k), (rng.random(k) < 0.2).astype(np.float64),
                         rng.random(k) < 0.05, rng.random(k) < 0.05, rng.standard_normal((k, Do)).astype(np.float32))
            for u in range(3):
                if kind == "cvpo":
                    eng.cvpo_update(B, seed=5 if (cyc == 0 and u == 0) else 0)
                else:
                    eng.sac_update(B, [0.3], 1 / 1.3, seed=5 if (cyc == 0 and u == 0) else 0)
        st = eng.actor_resident_stats()
        outs.append((got, eng.sac_get_params(0)[0].copy(), st))
        eng.close()
    assert len(outs[0][0]) == len(outs[1][0])
    for x, y in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(x, y)
    assert np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2]["requests"] == 120 and 2 <= outs[0][2]["launches"] <= 12, outs[0][2]
    assert outs[1][2] == dict(launches=0, requests=0, live=False)


def test_resident_actor_calls_spaced_around_its_idle_timeout():
    """The end / relaunch protocol under its race: calls 0.6 .. 1.4 idle timeouts apart, so that workgroups give up while a doorbell is
    being rung (the host then ends the rest, launches the next generation and rings again), 1 .. 64 rows (one to four workgroups have
    rows), a parameter upload in the middle.  Every answer equals the one-launch-per-call actor's; tools/diag/resident_soak.py is the long
    form (150 000 calls: profiles/r06_resident_actor_ab.txt)."""
    import time
    a, b = _mk_engine(True), _mk_engine(False)
    a.actor_set_resident(True, idle_timeout_us=120.0)
    rng = np.random.default_rng(2)
    pool = [rng.standard_normal((64, 8)).astype(np.float32) for _ in range(4)]
    thetas = [(0.2 * rng.standard_normal(a.n_params)).astype(np.float32) for _ in range(2)]
    want = {}
    for ti, th in enumerate(thetas):
        b.set_params(th)
        for pi, o in enumerate(pool):
            want[(ti, pi)] = b.actor_forward(o)[0].copy()
    n = 6000
    for i in range(n):
        ti = int(i >= n // 2)
        if i in (0, n // 2):
            a.set_params(thetas[ti])
        gap = rng.uniform(0.6, 1.4) * 120e-6 if i % 3 else 0.0
        t = time.perf_counter()
        while time.perf_counter() - t < gap:
            pass
        pi, k = int(rng.integers(0, 4)), int(rng.integers(1, 65))
        assert np.array_equal(a.actor_forward(pool[pi][:k])[0], want[(ti, pi)][:k]), (i, k)
    st = a.actor_resident_stats()
    assert st["requests"] == n and 100 < st["launches"] < n, st          # the timeouts did fire, and not before every call
    a.close(); b.close()
