"""The build-owned shared-memory vector env (fsrl_amd/env/shmem.py): worker processes + one shared block, the host side
of the headline metric (north_star: FastCollector over ShmemVectorEnv; examples/mlp/train_ppol_agent.py:120-123)."""
import numpy as np
import pytest


def _rollout(env, steps, seed):
    rng = np.random.default_rng(seed)
    n = len(env)
    obs, _ = env.reset()
    out = [obs.copy()]
    for t in range(steps):
        ids = np.arange(n) if t % 3 else np.arange(0, n, 2)          # full and partial id sets
        act = rng.uniform(-1, 1, (len(ids), env.act_dim)).astype(np.float32)
        o, r, term, trunc, info = env.step(act, ids)
        out += [o.copy(), r.copy(), trunc.copy(), info["cost"].copy()]
        done = np.flatnonzero(term | trunc)
        if done.size:
            ro, _ = env.reset(ids[done])
            out.append(ro.copy())
    return out


@pytest.mark.parametrize("spin_us", [0.0, 50.0])
def test_one_worker_is_bit_identical_to_the_in_process_env(spin_us):
    """the futex handshake with and without a spin phase before the sleep"""
    from fsrl_amd.env import ShmemVectorEnv, SyntheticSafetyVectorEnv
    a = SyntheticSafetyVectorEnv(env_num=6, episode_len=11, seed=3)
    b = ShmemVectorEnv(env_num=6, workers=1, episode_len=11, seed=3, spin_us=spin_us)
    try:
        ra, rb = _rollout(a, 40, 0), _rollout(b, 40, 0)
        assert len(ra) == len(rb)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)
    finally:
        b.close()


@pytest.mark.parametrize("workers", [3, 8])
def test_workers_step_their_slices_and_time_limits_hold(workers):
    from fsrl_amd.env import ShmemVectorEnv
    env = ShmemVectorEnv(env_num=8, workers=workers, episode_len=7, seed=1)
    try:
        obs, _ = env.reset()
        assert obs.shape == (8, 8) and np.abs(obs).max() > 0
        lens = np.zeros(8, int)
        for t in range(30):
            ids = np.arange(8) if t % 2 else np.array([1, 2, 5, 7])
            o, r, term, trunc, info = env.step(np.zeros((len(ids), 2), np.float32), ids)
            lens[ids] += 1
            assert o.shape == (len(ids), 8) and r.shape == (len(ids), ) and set(np.unique(info["cost"])) <= {0.0, 1.0}
            assert np.array_equal(trunc, lens[ids] == 7)             # every env keeps ITS OWN step count
            d = ids[trunc]
            if d.size:
                env.reset(d)
                lens[d] = 0
        # same seed, same worker count -> same trajectories (each worker has its own stream)
        env2 = ShmemVectorEnv(env_num=8, workers=workers, episode_len=7, seed=1)
        try:
            o1, _ = env.reset(); o2, _ = env2.reset()
            assert not np.array_equal(o1, o2)                        # env advanced its streams, env2 is fresh
        finally:
            env2.close()
    finally:
        env.close()


def test_collector_runs_over_the_shmem_env():
    """FastCollector (host side only: no buffer, random actions) collects exactly n episodes from worker processes."""
    from fsrl_amd.data import FastCollector
    from fsrl_amd.env import ShmemVectorEnv

    class _Pol:                                # the collector needs map_action(_inverse) only on the random path
        action_space = None
        def map_action(self, a): return a
        def map_action_inverse(self, a): return a

    env = ShmemVectorEnv(env_num=6, workers=3, episode_len=9, seed=2, spin_us=20.0)      # spin, then sleep on the futex
    try:
        col = FastCollector(_Pol(), env, None)
        st = col.collect(n_episode=10, random=True)
        assert st["n/ep"] == 10 and st["len"] == 9.0 and st["truncated"] == 1.0
        import time
        time.sleep(0.2)                                   # the workers sleep on their lane's generation word ...
        st = col.collect(n_episode=6, random=True)        # ... and are woken by the next command
        assert st["n/ep"] == 6 and st["len"] == 9.0
    finally:
        env.close()


def test_two_lanes_step_independently_and_match_the_synchronous_env():
    """step_async / step_wait: the two halves of the workers take commands on their own generation / completion words.
    Interleaving the lanes (lane 0 steps while lane 1 is being collected) gives every env the trajectory the synchronous
    step() gives it: a worker's random stream is its own."""
    from fsrl_amd.env import ShmemVectorEnv
    a = ShmemVectorEnv(env_num=8, workers=4, episode_len=50, seed=5)
    b = ShmemVectorEnv(env_num=8, workers=4, episode_len=50, seed=5)
    try:
        assert a.n_lanes == 2 and [list(l) for l in a.lanes] == [[0, 1, 2, 3], [4, 5, 6, 7]]
        oa, _ = a.reset(); ob, _ = b.reset()
        assert np.array_equal(oa, ob)
        rng = np.random.default_rng(0)
        l0, l1 = a.lanes
        acts = rng.uniform(-1, 1, (6, 8, 2)).astype(np.float32)
        b.step_async(acts[0][l0], l0)
        b.step_async(acts[0][l1], l1)
        for t in range(5):
            ref = a.step(acts[t])
            g0 = b.step_wait(l0)
            b.step_async(acts[t + 1][l0], l0)              # lane 0 is stepping again while lane 1 is read
            g1 = b.step_wait(l1)
            b.step_async(acts[t + 1][l1], l1)
            for k in range(4):
                x, y0, y1 = ref[k], g0[k], g1[k]
                if isinstance(x, dict):
                    x, y0, y1 = x["cost"], y0["cost"], y1["cost"]
                assert np.array_equal(x[l0], y0) and np.array_equal(x[l1], y1)
        b.step_wait(l0); b.step_wait(l1)
        with np.testing.assert_raises(AssertionError):
            b.step_async(acts[0][:5], np.arange(5))        # ids of one call belong to one lane
    finally:
        a.close(); b.close()


def test_a_dead_worker_is_reported_not_waited_for_forever():
    from fsrl_amd.env import ShmemVectorEnv
    env = ShmemVectorEnv(env_num=4, workers=2, episode_len=9, seed=2)
    try:
        env.reset()
        env._procs[1].terminate(); env._procs[1].join(5)
        with pytest.raises(RuntimeError):
            env.step(np.zeros((4, 2), np.float32))
    finally:
        env.close()


# ---- env factories: the reference's `ShmemVectorEnv([lambda: gym.make(task) ...])` / `DummyVectorEnv` (train_ppol_agent.py:120-123)
def _fns(n, horizon=9):
    from fsrl_amd.env import PointCircleEnv
    return [lambda: PointCircleEnv(max_episode_steps=horizon) for _ in range(n)]


def _rollout_fns(env, steps, n, seed):
    rng = np.random.default_rng(seed)
    obs, _ = env.reset()
    out = [obs.copy()]
    for t in range(steps):
        ids = np.arange(n) if t % 3 else np.arange(0, n, 2)
        act = rng.uniform(-1, 1, (len(ids), 2)).astype(np.float32)
        o, r, term, trunc, info = env.step(act, ids)
        out += [o.copy(), r.copy(), term.copy(), trunc.copy(), info["cost"].copy()]
        done = np.flatnonzero(term | trunc)
        if done.size:
            ro, _ = env.reset(ids[done])
            out.append(ro.copy())
    return out


def test_dummy_vector_env_steps_each_instance():
    from fsrl_amd.env import DummyVectorEnv, PointCircleEnv
    env = DummyVectorEnv(_fns(3), seed=5)
    singles = [PointCircleEnv(max_episode_steps=9) for _ in range(3)]
    o, _ = env.reset()
    assert o.shape == (3, 6) and o.dtype == np.float32 and len(env) == 3
    for i, e in enumerate(singles):
        assert np.array_equal(o[i], e.reset(seed=5 + i)[0])           # tianshou's venv.seed(s): env i gets s + i
    act = np.array([[0.5, -1.0], [1.0, 1.0]], np.float32)
    o, r, term, trunc, info = env.step(act, np.array([0, 2]))
    for j, i in enumerate((0, 2)):
        so, sr, st, stc, si = singles[i].step(act[j])
        assert np.array_equal(o[j], so) and r[j] == sr and term[j] == st and trunc[j] == stc and info["cost"][j] == si["cost"]
    assert r.dtype == np.float64 and term.dtype == bool


def test_old_gym_api_and_missing_cost():
    """4-tuple step / bare-obs reset envs (old gym) and envs without a cost signal go through the same adapter"""
    from fsrl_amd.env import Box, DummyVectorEnv

    class Old:
        observation_space, action_space = Box(-1, 1, (3, )), Box(-1, 1, (1, ))
        def __init__(self): self.t = 0
        def seed(self, s): self.s = s
        def reset(self): self.t = 0; return np.full(3, getattr(self, "s", -1), np.float32)
        def step(self, a):
            self.t += 1
            return np.full(3, self.t, np.float32), 1.0, self.t >= 2, {"TimeLimit.truncated": self.t >= 2}

    env = DummyVectorEnv([Old, Old], seed=7)
    o, _ = env.reset()
    assert np.array_equal(o[:, 0], [7, 8])
    env.step(np.zeros((2, 1)))
    o, r, term, trunc, info = env.step(np.zeros((2, 1)))
    assert trunc.all() and not term.any() and np.array_equal(info["cost"], [0.0, 0.0]) and np.array_equal(o[:, 0], [2, 2])


@pytest.mark.parametrize("workers", [1, 2, 5])
def test_shmem_env_over_factories_matches_the_in_process_env(workers):
    """worker processes stepping the caller's own envs (factories sent with cloudpickle) == the same instances stepped in
    process, whatever the worker count: every env has its own state and seed"""
    from fsrl_amd.env import DummyVectorEnv, ShmemVectorEnv
    a = DummyVectorEnv(_fns(5), seed=11)
    b = ShmemVectorEnv(_fns(5), workers=workers, seed=11)
    try:
        assert len(b) == 5 and b.observation_space.shape == (6, ) and b.action_space.shape == (2, )
        assert b.spec.max_episode_steps == 9
        ra, rb = _rollout_fns(a, 40, 5, 0), _rollout_fns(b, 40, 5, 0)
        assert len(ra) == len(rb)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)
    finally:
        b.close()


def test_collector_over_factory_envs():
    """FastCollector over DummyVectorEnv / ShmemVectorEnv(env_fns): n episodes, the cost from info["cost"]"""
    from fsrl_amd.data import FastCollector
    from fsrl_amd.env import DummyVectorEnv, ShmemVectorEnv

    class _Pol:
        action_space = None
        def map_action(self, a): return a
        def map_action_inverse(self, a): return a

    stats = []
    for make in (lambda: DummyVectorEnv(_fns(4), seed=3), lambda: ShmemVectorEnv(env_fns=_fns(4), workers=2, seed=3)):
        env = make()
        try:
            np.random.seed(0)
            env.action_space.seed(0)
            col = FastCollector(_Pol(), env, None)
            st = col.collect(n_episode=8, random=True)
            stats.append(st)
            assert st["n/ep"] == 8 and st["n/st"] == 8 * 9 and st["len"] == 9
        finally:
            env.close()
    assert stats[0]["rew"] == stats[1]["rew"] and stats[0]["cost"] == stats[1]["cost"]


def test_collector_wraps_a_single_env():
    """fast_collector.py:55-58: a bare env is wrapped into a one-env vector env, with the reference's warning"""
    from fsrl_amd.data import FastCollector
    from fsrl_amd.env import PointCircleEnv

    class _Pol:
        action_space = None
        def map_action(self, a): return a
        def map_action_inverse(self, a): return a

    with pytest.warns(UserWarning, match="Single environment"):
        col = FastCollector(_Pol(), PointCircleEnv(max_episode_steps=7), None)
    st = col.collect(n_episode=3, random=True)
    assert col.env_num == 1 and st["n/ep"] == 3 and st["n/st"] == 21


def test_step_with_permuted_ids_returns_rows_in_the_requested_order():
    """step(act, ids) takes ids in any order; a permuted contiguous range like [0, 2, 1, 3] must not take the slice shortcut."""
    from fsrl_amd.env import ShmemVectorEnv
    a = ShmemVectorEnv(env_num=4, workers=2, episode_len=50, seed=5)
    b = ShmemVectorEnv(env_num=4, workers=2, episode_len=50, seed=5)
    try:
        a.reset(); b.reset()
        rng = np.random.default_rng(0)
        for _ in range(5):
            act = rng.standard_normal((4, 2)).astype(np.float32)
            perm = np.array([0, 2, 1, 3])
            oa, ra, ta, ua, ia = a.step(act, np.arange(4))
            ob, rb, tb, ub, ib = b.step(act[perm], perm)
            assert np.array_equal(ob, oa[perm]) and np.array_equal(rb, ra[perm]) and np.array_equal(ib["cost"], ia["cost"][perm])
    finally:
        a.close(); b.close()


class _Raises:
    """an env whose third step raises (picklable: module level)"""
    class _Space:
        shape = (3, )
        low = -np.ones(3, np.float32); high = np.ones(3, np.float32)
    observation_space = action_space = _Space()

    def __init__(self):
        self.t = 0

    def reset(self, seed=None):
        self.t = 0
        return np.zeros(3, np.float32), {}

    def step(self, a):
        self.t += 1
        if self.t == 3:
            raise ValueError("boom inside the user's env")
        return np.zeros(3, np.float32), 0.0, False, False, {}


def test_an_exception_inside_a_workers_env_reaches_the_collector_quickly():
    import time
    from fsrl_amd.env import ShmemVectorEnv
    env = ShmemVectorEnv([_Raises, _Raises], workers=2, seed=None)
    try:
        env.reset()
        act = np.zeros((2, 3), np.float32)
        env.step(act); env.step(act)
        t0 = time.time()
        with pytest.raises(RuntimeError, match="raised inside"):
            env.step(act)
        assert time.time() - t0 < 5.0
    finally:
        env.close()


def test_reset_kwargs_are_refused_not_dropped_and_typeerrors_of_a_users_reset_are_not_masked():
    from fsrl_amd.env import DummyVectorEnv, ShmemVectorEnv
    env = ShmemVectorEnv(env_num=2, workers=1, episode_len=5, seed=1)
    try:
        with pytest.raises(TypeError, match="reset kwargs"):
            env.reset(options={"x": 1})
    finally:
        env.close()

    class Bad(_Raises):
        def reset(self, seed=None):
            raise TypeError("a bug in the user's reset")

    with pytest.raises(TypeError, match="user's reset"):
        DummyVectorEnv([Bad], seed=3).reset()


def test_worker_processes_can_be_capped_at_the_usable_cpus():
    from fsrl_amd.env import ShmemVectorEnv
    from fsrl_amd.parallel import usable_cpus
    cap = max(1, int(usable_cpus()))
    env = ShmemVectorEnv(env_num=2 * cap + 3, workers=2 * cap + 3, episode_len=5, seed=1, cap_workers=True)
    try:
        assert env.workers == cap and env.workers_requested == 2 * cap + 3 and len(env._procs) == cap
        obs, _ = env.reset()
        assert obs.shape == (2 * cap + 3, 8)
        o, r, t, u, i = env.step(np.zeros((2 * cap + 3, 2), np.float32))
        assert o.shape == (2 * cap + 3, 8) and len(i["cost"]) == 2 * cap + 3
    finally:
        env.close()
    env = ShmemVectorEnv(env_num=cap + 2, workers=cap + 2, episode_len=5, seed=1)
    try:
        assert env.workers == cap + 2
    finally:
        env.close()


class _SlowEnv:
    """a duck-typed env whose step burns ~200 us (a real simulator's order of magnitude)"""

    def __init__(self):
        from types import SimpleNamespace
        self.observation_space = SimpleNamespace(shape=(3, ))
        self.action_space = SimpleNamespace(shape=(2, ), low=-np.ones(2, np.float32), high=np.ones(2, np.float32))
        self.t = 0

    def reset(self, seed=None):
        self.t = 0
        return np.zeros(3, np.float32), {}

    def step(self, a):
        import time
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 200e-6:
            pass
        self.t += 1
        return np.full(3, self.t, np.float32), 1.0, False, self.t >= 4, {"cost": 0.0}


def test_auto_worker_count_follows_the_env_step_cost():
    """cap_workers="auto" (r5): an env that costs (almost) nothing per step runs on 4 processes whatever was requested -- a vector
    step of such an env is the wake-up of its workers -- and an env with a real step cost keeps one process per env (the
    reference's layout).  The cost is known for the synthetic dynamics (busy_us) and measured on the probe instance for factories."""
    from fsrl_amd.env import ShmemVectorEnv
    env = ShmemVectorEnv(env_num=12, workers=12, episode_len=5, seed=1, busy_us=0.0, cap_workers="auto")
    try:
        assert env.workers == 4 and env.workers_requested == 12 and "4 processes instead of 12" in env.worker_mode
        obs, _ = env.reset()
        o, r, t, u, i = env.step(np.zeros((12, 2), np.float32))
        assert o.shape == (12, 8)
    finally:
        env.close()
    env = ShmemVectorEnv(env_num=6, workers=6, episode_len=5, seed=1, busy_us=100.0, cap_workers="auto")
    try:
        assert env.workers == 6 and "as requested" in env.worker_mode
    finally:
        env.close()
    env = ShmemVectorEnv(env_fns=[_SlowEnv for _ in range(5)], cap_workers="auto")
    try:
        assert env.workers == 5 and "as requested" in env.worker_mode, env.worker_mode
        obs, _ = env.reset()
        o, r, t, u, i = env.step(np.zeros((5, 2), np.float32))
        assert np.array_equal(o, np.ones((5, 3), np.float32))
    finally:
        env.close()
