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
