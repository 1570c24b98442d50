"""Build-owned SYNTHETIC vector environment with SafetyCarCircle-v0's interface (obs 8, act 2,
300-step time limit, binary-ish cost).  The real Bullet-Safety-Gym / gymnasium packages are
not available in the build or GPU images; this env exists so that env-steps/s and the
collector -> store -> update loop can be exercised end to end.  It is NOT a physics model:
`obs' = A obs + B act + noise`, reward = progress along a circle, cost = 1 outside a band.
`busy_us`: host time one env step takes in total, to mimic a simulator's cost (0 = free)."""
import time

import numpy as np


class Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low = np.full(shape, low, dtype)
        self.high = np.full(shape, high, dtype)
        self.shape = tuple(shape)
        self.dtype = dtype
        self._rng = np.random.default_rng(0)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def seed(self, seed):
        self._rng = np.random.default_rng(seed)


def stable_coupling(obs_dim):
    """the `coupling` that keeps the synthetic dynamics as contractive at obs_dim columns as the default is at 8"""
    return 0.02 * float(np.sqrt(8.0 / max(int(obs_dim), 8)))


class SyntheticSafetyVectorEnv:
    """Vector env with the tianshou BaseVectorEnv calling convention the collector uses:
    `len(env)`, `reset(ids=None) -> (obs, info)`, `step(act, ids) -> (obs, rew, term, trunc, info)`."""

    def __init__(self, env_num=20, obs_dim=8, act_dim=2, episode_len=300, seed=0, busy_us=0.0, coupling=0.02, cost_threshold=1.0):
        # coupling: scale of the random part of A = 0.95 I + coupling N(0, 1).  The default keeps every existing fixture; WIDE
        # observations need coupling = 0.02 sqrt(8 / obs_dim) (`stable_coupling`): the random part's spectral radius grows like
        # coupling sqrt(obs_dim), and at obs 33 / 60 the default dynamics have eigenvalues of modulus 1.05 / 1.08 -- a 1000-step
        # episode overflows
        # cost_threshold: cost = 1 where |state[1]| exceeds it.  The default (1.0) makes a 1000-step episode of an untrained policy cost
        # several hundred -- against the reference's PID gains (tuned for limits of 10 - 25) the multiplier then swings between 0 and
        # 100 from one collect to the next; the multi-seed fixtures use a higher threshold so that episode costs sit near the limit
        self.cost_threshold = float(cost_threshold)
        self.env_num, self.obs_dim, self.act_dim = env_num, obs_dim, act_dim
        self.episode_len, self.busy_us = episode_len, busy_us
        self.observation_space = Box(-np.inf, np.inf, (obs_dim, ))
        self.action_space = Box(-1.0, 1.0, (act_dim, ))
        from types import SimpleNamespace
        self.spec = SimpleNamespace(id="SyntheticSafety-v0", max_episode_steps=episode_len)   # gym's env.spec (CVPO reads it)
        self.rng = np.random.default_rng(seed)
        k = np.random.default_rng(1234)
        self.A = (0.95 * np.eye(obs_dim) + coupling * k.standard_normal((obs_dim, obs_dim))).astype(np.float32)
        self.B = (0.3 * k.standard_normal((act_dim, obs_dim))).astype(np.float32)
        self.state = np.zeros((env_num, obs_dim), np.float32)
        self.t = np.zeros(env_num, int)

    def __len__(self):
        return self.env_num

    def seed(self, seed):
        self.rng = np.random.default_rng(seed)

    def reset(self, ids=None, **kwargs):
        ids = np.arange(self.env_num) if ids is None else np.asarray(ids)
        self.state[ids] = self.rng.standard_normal((len(ids), self.obs_dim)).astype(np.float32)
        self.t[ids] = 0
        return self.state[ids].copy(), {}

    def step(self, act, ids=None):
        # an env step costs busy_us of host time IN TOTAL (the few microseconds of numpy below count towards it, like a
        # simulator's own arithmetic would): the clock is read first, the remainder is burnt at the end
        t_in = time.perf_counter() if self.busy_us > 0 else 0.0
        ids = np.arange(self.env_num) if ids is None else np.asarray(ids)
        act = np.asarray(act, np.float32).reshape(len(ids), self.act_dim)
        end = t_in + self.busy_us * 1e-6 * len(ids) if self.busy_us > 0 else 0.0
        s = self.state[ids] @ self.A + act @ self.B
        s += 0.05 * self.rng.standard_normal(s.shape).astype(np.float32)
        self.state[ids] = s
        self.t[ids] += 1
        rew = (s[:, 0] * act[:, 0] - 0.1 * (act**2).sum(1)).astype(np.float64) + 0.5
        cost = (np.abs(s[:, 1]) > self.cost_threshold).astype(np.float64)
        truncated = self.t[ids] >= self.episode_len
        terminated = np.zeros(len(ids), bool)
        while end and time.perf_counter() < end:
            pass
        return s.copy(), rew, terminated, truncated, {"cost": cost}

    def close(self):
        pass
