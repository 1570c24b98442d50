"""Vector-env adapters over user env factories: what the reference gets from tianshou's `DummyVectorEnv([lambda: gym.make(task)
...])` / `ShmemVectorEnv(...)` (examples/mlp/train_ppol_agent.py:120-123; fsrl/data/fast_collector.py:55-58 wraps a single env
the same way).  gymnasium is not in this image, so the envs are duck-typed:

    reset(**kwargs) -> (obs, info)      | obs                                   (gymnasium | old gym)
    step(action)    -> (obs, rew, terminated, truncated, info) | (obs, rew, done, info)
    observation_space / action_space with .shape (and .low / .high for the action bound), optional .spec, .close()

and the safety cost of a step is `info["cost"]` (Bullet-Safety-Gym / Safety-Gymnasium; fast_collector.py:262 reads it), 0 when absent.

`EnvList` gives a list of such envs the calling convention the collector uses on every vector env of this package -- `len(env)`,
`reset(ids=None) -> (obs, info)`, `step(act, ids) -> (obs, rew, terminated, truncated, {"cost": cost})` -- and is what both
`DummyVectorEnv` (in-process) and the worker processes of `ShmemVectorEnv(env_fns=...)` step."""
import numpy as np


def _accepts_seed(env):
    """does env.reset take a `seed` keyword (gymnasium / gym >= 0.22)?  Decided from the signature, not by catching the
    TypeError of a trial call -- a TypeError raised INSIDE a user's reset must reach the user."""
    import inspect
    # a wrapper's `reset(self, **kwargs)` says nothing (old-gym Wrapper / TimeLimit forward to an inner reset() that may take no
    # seed): walk .env / .unwrapped down to the first explicit signature
    seen = 0
    while env is not None and seen < 32:
        try:
            params = inspect.signature(env.reset).parameters
        except (TypeError, ValueError):
            return True
        if "seed" in params:
            return True
        if not any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values()):
            return False
        inner = getattr(env, "env", None)
        if inner is None or inner is env:
            un = getattr(env, "unwrapped", None)
            inner = un if (un is not None and un is not env) else None
        if inner is None:
            return True                                     # a bare reset(**kwargs): nothing below it to ask
        env, seen = inner, seen + 1
    return True


def _reset_one(env, seed, kwargs):
    if seed is not None:
        if _accepts_seed(env):
            out = env.reset(seed=int(seed), **kwargs)
        else:                                               # old gym: env.seed(s); env.reset()
            if hasattr(env, "seed"):
                env.seed(int(seed))
            out = env.reset(**kwargs)
    else:
        out = env.reset(**kwargs)
    if isinstance(out, tuple) and len(out) == 2 and isinstance(out[1], dict):
        return out[0]
    return out


def _step_one(env, act):
    out = env.step(act)
    if len(out) == 5:
        obs, rew, term, trunc, info = out
    else:                                                   # old gym: done + the TimeLimit wrapper's flag
        obs, rew, done, info = out
        trunc = bool(info.get("TimeLimit.truncated", False))
        term = bool(done) and not trunc
    cost = info.get("cost", 0.0) if isinstance(info, dict) else 0.0
    return obs, rew, term, trunc, cost


class EnvList:
    def __init__(self, envs, seed=None, first_index=0):
        self.envs = list(envs)
        self.env_num = len(self.envs)
        e = self.envs[0]
        self.observation_space, self.action_space = e.observation_space, e.action_space
        self.spec = getattr(e, "spec", None)
        self.obs_dim = int(np.prod(self.observation_space.shape))
        self.act_dim = int(np.prod(self.action_space.shape))
        # tianshou's venv.seed(s): env i gets s + i, consumed by its next reset
        self._pending_seed = [None] * self.env_num
        if seed is not None:
            self.seed(seed, first_index)

    def __len__(self):
        return self.env_num

    def seed(self, seed, first_index=0):
        self._pending_seed = [None if seed is None else int(seed) + first_index + i for i in range(self.env_num)]

    def reset(self, ids=None, **kwargs):
        ids = np.arange(self.env_num) if ids is None else np.asarray(ids)
        obs = np.empty((len(ids), self.obs_dim), np.float32)
        for j, i in enumerate(ids):
            s, self._pending_seed[i] = self._pending_seed[i], None
            obs[j] = np.asarray(_reset_one(self.envs[i], s, kwargs), np.float32).reshape(-1)
        return obs, {}

    def step(self, act, ids=None):
        ids = np.arange(self.env_num) if ids is None else np.asarray(ids)
        n = len(ids)
        act = np.asarray(act).reshape(n, *self.action_space.shape)
        obs = np.empty((n, self.obs_dim), np.float32)
        rew, cost = np.empty(n, np.float64), np.empty(n, np.float64)
        term, trunc = np.empty(n, bool), np.empty(n, bool)
        for j, i in enumerate(ids):
            o, rew[j], term[j], trunc[j], cost[j] = _step_one(self.envs[i], act[j])
            obs[j] = np.asarray(o, np.float32).reshape(-1)
        return obs, rew, term, trunc, {"cost": cost}

    def close(self):
        for e in self.envs:
            if hasattr(e, "close"):
                e.close()


class DummyVectorEnv(EnvList):
    """In-process vector env over env factories (tianshou's DummyVectorEnv): `DummyVectorEnv([lambda: make(task) for _ in range(n)])`."""

    def __init__(self, env_fns, seed=None):
        super().__init__([fn() for fn in env_fns], seed=seed)


def as_vector_env(env):
    """A single env becomes a one-env vector env (fsrl/data/fast_collector.py:55-58, fsrl/agent/base_agent.py:160-163, with the
    reference's warning); vector envs pass through."""
    if hasattr(env, "__len__"):
        return env
    import warnings
    warnings.warn("Single environment detected, wrap to DummyVectorEnv.")
    return EnvList([env])
