"""A single gym-style safety env (gymnasium API, no gymnasium import): a point mass rewarded for circling the origin and
charged a cost outside the band |x| <= x_lim -- the task family of SafetyPointCircle / SafetyCarCircle in a dozen lines.
It exists so that the env-factory path (`DummyVectorEnv([lambda: PointCircleEnv() ...])`, `ShmemVectorEnv(env_fns=...)`)
has a real per-instance env to step in tests and examples; it is not one of the reference's tasks."""
from types import SimpleNamespace

import numpy as np

from fsrl_amd.env.synthetic import Box


class PointCircleEnv:
    def __init__(self, max_episode_steps=100, radius=1.5, x_lim=1.0, dt=0.1):
        self.observation_space = Box(-np.inf, np.inf, (6, ))
        self.action_space = Box(-1.0, 1.0, (2, ))
        self.spec = SimpleNamespace(id="PointCircle-v0", max_episode_steps=max_episode_steps)
        self.radius, self.x_lim, self.dt = radius, x_lim, dt
        self.rng = np.random.default_rng(0)
        self.pos, self.vel, self.t = np.zeros(2), np.zeros(2), 0

    def _obs(self):
        r = np.linalg.norm(self.pos)
        return np.array([*self.pos, *self.vel, r - self.radius, self.x_lim - abs(self.pos[0])], np.float32)

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.rng = np.random.default_rng(seed)
        self.pos = self.rng.uniform(-0.5, 0.5, 2)
        self.vel = np.zeros(2)
        self.t = 0
        return self._obs(), {}

    def step(self, action):
        a = np.clip(np.asarray(action, np.float64).reshape(2), -1.0, 1.0)
        self.vel = 0.9 * self.vel + self.dt * a
        self.pos = self.pos + self.dt * self.vel
        self.t += 1
        r = np.linalg.norm(self.pos) + 1e-8
        # tangential speed (counter-clockwise), discounted by the distance from the target circle
        rew = float((-self.pos[1] * self.vel[0] + self.pos[0] * self.vel[1]) / r / (1.0 + abs(r - self.radius)))
        cost = float(abs(self.pos[0]) > self.x_lim)
        terminated = bool(r > 4.0 * self.radius)
        truncated = self.t >= self.spec.max_episode_steps
        return self._obs(), rew, terminated, truncated, {"cost": cost}

    def close(self):
        pass
