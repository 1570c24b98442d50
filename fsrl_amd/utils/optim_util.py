"""PID controller of a Lagrange multiplier, host float64: the multiplier reaches the HIP kernels as a launch scalar.

Public surface and checkpoint keys of fsrl/utils/optim_util.py:18-62 (`step(value, threshold)`, `get_lag()`,
`state_dict()` with pid / error_old / error_integral / lagrangian); the arithmetic is the reference's, spelled as plain
float operations: with e_t = mean(value - threshold),
    I_t = max(0, I_{t-1} + e_t),   D_t = max(0, e_t - e_{t-1}),   lambda_t = max(0, Kp e_t + Ki I_t + Kd D_t).
Pinned bit for bit by tests/golden/pid_trace.npz (tests/test_oracle_scans.py)."""
from typing import Dict, Sequence

import numpy as np

_STATE_KEYS = ("pid", "error_old", "error_integral", "lagrangian")


class LagrangianOptimizer:
    def __init__(self, pid: Sequence[float] = (0.05, 0.0005, 0.1)) -> None:
        assert len(pid) == 3, " the pid param should be a list with 3 numbers"
        self.pid = tuple(pid)
        self.error_old = self.error_integral = self.lagrangian = 0.0

    def step(self, value, threshold: float) -> None:
        """One controller update from the measured constraint value(s) of the last collect."""
        gain_p, gain_i, gain_d = self.pid
        err = float(np.mean(np.asarray(value, np.float64) - threshold))
        rise = err - self.error_old
        self.error_integral = max(0.0, self.error_integral + err)
        self.lagrangian = max(0.0, gain_p * err + gain_i * self.error_integral + gain_d * max(0.0, rise))
        self.error_old = err

    def get_lag(self) -> float:
        return self.lagrangian

    def state_dict(self) -> Dict:
        return {k: getattr(self, k) for k in _STATE_KEYS}

    def load_state_dict(self, params: Dict) -> None:
        for k in _STATE_KEYS:
            setattr(self, k, params[k])
