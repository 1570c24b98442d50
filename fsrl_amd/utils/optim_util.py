"""PID Lagrange-multiplier controller, host float64 (the multipliers enter the HIP kernels as
launch scalars).  Interface and arithmetic of fsrl/utils/optim_util.py:18-62; the reference's
`np.mean` / `np.maximum` on scalars are plain float64 operations."""
from typing import Sequence


class LagrangianOptimizer:
    def __init__(self, pid: Sequence[float] = (0.05, 0.0005, 0.1)) -> None:
        assert len(pid) == 3, " the pid param should be a list with 3 numbers"
        self.pid = tuple(pid)
        self.error_old = 0.0
        self.error_integral = 0.0
        self.lagrangian = 0.0

    def step(self, value: float, threshold: float) -> None:
        import numpy as np
        error_new = float(np.mean(np.asarray(value, np.float64) - threshold))
        error_diff = max(0.0, error_new - self.error_old)
        self.error_integral = max(0.0, self.error_integral + error_new)
        self.error_old = error_new
        kp, ki, kd = self.pid
        self.lagrangian = max(0.0, kp * error_new + ki * self.error_integral + kd * error_diff)

    def get_lag(self) -> float:
        return self.lagrangian

    def state_dict(self) -> dict:
        return {"pid": self.pid, "error_old": self.error_old,
                "error_integral": self.error_integral, "lagrangian": self.lagrangian}

    def load_state_dict(self, params: dict) -> None:
        self.pid = params["pid"]
        self.error_old = params["error_old"]
        self.error_integral = params["error_integral"]
        self.lagrangian = params["lagrangian"]
