"""PID Lagrange-multiplier controller of the oracle (float64).

Reference: fsrl/utils/optim_util.py:28-41 (LagrangianOptimizer.step) and the way
fsrl/policy/lagrangian_base.py:98-120,145-166 feeds/consumes it.  TEST INFRASTRUCTURE ONLY.
"""
import numpy as np


class PIDLagrangian:
    def __init__(self, pid=(0.05, 0.0005, 0.1)):
        assert len(pid) == 3
        self.kp, self.ki, self.kd = (float(x) for x in pid)
        self.error_old = 0.0
        self.error_integral = 0.0
        self.lagrangian = 0.0

    def step(self, value, threshold):
        e = float(np.mean(np.asarray(value, np.float64) - threshold))
        d = max(0.0, e - self.error_old)
        self.error_integral = max(0.0, self.error_integral + e)
        self.error_old = e
        self.lagrangian = max(0.0, self.kp * e + self.ki * self.error_integral + self.kd * d)
        return self.lagrangian


def rescaling_factor(lagrangians, rescaling=True):
    """1/(sum(lambda)+1) -- lagrangian_base.py:156."""
    return 1.0 / (float(np.sum(lagrangians)) + 1.0) if rescaling else 1.0
