"""LagrangianPolicy: PID multipliers stepped from the episodic training cost once per collect
(fsrl/policy/lagrangian_base.py:16-166).  The multipliers stay host float64 and are handed to
the HIP kernels as launch scalars together with rescaling = 1/(sum(lambda)+1)."""
from typing import Dict, List, Tuple, Union

import numpy as np

from fsrl_amd.policy.base_policy import BasePolicy
from fsrl_amd.utils.optim_util import LagrangianOptimizer


class LagrangianPolicy(BasePolicy):
    def __init__(self, actor, critics, dist_fn=None, logger=None, use_lagrangian: bool = True,
                 lagrangian_pid: Tuple = (0.05, 0.0005, 0.1), cost_limit: Union[List, float] = np.inf,
                 rescaling: bool = True, gamma: float = 0.99, max_batchsize: int = 99999,
                 reward_normalization: bool = False, deterministic_eval: bool = True,
                 action_scaling: bool = True, action_bound_method: str = "clip",
                 observation_space=None, action_space=None, lr_scheduler=None) -> None:
        super().__init__(actor, critics, dist_fn, logger, gamma, max_batchsize, reward_normalization,
                         deterministic_eval, action_scaling, action_bound_method, observation_space,
                         action_space, lr_scheduler)
        self.rescaling = rescaling
        self.use_lagrangian = use_lagrangian
        self.cost_limit = [cost_limit] * (self.critics_num - 1) if np.isscalar(cost_limit) else cost_limit
        if self.use_lagrangian:
            assert len(self.cost_limit) == (self.critics_num - 1), \
                "cost_limit must has equal len of critics_num"
            self.lag_optims = [LagrangianOptimizer(lagrangian_pid) for _ in range(self.critics_num - 1)]
        else:
            self.lag_optims = []

    def pre_update_fn(self, stats_train: Dict, **kwarg) -> None:
        self.update_lagrangian(stats_train["cost"])

    def update_cost_limit(self, cost_limit: float) -> None:
        self.cost_limit = [cost_limit] * (self.critics_num - 1) if np.isscalar(cost_limit) else cost_limit

    def update_lagrangian(self, cost_values: Union[List, float]) -> None:
        if np.isscalar(cost_values):
            cost_values = [cost_values]
        for i, lag_optim in enumerate(self.lag_optims):
            lag_optim.step(cost_values[i], self.cost_limit[i])

    def get_extra_state(self):
        return [optim.state_dict() for optim in self.lag_optims] if len(self.lag_optims) else None

    def set_extra_state(self, state):
        # torch hands the saved extra state itself; the reference also tolerates a full dict
        if isinstance(state, dict) and "_extra_state" in state:
            state = state["_extra_state"]
        if state and self.lag_optims:
            for i, sd in enumerate(state):
                self.lag_optims[i].load_state_dict(sd)

    def lagrangians_and_rescaling(self):
        """(lambda_i list, rescaling) exactly as safety_loss derives them (lagrangian_base.py:153-156)."""
        lags = [optim.get_lag() for optim in self.lag_optims]
        rescaling = 1. / (np.sum(lags) + 1) if self.rescaling else 1
        return lags, float(rescaling)
