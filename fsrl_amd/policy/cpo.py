"""CPO over the HIP engine: constructor arguments and logger keys of fsrl/policy/cpo.py:16-370.
`update()` = process_fn (GAE, full-batch advantage normalisation, old log-prob / mean / std) +
`repeat` x { optim_critic_iters critic Adam steps ; policy_loss: g, b, two CG solves with exact
Hessian-vector products, 5-case dual solve, backtracking line search } on the MI355X through
`fsrl_tr_begin` / `fsrl_cpo_learn` (include/fsrl_hip.h)."""
from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch
from torch import nn

from fsrl_amd.policy.base_policy import BasePolicy, DeviceBatch
from fsrl_amd.policy.trpo_lag import _split_sizes

CPO_ACTOR_KEYS = ("loss/kl", "loss/entropy", "loss/rew_loss", "loss/cost_loss", "loss/optim_A",
                  "loss/optim_B", "loss/optim_C", "loss/optim_Q", "loss/optim_R", "loss/optim_S",
                  "loss/optim_lam", "loss/optim_nu", "loss/optim_case", "loss/step_size")
CPO_CRITIC_KEYS = ("loss/vf0", "loss/vf1", "loss/vf_total")


class CPO(BasePolicy):
    def __init__(self, actor: nn.Module, critics: Union[nn.Module, List[nn.Module]],
                 optim: torch.optim.Optimizer, dist_fn, logger=None,
                 # CPO specific arguments
                 target_kl: float = 0.01, backtrack_coeff: float = 0.8, damping_coeff: float = 0.1,
                 max_backtracks: int = 10, optim_critic_iters: int = 20, l2_reg: float = 0.001,
                 gae_lambda: float = 0.95, advantage_normalization: bool = True,
                 cost_limit: Union[List, float] = np.inf,
                 # Base policy common arguments
                 gamma: float = 0.99, max_batchsize: int = 99999, reward_normalization: bool = False,
                 deterministic_eval: bool = True, action_scaling: bool = True,
                 action_bound_method: str = "clip", observation_space=None, action_space=None,
                 lr_scheduler=None, device: Union[int, str] = 0, env_num: int = 1,
                 buffer_size: int = 100000, reference_rng: bool = False) -> None:
        super().__init__(actor, critics, dist_fn, logger, gamma, max_batchsize, reward_normalization,
                         deterministic_eval, action_scaling, action_bound_method, observation_space,
                         action_space, lr_scheduler)
        assert self.critics_num == 2, "CPO does not support multiple costs"
        self.optim = optim
        self._reference_rng = reference_rng                      # Adam over the critic parameters only (cpo_agent.py:147)
        self._cost_limit = cost_limit
        self._lambda, self._norm_adv = gae_lambda, advantage_normalization
        self._max_backtracks, self._optim_critic_iters = max_backtracks, optim_critic_iters
        self._l2_reg, self._delta = l2_reg, target_kl
        self._backtrack_coeff, self._damping_coeff = backtrack_coeff, damping_coeff
        self._ave_cost_return = 0.0
        self._make_engine(device, env_num, buffer_size, optim, gae_lambda=gae_lambda, target_kl=None)

    def pre_update_fn(self, stats_train: Dict, **kwarg) -> Any:
        self._ave_cost_return = stats_train["cost"]

    def update_cost_limit(self, cost_limit: float) -> None:
        self._cost_limit = cost_limit

    def _burn(self, n_rows: int, forwards: int) -> None:
        """The reference's forward() samples an action ([n, Da] normals from torch's stream) every time it is
        called in training mode, also inside update(); reference_rng=True consumes the same amount."""
        if self._reference_rng and (self.training or not self._deterministic_eval):
            da = self.engine.cfg.act_dim
            for _ in range(forwards):
                torch.normal(torch.zeros(n_rows, da), torch.ones(n_rows, da))

    def process_fn(self, batch=None, buffer=None, indices=None, **kwargs: Any):
        """cpo.py:123-145 on the device = `fsrl_tr_begin`: sample(0), V / GAE per critic, full-batch advantage normalisation,
        logp_old / mean_old / std_old.  -> DeviceBatch; `batch` / `indices` ignored (the on-policy batch is the whole store)."""
        assert getattr(buffer, "engine", None) is self.engine
        self.updating = True
        g = self.optim.param_groups[0]
        n = self.engine.tr_begin(target_kl=self._delta, backtrack_coeff=self._backtrack_coeff,
                                 damping=self._damping_coeff, l2_reg=self._l2_reg, critic_lr=g["lr"],
                                 max_backtracks=self._max_backtracks, optim_critic_iters=self._optim_critic_iters,
                                 cg_iters=10, norm_adv=self._norm_adv, cost_limit=float(self._cost_limit))
        self._pending = DeviceBatch(self.engine, n, 0)
        return self._pending

    def learn(self, batch, batch_size: int = 99999, repeat: int = 4, **kwargs: Any):
        """cpo.py:353-370 on the device = `fsrl_cpo_learn[_mb]`; `batch` is what process_fn returned."""
        assert isinstance(batch, DeviceBatch) and batch is getattr(self, "_pending", None), \
            "learn() takes the DeviceBatch the last process_fn() returned (the processed batch lives in HBM)"
        self._pending = None
        eng, n = self.engine, batch.n
        # Batch.split(batch_size, merge_last=True) inside learn (cpo.py:357-358) draws one np.random.permutation per repeat
        # from numpy's global stream -- also when one minibatch covers the batch (then the order only moves sums and the
        # device keeps store order)
        perms = [np.random.permutation(n) for _ in range(repeat)] if n > 0 else None
        sizes = _split_sizes(n, batch_size)
        stats = (eng.cpo_learn(float(self._ave_cost_return), repeat, batch_size=batch_size,
                               perms=perms if len(sizes) > 1 else None) if n > 0 else np.zeros((0, 17), np.float32))
        for row in stats:
            self.gradient_steps += 1
            self.logger.store(**dict(zip(CPO_ACTOR_KEYS, (float(v) for v in row[:14]))))
            self.logger.store(**dict(zip(CPO_CRITIC_KEYS, (float(v) for v in row[14:]))))
        self.logger.store(gradient_steps=self.gradient_steps, tab="update")
        if n > 0:   # process_fn: one forward over the batch; per minibatch: 1 forward(s) + one per line-search evaluation
            self._burn(n, 1)
            for rows, ev in zip(sizes * repeat, eng.tr_linesearch_evals(cap=len(stats) + 1)):
                self._burn(rows, 1 + int(ev))
        self._mark_stale()                                       # host mirror refreshed on demand
        return {"gradient_steps": len(stats)}

    def update(self, sample_size: int, buffer, batch_size: int = 99999, repeat: int = 4, **kwargs: Any):
        """base_policy.py:332-355: sample(0) -> process_fn -> learn -> lr scheduler"""
        if buffer is None:
            return {}
        assert sample_size == 0
        try:
            batch = self.process_fn(None, buffer, None)
            result = self.learn(batch, batch_size=batch_size, repeat=repeat)
        finally:
            self.updating = False
        self._step_lr_scheduler()
        return result
