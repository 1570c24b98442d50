"""FOCOPS over the HIP engine: constructor arguments and logger keys of fsrl/policy/focops.py:64-251
(SURVEY 8f rank 4).  `update()` = process_fn + learn on the MI355X through `fsrl_focops_set_nu` +
`fsrl_ppo_begin / fsrl_ppo_pass / fsrl_ppo_end` of an FSRL_ALGO_FOCOPS context; the nu step
(focops.py:154-159) is host float32 arithmetic like the reference's tensor."""
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn

from fsrl_amd import _lib
from fsrl_amd.policy.base_policy import BasePolicy, DeviceBatch
from fsrl_amd.policy.ppo_lag import _chunk_sizes

FOCOPS_KEYS = ("loss/nu_loss", "loss/nu_value", "loss/actor_loss", "loss/kl", "loss/entropy", "loss/vf0", "loss/vf1",
               "loss/vf_total")


class FOCOPS(BasePolicy):
    def __init__(self, actor: nn.Module, critics: Union[nn.Module, List[nn.Module]], actor_optim: torch.optim.Optimizer,
                 critic_optim: torch.optim.Optimizer, dist_fn, logger=None, cost_limit: float = 10,
                 nu: Union[float, Tuple[float, float, torch.Tensor]] = 0.01, l2_reg: float = 1e-3, delta: float = 0.02,
                 eta: float = 0.02, tem_lambda: float = 0.95, gae_lambda: float = 0.95, max_grad_norm: Optional[float] = 0.5,
                 advantage_normalization: bool = True, recompute_advantage: bool = False, gamma: float = 0.99,
                 max_batchsize: int = 99999, reward_normalization: bool = False, deterministic_eval: bool = True,
                 action_scaling: bool = True, action_bound_method: str = "clip", observation_space=None,
                 action_space=None, lr_scheduler=None, device: Union[int, str] = 0, env_num: int = 1,
                 buffer_size: int = 100000, reference_rng: bool = False) -> None:
        super().__init__(actor, critics, dist_fn, logger, gamma, max_batchsize, reward_normalization, deterministic_eval,
                         action_scaling, action_bound_method, observation_space, action_space, lr_scheduler)
        assert self.critics_num == 2, "FOCOPS uses a reward and a cost critic"
        assert isinstance(nu, tuple), "the reference's nu_loss needs the (nu_max, nu_lr, nu) form (focops.py:154-159)"
        self.actor_optim, self.critics_optim = actor_optim, critic_optim
        self.cost_limit = cost_limit
        self._nu_max, self._nu_lr, self._nu = nu
        self._is_auto_nu = True
        self._ave_cost_return = 0.0
        self._reference_rng = reference_rng     # burn the torch draws the reference's forward() wastes in update()
        self._make_engine(device, env_num, buffer_size, actor_optim, algo=_lib.ALGO_FOCOPS, gae_lambda=gae_lambda,
                          norm_adv=advantage_normalization, target_kl=None, recompute_adv=bool(recompute_advantage))
        self.engine.focops_init(actor_lr=actor_optim.param_groups[0]["lr"], critic_lr=critic_optim.param_groups[0]["lr"],
                                l2_reg=l2_reg, delta=delta, eta=eta, tem_lambda=tem_lambda, max_grad_norm=max_grad_norm)

    def pre_update_fn(self, stats_train: Dict, **kwarg) -> Any:
        self._ave_cost_return = stats_train["cost"]

    def update_cost_limit(self, cost_limit: float) -> None:
        self.cost_limit = cost_limit

    def process_fn(self, batch=None, buffer=None, indices=None, batch_size: int = 256):
        """focops.py:126-153 on the device: the nu step (:155-158, host float32) + `fsrl_ppo_begin` (sample(0), V / GAE per critic,
        logp_old, mean_old / std_old).  -> DeviceBatch; `batch` / `indices` ignored (the on-policy batch is the whole store)."""
        assert getattr(buffer, "engine", None) is self.engine
        self.updating = True
        loss_nu = self.cost_limit - self._ave_cost_return            # focops.py:155-158, float32 tensor arithmetic
        self._nu = self._nu + (-self._nu_lr * loss_nu)
        self._nu = torch.clamp(self._nu, 0, self._nu_max)
        eng = self.engine
        _lib.check(eng.lib.fsrl_focops_set_nu(eng._ctx, float(self._nu), float(loss_nu)))
        n = eng.ppo_begin([0.0], 1.0, batch_size)
        try:
            if self._reference_rng and (self.training or not self._deterministic_eval):
                da = eng.cfg.act_dim
                for m in _chunk_sizes(n, self._max_batchsize):          # process_fn: forward per chunk of max_batchsize
                    torch.normal(torch.zeros(m, da), torch.ones(m, da))
        except BaseException:
            eng.ppo_abort(); self.updating = False
            raise
        self._pending = DeviceBatch(eng, n, batch_size)
        return self._pending

    def learn(self, batch, batch_size: int = 256, repeat: int = 4, **kwargs: Any):
        """focops.py:205-251 on the device: `fsrl_ppo_pass` x repeat + `fsrl_ppo_end`; `batch` is what process_fn returned."""
        assert isinstance(batch, DeviceBatch) and batch is getattr(self, "_pending", None), \
            "learn() takes the DeviceBatch the last process_fn() returned (the processed batch lives in HBM)"
        self._pending = None
        eng, n = self.engine, batch.n
        try:
            if batch.batch_size != batch_size:                       # another minibatch size than planned: begin again
                eng.ppo_abort()
                n = eng.ppo_begin([0.0], 1.0, batch_size)
            burn = self._reference_rng and (self.training or not self._deterministic_eval)
            da = eng.cfg.act_dim
            stopped_at = -1
            perm = np.random.permutation(n) if n > 0 else None           # Batch.split(shuffle=True) of the first pass
            for step in range(repeat):
                if burn:                                                 # policy_loss: forward per minibatch
                    for m in _chunk_sizes(n, batch_size):
                        torch.normal(torch.zeros(m, da), torch.ones(m, da))
                eng.ppo_pass(perm, wait=False)
                # next permutation while the device works; rolled back if this pass was the last (see ppo_lag.py here)
                rng_state = np.random.get_state() if (n > 0 and step + 1 < repeat) else None
                if rng_state is not None:
                    perm = np.random.permutation(n)
                if eng.ppo_pass_result():
                    if rng_state is not None:
                        np.random.set_state(rng_state)
                    stopped_at = step
                    self.logger.print("Early stop at step %d due to reaching max kl." % step)
                    break
            stats = eng.ppo_end_stats(max(1, -(-n // max(batch_size, 1))) * max(repeat, 1))[:, :_lib.FOCOPS_NSTATS]
        except BaseException:                                        # never leave the library between begin and end
            eng.ppo_abort()
            self.updating = False
            raise
        table = getattr(self.logger, "store_rows", None)         # fsrl_amd loggers take the per-step table at once
        if table is not None:
            table(FOCOPS_KEYS, stats)
        for row in (stats if table is None else ()):
            d = dict(zip(FOCOPS_KEYS, (float(v) for v in row)))
            self.logger.store(**{k: d[k] for k in FOCOPS_KEYS[:2]})
            self.logger.store(**{k: d[k] for k in FOCOPS_KEYS[2:5]})
            self.logger.store(**{k: d[k] for k in FOCOPS_KEYS[5:]})
        self.gradient_steps += len(stats)
        self.logger.store(gradient_steps=self.gradient_steps, tab="update")
        self._mark_stale()                                       # host mirror refreshed on demand
        return {"gradient_steps": len(stats), "early_stop_pass": stopped_at}

    def update(self, sample_size: int, buffer, batch_size: int = 256, repeat: int = 4, **kwargs: Any):
        """base_policy.py:332-355: sample(0) -> process_fn -> learn -> lr scheduler"""
        if buffer is None:
            return {}
        assert sample_size == 0
        batch = self.process_fn(None, buffer, None, batch_size=batch_size)
        result = self.learn(batch, batch_size=batch_size, repeat=repeat)
        self._step_lr_scheduler()
        self.updating = False
        return result
