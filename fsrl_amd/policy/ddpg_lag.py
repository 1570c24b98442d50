"""DDPGLagrangian over the HIP engine: constructor arguments and logger keys of
fsrl/policy/ddpg_lag.py:64-231.  `update(batch_size, buffer)` = sample + n-step targets
through the target actor / target critics + critic step + actor step + Polyak of both, on the MI355X
through `fsrl_sac_update` in its deterministic mode."""
from copy import deepcopy
from typing import Any, List, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn

from fsrl_amd import _lib
from fsrl_amd.data.batch import Batch
from fsrl_amd.engine import Engine, EngineConfig
from fsrl_amd.policy.base_policy import ReplayDeviceBatch
from fsrl_amd.policy.lagrangian_base import LagrangianPolicy
from fsrl_amd.policy.sac_lag import SAC_KEYS, SACLagrangian

_DROP = ("loss/alpha_loss", "loss/alpha_value")


class DDPGLagrangian(LagrangianPolicy):
    def __init__(self, actor: nn.Module, critics: Union[nn.Module, List[nn.Module]], actor_optim, critic_optim,
                 logger=None, tau: float = 0.05, exploration_noise=None, n_step: int = 2, use_lagrangian: bool = True,
                 lagrangian_pid: Tuple = (0.05, 0.0005, 0.1), cost_limit: Union[List, float] = np.inf,
                 rescaling: bool = True, gamma: float = 0.99, reward_normalization: bool = False,
                 deterministic_eval: bool = True, action_scaling: bool = True, action_bound_method: str = "clip",
                 observation_space=None, action_space=None, lr_scheduler=None, device: Union[int, str] = 0,
                 env_num: int = 1, buffer_size: int = 100000, reference_rng: bool = False, seed: int = 0) -> None:
        super().__init__(actor, critics, None, logger, use_lagrangian, lagrangian_pid, cost_limit, rescaling, gamma,
                         99999, reward_normalization, deterministic_eval, action_scaling, action_bound_method,
                         observation_space, action_space, lr_scheduler)
        assert self.critics_num == 2, "the HIP path supports one cost constraint (reward + cost critics)"
        assert 0.0 <= tau <= 1.0, "tau should be in [0, 1]"
        self.actor_old = deepcopy(self.actor)
        self.actor_old.eval()
        self.actor_optim, self.critics_optim = actor_optim, critic_optim
        self.critics_old = deepcopy(self.critics)
        self.critics_old.eval()
        self.tau, self._noise, self._n_step = tau, exploration_noise, n_step
        from fsrl_amd.utils.net import mlp_geometry
        obs_dim, hidden_sizes = mlp_geometry(actor.preprocess)
        act_dim = actor.last.model[0].weight.shape[0]
        dev = device if isinstance(device, int) else (int(str(device).split(":")[-1]) if ":" in str(device) else 0)
        self.engine = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=int(obs_dim), act_dim=int(act_dim),
                                          hidden_sizes=hidden_sizes, n_critics=2, env_num=int(env_num),
                                          buffer_size=int(buffer_size), gamma=gamma, max_action=float(actor._max),
                                          target_kl=None), device=dev)
        self.engine.sac_init(actor_lr=actor_optim.param_groups[0]["lr"], critic_lr=critic_optim.param_groups[0]["lr"],
                             tau=tau, n_step=n_step, use_lagrangian=use_lagrangian, deterministic=True,
                             exploration_sigma=getattr(exploration_noise, "_sigma", 0.0))
        self.engine.sac_set_params(SACLagrangian._flat([self.actor]), SACLagrangian._flat(list(self.critics)), 0.0)
        self._dirty, self._reference_rng, self._seed, self._pending = False, reference_rng, int(seed), 0
        self._rest_dirty = False

    def set_exp_noise(self, noise) -> None:
        self._noise = noise

    def train(self, mode: bool = True):
        self.training = mode
        self.actor.train(mode)
        self.critics.train(mode)
        return self

    def _pull_params(self, everything: bool = False) -> None:
        SACLagrangian._unflat([self.actor], self.engine.sac_get_params(0)[0])
        if everything:
            SACLagrangian._unflat([self.actor_old], self.engine.sac_get_params(3)[0])
            SACLagrangian._unflat(list(self.critics), self.engine.sac_get_params(1)[0])
            SACLagrangian._unflat(list(self.critics_old), self.engine.sac_get_params(2)[0])
        self._dirty = False                      # the actor mirror is current ...
        if everything:
            self._rest_dirty = False             # ... critics / targets only after a full pull

    def state_dict(self, *args, **kwargs):
        if self._pending:
            self._drain()
        if self._dirty or self._rest_dirty:
            self._pull_params(everything=True)
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True):
        out = nn.Module.load_state_dict(self, state_dict, strict=strict)
        if getattr(self, "engine", None) is not None:
            for which, mods in ((0, [self.actor]), (3, [self.actor_old]), (1, list(self.critics)), (2, list(self.critics_old))):
                self.engine.sac_put_params(which, SACLagrangian._flat(mods))
            self._dirty = self._rest_dirty = False
        return out

    def forward(self, batch: Batch, state=None, model: str = "actor", input: str = "obs", **kwargs: Any) -> Batch:
        if self._dirty or (model != "actor" and self._rest_dirty):
            self._pull_params(everything=model != "actor")
        actions, hidden = getattr(self, model)(batch[input], state=state)
        return Batch(act=actions, state=hidden)

    def exploration_noise(self, act, batch):
        if self._noise is None:
            return act
        return act + self._noise(act.shape) if isinstance(act, np.ndarray) else act

    def _log_rows(self, rows) -> None:
        table = getattr(self.logger, "store_rows", None)         # fsrl_amd loggers take the drained rows at once
        if table is not None and len(rows):
            drop = _DROP + (() if self.use_lagrangian else ("loss/lagrangian", "loss/actor_safety"))
            cols = [j for j, k in enumerate(SAC_KEYS) if k not in drop]
            table([SAC_KEYS[j] for j in cols], np.asarray(rows)[:, cols])
            return
        for st in rows:
            d = {k: float(v) for k, v in zip(SAC_KEYS, st) if k not in _DROP}
            if not self.use_lagrangian:
                d.pop("loss/lagrangian"); d.pop("loss/actor_safety")
            qs = {k: d.pop(k) for k in ("loss/q0", "loss/q1", "loss/q_total")}
            self.logger.store(**d)
            self.logger.store(**qs)

    def _drain(self) -> None:
        if self._pending:
            self._log_rows(self.engine.sac_drain())
            self._pending = 0

    def post_update_fn(self, **kwarg: Any) -> None:
        self._drain()
        super().post_update_fn(**kwarg)

    def process_fn(self, batch=None, buffer=None, indices=None, sample_size: int = 0):
        """ddpg_lag.py:125-140 (`compute_nstep_returns` over `_target_q`: the target actor is deterministic, no noise is drawn):
        the sample's indices; the n-step targets are formed on the device inside learn's critic launch.  -> ReplayDeviceBatch.
        indices=None: the library draws `sample_size` rows itself (Philox, in learn); `batch` is ignored (the rows live in HBM)."""
        assert getattr(buffer, "engine", None) is self.engine, \
            "DDPGLagrangian.process_fn needs the HipVectorReplayBuffer bound to this policy's engine"
        self.updating = True
        if indices is None:
            return ReplayDeviceBatch(self.engine, int(sample_size))
        return ReplayDeviceBatch(self.engine, len(indices), indices)

    def learn(self, batch, **kwargs: Any):
        """ddpg_lag.py:178-223 on the device = `fsrl_sac_update` of a deterministic-actor context; `batch` is what process_fn
        returned."""
        assert isinstance(batch, ReplayDeviceBatch) and batch.engine is self.engine and len(batch) >= 1, \
            "learn() takes the ReplayDeviceBatch process_fn() returned (the sampled rows live in HBM)"
        B = len(batch)
        lags, rescaling = self.lagrangians_and_rescaling() if self.use_lagrangian else ([], 1.0)
        if batch.indices is not None:
            zero = np.zeros((B, self.engine.cfg.act_dim), np.float32)
            st = self.engine.sac_update(B, lags, rescaling, indices=batch.indices, eps_target=zero, eps_pi=zero)
            self._log_rows(st[None])
        else:
            self.engine.sac_update(B, lags, rescaling, seed=self._seed + 1 if self.gradient_steps == 0 else 0, sync=False)
            self._pending += 1
            if self._pending >= 2048:
                self._drain()
        self.gradient_steps += 1
        self._dirty = self._rest_dirty = True      # actor mirror AND critics / targets are behind the device now
        return {}

    def update(self, sample_size: int, buffer, **kwargs: Any):
        """base_policy.py:332-355: buffer.sample -> process_fn -> learn -> lr scheduler"""
        if buffer is None:
            return {}
        assert getattr(buffer, "engine", None) is self.engine
        B = int(sample_size)
        indices = buffer.sample_indices(B) if self._reference_rng else None      # numpy RNG, like the reference | device RNG
        result = self.learn(self.process_fn(None, buffer, indices, sample_size=B))
        self._step_lr_scheduler()
        self.updating = False
        return result
