"""BasePolicy: the contract trainer and collector rely on (fsrl/policy/base_policy.py:86-355),
with `update()` delegated to the HIP engine by the concrete policies.

The actor / critic `nn.Module`s are kept as the HOST MIRROR of the device parameters: they give
`state_dict()` the reference's exact key names and serve collector-time inference on the CPU;
`_push_params()` / `_pull_params()` move the flat vector across the C ABI."""
from abc import ABC, abstractmethod
from typing import Any, List, Optional, Union

import numpy as np
import torch
from torch import nn

from fsrl_amd.data.batch import Batch
from fsrl_amd.utils.logger import BaseLogger, DummyLogger
from fsrl_amd.utils.net import ActorCritic


class DeviceBatch:
    """What `process_fn` returns on the HIP path: a HANDLE to the processed batch, which lives in HBM (buffer.sample(0) order:
    env-major, chronological) -- the reference's process_fn returns the Batch with `values / rets / advs / logp_old` attached
    (fsrl/policy/base_policy.py:384-451, ppo_lag.py:134-150); here those columns are fetched from the device on first access.
    `learn(batch, ...)` takes exactly this object back."""

    def __init__(self, engine, n: int, batch_size: int):
        self.engine, self.n, self.batch_size = engine, int(n), int(batch_size)
        self._cache = {}

    def __len__(self) -> int:
        return self.n

    def _get(self, which: str):
        if which not in self._cache:
            self._cache[which] = torch.from_numpy(np.ascontiguousarray(self.engine.batch_get(which)))
        return self._cache[which]

    values = property(lambda self: self._get("values"))          # [N, critics] V(obs) at process time
    rets = property(lambda self: self._get("rets"))              # [N, critics] GAE returns
    advs = property(lambda self: self._get("advs"))              # [N, critics] advantages (CPO / TRPO-Lag: normalised over the batch)
    logp_old = property(lambda self: self._get("logp_old"))      # [N]


class ReplayDeviceBatch:
    """What the replay agents' `process_fn` returns (SAC-Lag, DDPG-Lag, CVPO): the SAMPLE of the update -- the indices into the
    HBM-resident store and the noise the reference's process_fn consumed (its `_target_q` forward at s_{t+n}:
    fsrl/policy/sac_lag.py:136-150, cvpo.py:204-218).  The reference's process_fn also attaches the n-step returns
    (base_policy.py:453-567); here the float64 n-step targets are formed inside the critics' launch of `learn`, from exactly these
    indices, so the handle carries no tensor.  `indices is None` = the library's own Philox sample (drawn on the device in learn).
    `learn(batch)` takes this object."""

    def __init__(self, engine, n: int, indices=None, eps_target=None):
        self.engine, self.n = engine, int(n)
        self.indices = None if indices is None else np.ascontiguousarray(indices, np.int64).reshape(-1)
        self.eps_target = eps_target
        assert self.indices is None or self.indices.size == self.n

    def __len__(self) -> int:
        return self.n


class BasePolicy(ABC, nn.Module):
    def __init__(self, actor: nn.Module, critics: Union[nn.Module, List[nn.Module]], dist_fn=None,
                 logger: BaseLogger = None, gamma: float = 0.99, max_batchsize: Optional[int] = 99999,
                 reward_normalization: bool = False, deterministic_eval: bool = True,
                 action_scaling: bool = True, action_bound_method: str = "clip",
                 observation_space=None, action_space=None, lr_scheduler=None) -> None:
        super().__init__()
        # the attribute names are the drop-in surface: state_dict() keys, the collector and the trainers read them
        # (fsrl/policy/base_policy.py:96-130)
        assert 0.0 <= gamma <= 1.0, "discount factor should be in [0, 1]."
        assert action_bound_method in self._BOUND
        self.actor, self.critics = actor, self._as_module_list(critics)
        self.critics_num = len(self.critics)
        self._actor_critic = ActorCritic(self.actor, self.critics)
        self.dist_fn, self.lr_scheduler = dist_fn, lr_scheduler
        self.logger = DummyLogger() if logger is None else logger
        self._gamma, self._rew_norm, self._max_batchsize = gamma, bool(reward_normalization), max_batchsize
        self._deterministic_eval = deterministic_eval
        self.observation_space, self.action_space, self.action_type = observation_space, action_space, "continuous"
        self.action_scaling, self.action_bound_method = action_scaling, action_bound_method
        self.updating, self.gradient_steps = False, 0
        self.engine = None  # set by the concrete policy

    @staticmethod
    def _as_module_list(critics):
        if isinstance(critics, nn.Module):
            return nn.ModuleList([critics])
        if isinstance(critics, (list, tuple)):
            return nn.ModuleList(critics)
        raise TypeError("critics should not be %s" % (type(critics)))

    # ------------------------------------------------------------------ engine
    def _make_engine(self, device, env_num, buffer_size, optim=None, **cfg_over):
        """Create the HIP context with the geometry of the host networks (hidden_sizes of any depth / width: two layers of at
        most 256 units run on the fused kernels, anything else as a layered context)."""
        from fsrl_amd.engine import Engine, EngineConfig
        from fsrl_amd.utils.net import mlp_geometry
        obs_dim, hidden_sizes = mlp_geometry(self.actor.preprocess)
        act_dim = self.actor.mu.model[0].weight.shape[0]
        dev = device if isinstance(device, int) else (int(str(device).split(":")[-1]) if ":" in str(device) else 0)
        kw = dict(obs_dim=int(obs_dim), act_dim=int(act_dim), hidden_sizes=hidden_sizes, n_critics=self.critics_num,
                  env_num=int(env_num), buffer_size=int(buffer_size),
                  max_action=float(getattr(self.actor, "_max", 1.0)), gamma=self._gamma,
                  unbounded=bool(getattr(self.actor, "_unbounded", False)), rew_norm=self._rew_norm)
        if optim is not None:
            g = optim.param_groups[0]
            kw.update(lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], adam_eps=g["eps"])
        kw.update(cfg_over)
        self.engine = Engine(EngineConfig(**kw), device=dev)
        self._push_params()

    @property
    def ret_rms(self):
        """BasePolicy.ret_rms of the reference (base_policy.py:111): one running mean / var / count of the normalised returns
        per critic, read from the engine (`fsrl_ret_rms_get`) -- a list of objects with tianshou's RunningMeanStd attribute
        names.  Without reward_normalization: the untouched initial state (mean 0, var 1, count 0)."""
        from types import SimpleNamespace
        rows = (self.engine.ret_rms_get() if self._rew_norm and self.engine is not None
                else np.array([[0.0, 1.0, 0.0]] * self.critics_num))
        return [SimpleNamespace(mean=float(m), var=float(v), count=float(n)) for m, v, n in rows]

    # ------------------------------------------------------------------ parameter plumbing
    def _flat_params(self, fresh: bool = True) -> np.ndarray:
        """The host mirror as one vector (refreshed from the device first when an update has run since)"""
        if fresh and getattr(self, "_stale", False):
            self._pull_params()
        return torch.cat([p.detach().reshape(-1) for p in self._actor_critic.parameters()]).numpy().astype(np.float32)

    def _push_params(self) -> None:
        self._stale = False                      # the mirror is the source now
        self.engine.set_params(self._flat_params(fresh=False))

    def _mark_stale(self) -> None:
        """The device parameters moved on (an update ran): the host mirror is refreshed when it is next needed -- acting on
        the host, state_dict() -- not after every update (collecting with the device actor never needs it)."""
        self._stale = True

    def _pull_params(self) -> None:
        self._stale = False
        flat = torch.from_numpy(self.engine.get_params())
        off = 0
        with torch.no_grad():
            for p in self._actor_critic.parameters():
                n = p.numel()
                p.copy_(flat[off:off + n].view_as(p))
                off += n

    def state_dict(self, *args, **kwargs):
        if getattr(self, "_stale", False):
            self._pull_params()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True):
        out = super().load_state_dict(state_dict, strict=strict)
        if self.engine is not None:
            self._push_params()
            self._stale = False
        return out

    # ------------------------------------------------------------------ learning-rate schedule
    def _lr_groups(self):
        """(engine optimiser group, host torch optimiser) pairs; include/fsrl_hip.h fsrl_set_lr names the groups."""
        groups = []
        if getattr(self, "optim", None) is not None:
            groups.append((0, self.optim))
        if getattr(self, "actor_optim", None) is not None:
            groups.append((0, self.actor_optim))
        if getattr(self, "critics_optim", None) is not None:
            groups.append((1, self.critics_optim))
        if getattr(self, "_alpha_optim", None) is not None:
            groups.append((2, self._alpha_optim))
        return groups

    def _step_lr_scheduler(self) -> None:
        """End of BasePolicy.update (fsrl/policy/base_policy.py:352-354): step the caller's scheduler on the host
        optimisers, then move every group's new rate into the engine -- the kernels read the engine's copy."""
        if self.lr_scheduler is None:
            return
        self.lr_scheduler.step()
        for gid, opt in self._lr_groups():
            self.engine.set_lr(gid, float(opt.param_groups[0]["lr"]))

    # ------------------------------------------------------------------ acting
    def forward(self, batch: Batch, state=None, **kwargs: Any) -> Batch:
        if getattr(self, "_stale", False):
            self._pull_params()
        logits, hidden = self.actor(batch.obs, state=state)
        dist = self.dist_fn(*logits) if isinstance(logits, tuple) else self.dist_fn(logits)
        if self._deterministic_eval and not self.training:
            act = logits[0]
        else:
            act = dist.sample()
        return Batch(logits=logits, act=act, state=hidden, dist=dist)

    def pre_update_fn(self, **kwarg: Any) -> Any:
        pass

    def post_update_fn(self, **kwarg: Any) -> Any:
        pass

    def exploration_noise(self, act, batch):
        return act

    # The two action maps (reference semantics: fsrl/policy/base_policy.py:226-290; the device applies the forward map
    # itself inside fsrl_collect_step with the same bound codes: 0 none, 1 clip, 2 tanh).
    _BOUND = {"": 0, "clip": 1, "tanh": 2}

    def _act_range(self):
        """(low, width) of the env's action box, or None when actions pass through unscaled"""
        sp = self.action_space
        if sp is None or not self.action_scaling:
            return None
        return sp.low, sp.high - sp.low

    def map_action(self, act):
        """policy output -> env action: squash into [-1, 1] (clip or tanh), then stretch onto [low, high]"""
        if self.action_space is None or not isinstance(act, np.ndarray):
            return act
        code = self._BOUND[self.action_bound_method]
        unit = np.clip(act, -1.0, 1.0) if code == 1 else (np.tanh(act) if code == 2 else act)
        rng = self._act_range()
        if rng is None:
            return unit
        assert np.min(unit) >= -1.0 and np.max(unit) <= 1.0, "action scaling only accepts raw action range = [-1, 1]"
        low, width = rng
        return low + width * (unit + 1.0) / 2.0

    def map_action_inverse(self, act):
        """env action -> the policy's raw output space (used for random-action collection): undo the stretch, then atanh"""
        unit = np.asarray(act)
        if self.action_space is None:
            return unit
        rng = self._act_range()
        if rng is not None:
            low, width = rng
            tiny = np.finfo(np.float32).eps.item()
            width = np.where(width < tiny, width + tiny, width)       # a degenerate dimension must not divide by zero
            unit = (unit - low) * 2.0 / width - 1.0
        if self._BOUND[self.action_bound_method] == 2:
            unit = (np.log(1.0 + unit) - np.log(1.0 - unit)) / 2.0
        return unit

    # ------------------------------------------------------------------ update
    @abstractmethod
    def learn(self, batch, **kwargs: Any):
        """The on-policy agents (PPO-Lag, FOCOPS, CPO, TRPO-Lag): `process_fn(batch, buffer, indices)` returns a DeviceBatch (the
        library's begin call), `learn(that batch, batch_size, repeat)` runs the passes (the library's pass / learn / end calls);
        `update()` is the two in a row, as fsrl/policy/base_policy.py:332-355.  The replay agents (SAC-Lag, DDPG-Lag, CVPO):
        `process_fn(batch, buffer, indices)` returns a ReplayDeviceBatch (the sample: indices + the noise process_fn consumed),
        `learn(that batch)` is the ONE fused library call (n-step targets and the optimiser steps on the device); `update()` is
        buffer.sample_indices -> process_fn -> learn."""

    @abstractmethod
    def update(self, sample_size: int, buffer, **kwargs: Any):
        pass
