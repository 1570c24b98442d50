"""SACLagrangian over the HIP engine: constructor arguments and logger keys of
fsrl/policy/sac_lag.py:16-277.  `update(batch_size, buffer)` = sample + n-step soft targets +
critic step + actor step + alpha step + Polyak on the MI355X through `fsrl_sac_update`; the
sampled indices (numpy RNG, tianshou's sub-buffer-proportional rule) and the two rsample noise
draws (torch RNG) are produced here so the random streams match the reference's."""
from copy import deepcopy
from typing import Any, List, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn
from torch.distributions import Independent, Normal

from fsrl_amd import _lib
from fsrl_amd.data.batch import Batch
from fsrl_amd.engine import Engine, EngineConfig
from fsrl_amd.policy.base_policy import ReplayDeviceBatch
from fsrl_amd.policy.lagrangian_base import LagrangianPolicy

SAC_KEYS = ("loss/rescaling", "loss/lagrangian", "loss/actor_safety", "loss/alpha_loss", "loss/alpha_value",
            "loss/actor_rew", "loss/actor_total", "loss/q0", "loss/q1", "loss/q_total")


class SACLagrangian(LagrangianPolicy):
    def __init__(self, actor: nn.Module, critics: Union[nn.Module, List[nn.Module]], actor_optim, critic_optim,
                 logger=None, alpha=0.005, tau: float = 0.05, exploration_noise=None, n_step: int = 2,
                 use_lagrangian: bool = True, lagrangian_pid: Tuple = (0.05, 0.0005, 0.1),
                 cost_limit: Union[List, float] = np.inf, rescaling: bool = True, gamma: float = 0.99,
                 reward_normalization: bool = False, deterministic_eval: bool = True,
                 action_scaling: bool = True, action_bound_method: str = "clip", observation_space=None,
                 action_space=None, lr_scheduler=None, device: Union[int, str] = 0, env_num: int = 1,
                 buffer_size: int = 100000, reference_rng: bool = False, seed: int = 0) -> None:
        super().__init__(actor, critics, None, logger, use_lagrangian, lagrangian_pid, cost_limit, rescaling,
                         gamma, 10000, reward_normalization, deterministic_eval, action_scaling,
                         action_bound_method, observation_space, action_space, lr_scheduler)
        assert self.critics_num == 2, "the HIP path supports one cost constraint (reward + cost double critics)"
        assert 0.0 <= tau <= 1.0, "tau should be in [0, 1]"
        self.actor_optim, self.critics_optim = actor_optim, critic_optim
        self.critics_old = deepcopy(self.critics)
        self.critics_old.eval()
        self.tau, self._n_step, self._noise = tau, n_step, exploration_noise
        self._is_auto_alpha = isinstance(alpha, tuple)
        if self._is_auto_alpha:
            self._target_entropy, self._log_alpha, self._alpha_optim = alpha
            assert alpha[1].shape == torch.Size([1]) and alpha[1].requires_grad
            self._alpha = self._log_alpha.detach().exp()
            alpha_lr, alpha_fixed = self._alpha_optim.param_groups[0]["lr"], 0.0
        else:
            self._alpha, self._target_entropy, alpha_lr, alpha_fixed = alpha, None, 3e-4, float(alpha)
        self.__eps = np.finfo(np.float32).eps.item()
        from fsrl_amd.utils.net import mlp_geometry
        obs_dim, hidden_sizes = mlp_geometry(actor.preprocess)
        act_dim = actor.mu.model[0].weight.shape[0]
        dev = device if isinstance(device, int) else (int(str(device).split(":")[-1]) if ":" in str(device) else 0)
        self.engine = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=int(obs_dim), act_dim=int(act_dim),
                                          hidden_sizes=hidden_sizes, n_critics=2, env_num=int(env_num),
                                          buffer_size=int(buffer_size), gamma=gamma, target_kl=None),
                             device=dev)
        self.engine.sac_init(actor_lr=actor_optim.param_groups[0]["lr"],
                             critic_lr=critic_optim.param_groups[0]["lr"], alpha_lr=alpha_lr, tau=tau,
                             alpha=alpha_fixed, target_entropy=self._target_entropy, n_step=n_step,
                             auto_alpha=self._is_auto_alpha, use_lagrangian=use_lagrangian)
        self._push_params()
        self._dirty = self._rest_dirty = False
        # reference_rng=True: buffer.sample through numpy's and rsample through torch's global RNG,
        # exactly the streams the reference consumes (bit-comparable runs; one host round trip per
        # update).  False (default): sampling and noise on the device, updates only enqueue work
        # and the logged statistics are drained after the last update of a collect step.
        self._reference_rng, self._seed, self._pending = reference_rng, int(seed), 0

    # ------------------------------------------------------------------ parameter plumbing
    @staticmethod
    def _flat(mods):
        return torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()]).numpy().astype(np.float32)

    @staticmethod
    def _unflat(mods, flat):
        flat, off = torch.from_numpy(flat), 0
        with torch.no_grad():
            for m in mods:
                for p in m.parameters():
                    p.copy_(flat[off:off + p.numel()].view_as(p))
                    off += p.numel()

    def _push_params(self) -> None:
        la = float(self._log_alpha.detach()) if self._is_auto_alpha else 0.0
        self.engine.sac_set_params(self._flat([self.actor]), self._flat(list(self.critics)), la)

    def _pull_params(self, everything: bool = False) -> None:
        th, alpha = self.engine.sac_get_params(0)
        self._unflat([self.actor], th)
        if self._is_auto_alpha:
            self._alpha = torch.tensor([alpha])
            with torch.no_grad():
                self._log_alpha.fill_(float(np.log(alpha)))
        if everything:
            self._unflat(list(self.critics), self.engine.sac_get_params(1)[0])
            self._unflat(list(self.critics_old), self.engine.sac_get_params(2)[0])
        self._dirty = False                      # the actor mirror is current ...
        if everything:
            self._rest_dirty = False             # ... critics / targets only after a full pull

    def state_dict(self, *args, **kwargs):
        if getattr(self, "_pending", 0):
            self._drain()
        if getattr(self, "_dirty", False) or getattr(self, "_rest_dirty", False):
            self._pull_params(everything=True)
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True):
        out = nn.Module.load_state_dict(self, state_dict, strict=strict)
        if getattr(self, "engine", None) is not None:            # actor, critics, targets: each set on its own
            self.engine.sac_put_params(0, self._flat([self.actor]))
            self.engine.sac_put_params(1, self._flat(list(self.critics)))
            self.engine.sac_put_params(2, self._flat(list(self.critics_old)))
            self._dirty = self._rest_dirty = False
        return out

    def train(self, mode: bool = True):
        self.training = mode
        self.actor.train(mode)
        self.critics.train(mode)
        return self

    # ------------------------------------------------------------------ acting (host mirror)
    def forward(self, batch: Batch, state=None, input: str = "obs", **kwargs: Any) -> Batch:
        if self._dirty:
            self._pull_params()
        logits, hidden = self.actor(batch[input], state=state)
        dist = Independent(Normal(*logits), 1)
        act = logits[0] if (self._deterministic_eval and not self.training) else dist.rsample()
        log_prob = dist.log_prob(act).unsqueeze(-1)
        squashed = torch.tanh(act)
        log_prob = log_prob - torch.log((1 - squashed.pow(2)) + self.__eps).sum(-1, keepdim=True)
        return Batch(logits=logits, act=squashed, state=hidden, dist=dist, log_prob=log_prob)

    def exploration_noise(self, act, batch):
        if self._noise is None:
            return act
        return act + self._noise(act.shape) if isinstance(act, np.ndarray) else act

    def _log_rows(self, rows) -> None:
        table = getattr(self.logger, "store_rows", None)         # fsrl_amd loggers take the drained rows at once
        if table is not None and len(rows):
            drop = (() if self._is_auto_alpha else ("loss/alpha_loss", "loss/alpha_value")) + \
                   (() if self.use_lagrangian else ("loss/lagrangian", "loss/actor_safety"))
            cols = [j for j, k in enumerate(SAC_KEYS) if k not in drop]
            table([SAC_KEYS[j] for j in cols], np.asarray(rows)[:, cols])
            return
        for st in rows:
            d = dict(zip(SAC_KEYS, (float(v) for v in st)))
            if not self._is_auto_alpha:
                d.pop("loss/alpha_loss"); d.pop("loss/alpha_value")
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
        """sac_lag.py:136-150 (`compute_nstep_returns` over `_target_q`): what the reference's process_fn consumes is the sample's
        indices and ONE rsample at s_{t+n}; the n-step targets themselves are formed on the device inside learn's critic launch.
        -> ReplayDeviceBatch.  indices=None: the library draws `sample_size` rows itself (Philox, in learn); `batch` is accepted
        for signature parity and ignored (the rows live in the HBM store)."""
        assert getattr(buffer, "engine", None) is self.engine, \
            "SACLagrangian.process_fn needs the HipVectorReplayBuffer bound to this policy's engine"
        self.updating = True
        if indices is None:
            return ReplayDeviceBatch(self.engine, int(sample_size))
        B, Da = len(indices), self.engine.cfg.act_dim
        eps_t = torch.normal(torch.zeros(B, Da), torch.ones(B, Da)).numpy()       # rsample at s_{t+n}
        return ReplayDeviceBatch(self.engine, B, indices, eps_t)

    def learn(self, batch, **kwargs: Any):
        """sac_lag.py:185-269 on the device = `fsrl_sac_update` (critics' step, Polyak targets, actor step, alpha step);
        `batch` is what process_fn returned."""
        assert isinstance(batch, ReplayDeviceBatch) and batch.engine is self.engine and len(batch) >= 1, \
            "learn() takes the ReplayDeviceBatch process_fn() returned (the sampled rows live in HBM)"
        B, Da = len(batch), self.engine.cfg.act_dim
        lags, rescaling = self.lagrangians_and_rescaling() if self.use_lagrangian else ([], 1.0)
        if batch.indices is not None:
            eps_p = torch.normal(torch.zeros(B, Da), torch.ones(B, Da)).numpy()   # rsample at s_t
            st = self.engine.sac_update(B, lags, rescaling, indices=batch.indices, eps_target=batch.eps_target, eps_pi=eps_p)
            self._log_rows(st[None])
        else:
            seed = self._seed + 1 if self.gradient_steps == 0 else 0     # key the Philox stream once
            self.engine.sac_update(B, lags, rescaling, seed=seed, sync=False)
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
        indices = buffer.sample_indices(B) if self._reference_rng else None      # numpy RNG, tianshou rule | device RNG
        result = self.learn(self.process_fn(None, buffer, indices, sample_size=B))
        self._step_lr_scheduler()
        self.updating = False
        return result
