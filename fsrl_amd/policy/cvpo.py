"""CVPO over the HIP engine: constructor arguments and logger keys of fsrl/policy/cvpo.py:71-430.
`update(batch_size, buffer)` = sample + n-step targets + critic step + E-step (K particles, Adam on eta / lambda,
softmax weights) + M-step (weighted likelihood + KL multipliers) + Polyak on the MI355X through
`fsrl_cvpo_update`; `pre_update_fn` / `post_update_fn` reset the M-step multipliers / copy actor -> actor_old on
the device.  The two wall-clock keys the reference logs (estep/estep_time, mstep/mstep_time) are not produced:
both phases are one enqueue here."""
from copy import deepcopy
from typing import Any, List, Optional, Type, Union

import numpy as np
import torch
from torch import nn

from fsrl_amd import _lib
from fsrl_amd.data.batch import Batch
from fsrl_amd.engine import Engine, EngineConfig
from fsrl_amd.policy.base_policy import BasePolicy, ReplayDeviceBatch
from fsrl_amd.policy.sac_lag import SACLagrangian

# row of FSRL_CVPO_NSTATS floats -> the reference's logger keys (tab, key)
CVPO_KEYS = (("loss", "estep_loss"), (None, "estep/dual0"), (None, "estep/dual1"), ("mstep", "mstep_kl_mu"),
             ("mstep", "mstep_kl_std"), ("mstep", "mstep_loss_kl"), ("mstep", "mstep_loss_mle"),
             ("mstep", "mstep_loss_total"), ("mstep", "mstep_dual_mu"), ("mstep", "mstep_dual_std"), ("mstep", "entropy"),
             (None, "loss/loss_q0"), (None, "estep/val_q0"), (None, "loss/loss_q1"), (None, "estep/val_q1"),
             (None, "estep/thres_q1"), (None, "loss/q_total"))


class CVPO(BasePolicy):
    def __init__(self, actor: nn.Module, critics: Union[nn.Module, List[nn.Module]], actor_optim, critic_optim,
                 action_space, dist_fn: Type[torch.distributions.Distribution], max_episode_steps: int, logger=None,
                 cost_limit: Union[List, float] = np.inf, tau: float = 0.05, gamma: float = 0.99, n_step: int = 2,
                 estep_iter_num: int = 1, estep_kl: float = 0.02, estep_dual_max: float = 20, estep_dual_lr: float = 0.02,
                 sample_act_num: int = 16, mstep_iter_num: int = 1, mstep_kl_mu: float = 0.005,
                 mstep_kl_std: float = 0.0005, mstep_dual_max: float = 0.5, mstep_dual_lr: float = 0.1,
                 deterministic_eval: bool = True, action_scaling: bool = True, action_bound_method: str = "clip",
                 lr_scheduler=None, observation_space=None, device: Union[int, str] = 0, env_num: int = 1,
                 buffer_size: int = 100000, reference_rng: bool = False, seed: int = 0) -> None:
        super().__init__(actor, critics, dist_fn, logger, gamma, 99999, False, deterministic_eval, action_scaling,
                         action_bound_method, observation_space, action_space, lr_scheduler)
        assert self.critics_num == 2, "the HIP path supports one cost constraint (reward + cost critics)"
        assert 0.0 <= tau <= 1.0, "tau should be in [0, 1]"
        assert getattr(actor, "_c_sigma", False) and not getattr(actor, "_unbounded", True), \
            "the HIP CVPO path: state-conditioned sigma, bounded mean (the reference defaults, cvpo_agent.py:107-108)"
        self.actor_old = deepcopy(self.actor)
        self.actor_old.eval()
        self.actor_optim, self.critics_optim = actor_optim, critic_optim
        self.critics_old = deepcopy(self.critics)
        self.critics_old.eval()
        self.max_episode_steps = max_episode_steps
        self.tau, self._n_step = tau, n_step
        self._sample_act_num, self._mstep_iter_num = sample_act_num, mstep_iter_num
        self.update_cost_limit(cost_limit, push=False)
        double = hasattr(self.critics[0], "preprocess2")
        from fsrl_amd.utils.net import mlp_geometry
        obs_dim, hidden_sizes = mlp_geometry(actor.preprocess)
        act_dim = actor.mu.model[0].weight.shape[0]
        dev = device if isinstance(device, int) else (int(str(device).split(":")[-1]) if ":" in str(device) else 0)
        self.engine = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=int(obs_dim), act_dim=int(act_dim),
                                          hidden_sizes=hidden_sizes, n_critics=2, env_num=int(env_num),
                                          buffer_size=int(buffer_size), gamma=gamma, max_action=float(actor._max),
                                          target_kl=None), device=dev)
        self.engine.cvpo_init(self.qc_thres[0], actor_lr=actor_optim.param_groups[0]["lr"],
                              critic_lr=critic_optim.param_groups[0]["lr"], tau=tau, n_step=n_step, double_critic=double,
                              sample_act_num=sample_act_num, estep_iter_num=estep_iter_num, mstep_iter_num=mstep_iter_num,
                              estep_kl=estep_kl, estep_dual_max=estep_dual_max, estep_dual_lr=estep_dual_lr,
                              mstep_kl_mu=mstep_kl_mu, mstep_kl_std=mstep_kl_std, mstep_dual_max=mstep_dual_max,
                              mstep_dual_lr=mstep_dual_lr)
        self.engine.sac_set_params(SACLagrangian._flat([self.actor]), SACLagrangian._flat(list(self.critics)), 0.0)
        self.engine.cvpo_pre_update()
        # reference_rng: buffer.sample through numpy's and every Normal.sample through torch's global RNG, in the
        # order the reference consumes them (cvpo.py:208, 331, 334, 382); False: Philox on the device, async updates
        self._dirty, self._reference_rng, self._seed, self._pending = False, reference_rng, int(seed), 0
        self._rest_dirty = False

    def update_cost_limit(self, cost_limit, push: bool = True) -> None:
        """cvpo.py:165-176"""
        self.cost_limit = [cost_limit] * (self.critics_num - 1) if np.isscalar(cost_limit) else cost_limit
        g, T = self._gamma, self.max_episode_steps
        self.qc_thres = [c * (1 - g**T) / (1 - g) / T for c in self.cost_limit]
        if push:
            self.engine.cvpo_set_thres(self.qc_thres[0])

    def train(self, mode: bool = True):
        self.training = mode
        self.actor.train(mode)
        self.critics.train(mode)
        return self

    # ------------------------------------------------------------------ parameter plumbing
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
        if getattr(self, "_pending", 0):
            self._drain()
        if getattr(self, "_dirty", False) or getattr(self, "_rest_dirty", False):
            self._pull_params(everything=True)
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True):
        out = nn.Module.load_state_dict(self, state_dict, strict=strict)
        if getattr(self, "engine", None) is not None:
            for which, mods in ((0, [self.actor]), (3, [self.actor_old]), (1, list(self.critics)), (2, list(self.critics_old))):
                self.engine.sac_put_params(which, SACLagrangian._flat(mods))
            self._dirty = self._rest_dirty = False
        return out

    def get_extra_state(self):
        """cvpo.py:432-439: the reference saves nothing here (its duals are rebuilt with the policy)."""

    def set_extra_state(self, state):
        pass

    # ------------------------------------------------------------------ acting (host mirror)
    def forward(self, batch: Batch, state=None, model: str = "actor", input: str = "obs", **kwargs: Any) -> Batch:
        if self._dirty or (model != "actor" and self._rest_dirty):
            self._pull_params(everything=model != "actor")
        logits, hidden = getattr(self, model)(batch[input], state=state)
        dist = self.dist_fn(*logits) if isinstance(logits, tuple) else self.dist_fn(logits)
        act = logits[0] if (self._deterministic_eval and not self.training) else dist.sample()
        return Batch(logits=logits, act=act, state=hidden, dist=dist)

    # ------------------------------------------------------------------ update
    def pre_update_fn(self, **kwarg: Any) -> None:
        self.engine.cvpo_pre_update()

    def post_update_fn(self, **kwarg: Any) -> None:
        self._drain()
        self.engine.cvpo_post_update()
        self._dirty = self._rest_dirty = True      # actor mirror AND critics / targets are behind the device now

    def _log_rows(self, rows) -> None:
        table = getattr(self.logger, "store_rows", None)         # fsrl_amd loggers take the drained rows at once
        if table is not None and len(rows):
            table([k if tab is None else tab + "/" + k for tab, k in CVPO_KEYS], rows)
            return
        for st in rows:
            v = [float(x) for x in st]
            self.logger.store(tab="loss", estep_loss=v[0])
            self.logger.store(**{"estep/dual0": v[1]})
            self.logger.store(**{"estep/dual1": v[2]})
            self.logger.store(tab="mstep", **{k: v[3 + j] for j, (_, k) in enumerate(CVPO_KEYS[3:11])})
            self.logger.store(**{k: v[11 + j] for j, (_, k) in enumerate(CVPO_KEYS[11:])})

    def _drain(self) -> None:
        if self._pending:
            self._log_rows(self.engine.sac_drain())
            self._pending = 0

    def process_fn(self, batch=None, buffer=None, indices=None, sample_size: int = 0):
        """cvpo.py:204-222 (`compute_nstep_returns` over `_target_q`): the sample's indices and ONE actor sample at s_{t+n}; the
        n-step targets are formed on the device inside learn's critic launch.  -> ReplayDeviceBatch.  indices=None: the library
        draws `sample_size` rows itself (Philox, in learn); `batch` is ignored (the rows live in the HBM store)."""
        assert getattr(buffer, "engine", None) is self.engine, \
            "CVPO.process_fn needs the HipVectorReplayBuffer bound to this policy's engine"
        self.updating = True
        if indices is None:
            return ReplayDeviceBatch(self.engine, int(sample_size))
        B, Da = len(indices), self.engine.cfg.act_dim
        eps_t = torch.randn(B, Da).numpy()              # _target_q: forward(actor, obs_next).act          (cvpo.py:208)
        return ReplayDeviceBatch(self.engine, B, indices, eps_t)

    def learn(self, batch, **kwargs: Any):
        """cvpo.py:319-420 on the device = `fsrl_cvpo_update` (critics' step, E-step dual, M-step iterations); `batch` is what
        process_fn returned."""
        assert isinstance(batch, ReplayDeviceBatch) and batch.engine is self.engine and len(batch) >= 1, \
            "learn() takes the ReplayDeviceBatch process_fn() returned (the sampled rows live in HBM)"
        B, Da, K = len(batch), self.engine.cfg.act_dim, self._sample_act_num
        if batch.indices is not None:
            torch.randn(B, Da)                          # policy_loss: forward(actor_old).act, unused      (cvpo.py:331)
            eps_k = torch.randn(K, B, Da).numpy()       # old_dist.sample((K, ))                            (cvpo.py:334)
            for _ in range(self._mstep_iter_num):
                torch.randn(B, Da)                      # M-step: forward(actor).act, unused                (cvpo.py:382)
            st = self.engine.cvpo_update(B, indices=batch.indices, eps_target=batch.eps_target, eps_particles=eps_k)
            self._log_rows(st[None])
        else:
            self.engine.cvpo_update(B, seed=self._seed + 1 if self.gradient_steps == 0 else 0, sync=False)
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
