"""PPOLagrangian over the HIP engine.  Same constructor arguments and logger keys as
fsrl/policy/ppo_lag.py:16-257; `update()` = process_fn + learn on the MI355X through
`fsrl_ppo_begin / fsrl_ppo_pass / fsrl_ppo_end` (include/fsrl_hip.h)."""
from typing import Any, List, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn

from fsrl_amd.policy.base_policy import DeviceBatch
from fsrl_amd.policy.lagrangian_base import LagrangianPolicy

PPO_STAT_KEYS = ("loss/rescaling", "loss/lagrangian", "loss/actor_safety", "loss/actor_rew",
                 "loss/actor_total", "loss/kl", "loss/vf0", "loss/vf1", "loss/vf_total", "loss/total",
                 "loss/entropy")


def _chunk_sizes(n: int, size: int) -> List[int]:
    """Row counts of tianshou-0.5 Batch.split(size, merge_last=True)."""
    out, i = [], 0
    merge_last = n % size > 0
    while i < n:
        if merge_last and i + 2 * size >= n:
            out.append(n - i)
            break
        out.append(min(size, n - i))
        i += size
    return out


class PPOLagrangian(LagrangianPolicy):
    def __init__(self, actor: nn.Module, critics: Union[nn.Module, List[nn.Module]],
                 optim: torch.optim.Optimizer, dist_fn, logger=None,
                 # PPO specific arguments
                 target_kl: float = 0.02, vf_coef: float = 0.25, max_grad_norm: Optional[float] = None,
                 gae_lambda: float = 0.95, eps_clip: float = 0.2, dual_clip: Optional[float] = None,
                 value_clip: bool = False, advantage_normalization: bool = True,
                 recompute_advantage: bool = False,
                 # Lagrangian specific arguments
                 use_lagrangian: bool = True, lagrangian_pid: Tuple = (0.05, 0.0005, 0.1),
                 cost_limit: Union[List, float] = np.inf, rescaling: bool = True,
                 # Base policy common arguments
                 gamma: float = 0.99, max_batchsize: int = 99999, reward_normalization: bool = False,
                 deterministic_eval: bool = True, action_scaling: bool = True,
                 action_bound_method: str = "clip", observation_space=None, action_space=None,
                 lr_scheduler=None,
                 # engine placement (not in the reference: which GPU, how many env sub-buffers)
                 device: Union[int, str] = 0, env_num: int = 1, buffer_size: int = 100000,
                 reference_rng: bool = False) -> None:
        super().__init__(actor, critics, dist_fn, logger, use_lagrangian, lagrangian_pid, cost_limit,
                         rescaling, gamma, max_batchsize, reward_normalization, deterministic_eval,
                         action_scaling, action_bound_method, observation_space, action_space, lr_scheduler)
        assert dual_clip is None or dual_clip > 1.0, "Dual-clip PPO parameter should greater than 1.0."
        assert reward_normalization or not value_clip, "value clip is available only when `reward_normalization` is True"
        self._value_clip = bool(value_clip)
        assert 0.0 <= gae_lambda <= 1.0, "GAE lambda should be in [0, 1]."
        self.optim = optim
        self._lambda, self._weight_vf, self._grad_norm = gae_lambda, vf_coef, max_grad_norm
        self._target_kl, self._eps_clip, self._dual_clip = target_kl, eps_clip, dual_clip
        self._norm_adv = advantage_normalization
        # reference_rng=True: also consume torch's global RNG exactly where the reference does inside update() --
        # its forward() draws dist.sample() in training mode even when only the distribution is used
        # (process_fn: ppo_lag.py:146-148; learn: :225 via policy_loss :175) -- so that a whole collect/update
        # loop stays on the reference's random streams (tests/test_gpu_loop.py).  Costs ~N*Da normals per update.
        self._reference_rng = reference_rng
        self._make_engine(device, env_num, buffer_size, optim, gae_lambda=gae_lambda, eps_clip=eps_clip,
                          dual_clip=dual_clip, vf_coef=vf_coef, max_grad_norm=max_grad_norm,
                          target_kl=target_kl, norm_adv=advantage_normalization,
                          use_lagrangian=use_lagrangian, recompute_adv=bool(recompute_advantage),
                          value_clip=bool(value_clip))

    def _burn_samples(self, sizes) -> None:
        da = self.engine.cfg.act_dim
        for m in sizes:                                          # Independent(Normal).sample() of an [m, Da] batch
            torch.normal(torch.zeros(m, da), torch.ones(m, da))

    def process_fn(self, batch=None, buffer=None, indices=None, batch_size: int = 256):
        """ppo_lag.py:134-150 on the device = `fsrl_ppo_begin`: buffer.sample(0), V(obs) / V(obs_next) of every critic, the float64
        GAE scan, logp_old.  -> DeviceBatch (the processed batch stays in HBM).  `batch` / `indices` are accepted for signature
        parity and ignored: the on-policy batch is always the whole store.  batch_size: the minibatch size `learn` will use (the
        library plans its working set at begin; learn() re-begins if it is given another one)."""
        assert getattr(buffer, "engine", None) is self.engine, \
            "PPOLagrangian.process_fn needs the HipVectorReplayBuffer bound to this policy's engine"
        self.updating = True
        lags, rescaling = self.lagrangians_and_rescaling() if self.use_lagrangian else ([], 1.0)
        self._begin_args = (lags, rescaling)
        n = self.engine.ppo_begin(lags, rescaling, batch_size)      # buffer.sample(0) + process_fn
        try:
            if self._reference_rng and (self.training or not self._deterministic_eval):
                self._burn_samples(_chunk_sizes(n, self._max_batchsize))   # process_fn's forward over chunks of max_batchsize
        except BaseException:
            self.engine.ppo_abort(); self.updating = False
            raise
        self._pending = DeviceBatch(self.engine, n, batch_size)
        return self._pending

    def learn(self, batch, batch_size: int = 256, repeat: int = 4, **kwargs: Any):
        """ppo_lag.py:214-257 on the device = `fsrl_ppo_pass` x repeat + `fsrl_ppo_end`; `batch` is what process_fn returned."""
        assert isinstance(batch, DeviceBatch) and batch is getattr(self, "_pending", None), \
            "learn() takes the DeviceBatch the last process_fn() returned (the processed batch lives in HBM)"
        self._pending = None
        eng, n = self.engine, batch.n
        try:                                                     # begin ... end is a state machine in the library:
            if batch.batch_size != batch_size:                   # another minibatch size than planned: begin again (no RNG is consumed)
                eng.ppo_abort()
                n = eng.ppo_begin(*self._begin_args, batch_size)
            burn = self._reference_rng and (self.training or not self._deterministic_eval)
            stopped_at = -1
            perm = np.random.permutation(n) if n > 0 else None       # Batch.split(shuffle=True) of the first pass
            for step in range(repeat):                               # ppo_lag.py:217
                if burn:
                    self._burn_samples(_chunk_sizes(n, batch_size))  # one forward per minibatch
                eng.ppo_pass(perm, wait=False)
                # the next pass's permutation is drawn while the device runs this one; if this pass turns out to be the last
                # (KL early stop) numpy's stream is rolled back, so it stays where the reference's would be
                rng_state = np.random.get_state() if (n > 0 and step + 1 < repeat) else None
                if rng_state is not None:
                    perm = np.random.permutation(n)
                if eng.ppo_pass_result():
                    if rng_state is not None:
                        np.random.set_state(rng_state)
                    stopped_at = step
                    self.logger.print("Early stop at step %d due to reaching max kl." % step)
                    break
            steps_per_pass = max(1, -(-n // max(batch_size, 1)))
            stats = eng.ppo_end_stats(steps_per_pass * max(repeat, 1))
        except BaseException:                                    # leave it (and `updating`) clean on any error / interrupt
            eng.ppo_abort()
            self.updating = False
            raise
        # keys the reference would not log in this configuration (lagrangian_base.py:158-165, ppo_lag.py:169-170)
        drop = set()
        if not self.use_lagrangian or self.critics_num < 2:
            drop |= {"loss/lagrangian", "loss/actor_safety"}
        if self.critics_num < 2:
            drop.add("loss/vf1")
        cols = [j for j, k in enumerate(PPO_STAT_KEYS) if k not in drop]
        keys = [PPO_STAT_KEYS[j] for j in cols]
        table = getattr(self.logger, "store_rows", None)         # fsrl_amd loggers take the per-step table at once
        if table is not None:
            table(keys, stats[:, cols] if drop else stats)
        for row in (stats if table is None else ()):             # any other logger: the reference's per-step calls
            d = {k: float(row[j]) for j, k in zip(cols, keys)}
            total, entropy = d.pop("loss/total"), d.pop("loss/entropy")
            self.logger.store(**d)
            self.logger.store(total=total, entropy=entropy, tab="loss")
        self.gradient_steps += len(stats)
        self.logger.store(gradient_steps=self.gradient_steps, tab="update")
        self._mark_stale()                                       # host mirror refreshed on demand
        return {"gradient_steps": len(stats), "early_stop_pass": stopped_at}

    def update(self, sample_size: int, buffer, batch_size: int = 256, repeat: int = 4, **kwargs: Any):
        """base_policy.py:332-355: sample(0) -> process_fn -> learn -> lr scheduler"""
        if buffer is None:
            return {}
        assert sample_size == 0, "on-policy update consumes the whole buffer (sample_size=0)"
        batch = self.process_fn(None, buffer, None, batch_size=batch_size)
        result = self.learn(batch, batch_size=batch_size, repeat=repeat)
        self._step_lr_scheduler()
        self.updating = False
        return result
