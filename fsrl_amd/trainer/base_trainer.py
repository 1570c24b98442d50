"""The epoch loop that calls the hot path: collect -> policy_update_fn -> (test) -> checkpoint / log.

Interface of fsrl/trainer/base_trainer.py:181-356 as its callers use it -- iterate the trainer for
`(epoch, epoch_stats, info)` triples, or `run()` -- with the same logger keys (`train/*`, `test/*`, `update/*`) and the
same `info` fields (`train_speed` = env-steps/s, `train_model_time`, ...).  Organised differently: the best-so-far
rule lives in `_Best`, the wall-clock bookkeeping in `_Clock`, and an epoch is three short phases.  Plain Python."""
import time
from abc import ABC, abstractmethod
from typing import Any, Callable, Dict, Optional, Tuple, Union

import numpy as np

from fsrl_amd import parallel
from fsrl_amd.utils.logger import BaseLogger, DummyLogger


class _Best:
    """Best (reward, cost) seen so far under a cost limit: while the incumbent violates the limit, anything feasible
    or better-rewarded replaces it; once it is feasible, only feasible AND better-rewarded results do."""

    def __init__(self, limit: float) -> None:
        self.limit, self.reward, self.cost = limit, -np.inf, np.inf

    def offer(self, reward: float, cost: float) -> bool:
        feasible, richer = cost <= self.limit, reward > self.reward
        accept = (feasible or richer) if self.cost > self.limit else (feasible and richer)
        if accept:
            self.reward, self.cost = reward, cost
        return accept


class _Clock:
    """Wall-clock accounts of a run: total, and the part spent inside policy_update_fn."""

    def __init__(self) -> None:
        self.t0, self.in_update = time.time(), 0.0

    def restart(self) -> None:
        self.t0 = time.time()

    def elapsed(self) -> float:
        return max(0.0, time.time() - self.t0)


class BaseTrainer(ABC):
    def __init__(self, learning_type: str, policy, train_collector, test_collector=None,
                 max_epoch: int = 100, batch_size: int = 512, cost_limit: float = np.inf,
                 step_per_epoch: Optional[int] = None, repeat_per_collect: Optional[int] = None,
                 update_per_step: Union[int, float] = 1, save_model_interval: int = 1,
                 episode_per_test: Optional[int] = None, episode_per_collect: int = 1,
                 stop_fn: Optional[Callable[[float, float], bool]] = None, resume_from_log: bool = False,
                 logger: BaseLogger = None, verbose: bool = True, show_progress: bool = True):
        self.learning_type, self.policy = learning_type, policy
        self.train_collector, self.test_collector = train_collector, test_collector
        self.logger = DummyLogger() if logger is None else logger
        # schedule
        self.max_epoch, self.step_per_epoch, self.save_model_interval = max_epoch, step_per_epoch, save_model_interval
        self.episode_per_collect, self.episode_per_test = episode_per_collect, episode_per_test
        self.batch_size, self.repeat_per_collect, self.update_per_step = batch_size, repeat_per_collect, update_per_step
        self.stop_fn, self.cost_limit = stop_fn, cost_limit
        self.verbose, self.show_progress, self.resume_from_log = verbose, show_progress, resume_from_log
        # run state
        self.start_epoch = 0
        self.epoch, self.stop_fn_flag = self.start_epoch, False
        self.env_step, self.cum_cost, self.cum_episode = 0, 0, 0
        self._best, self._clock = _Best(cost_limit), _Clock()
        self._acc: Dict[str, float] = {}     # this rank's sums of the current epoch (parallel.EPOCH_KEYS)
        self.job_stats: Dict[str, float] = {}   # last epoch's job-level figures (all ranks; == this rank's alone)
        self.rank, self.world = parallel.rank_world()

    # attributes the reference exposes under these names
    best_perf_rew = property(lambda self: self._best.reward)
    best_perf_cost = property(lambda self: self._best.cost)
    update_time = property(lambda self: self._clock.in_update)
    start_time = property(lambda self: self._clock.t0)

    # ------------------------------------------------------------------ iteration protocol
    def reset(self) -> None:
        self.env_step, self.epoch, self.stop_fn_flag = 0, self.start_epoch, False
        self._clock.restart()
        self.train_collector.reset_stat()
        if self.test_collector is not None:
            assert self.episode_per_test is not None
            self.test_collector.reset_stat()

    def __iter__(self):
        self.reset()
        return self

    def __next__(self) -> Tuple[int, Dict, Dict]:
        if self.stop_fn_flag or self.epoch >= self.max_epoch:
            raise StopIteration
        self.epoch += 1
        self._acc = {}
        t_epoch = time.time()
        self._train_phase()
        if self.test_collector is not None:
            self.test_step()
        self._acc["duration"] = time.time() - t_epoch
        info = self._close_epoch()
        return self.epoch, self._epoch_stats, info

    def run(self) -> Dict[str, Union[float, str]]:
        for _ in self:
            pass
        return self.gather_update_info()

    # ------------------------------------------------------------------ the three phases of an epoch
    def _train_phase(self) -> None:
        """collect / update cycles until the epoch's step budget is spent"""
        self.policy.train()
        budget = self.step_per_epoch
        while budget > 0:
            tic = time.time()
            collected = self.train_step()
            budget -= int(collected["n/st"])
            self._bump("collect_time", time.time() - tic)
            tic, g0 = time.time(), int(getattr(self.policy, "gradient_steps", 0))
            self.policy_update_fn(collected)
            self._clock.in_update += time.time() - tic
            self._bump("update_time", time.time() - tic)
            self._bump("n_updates", 1 if self.learning_type == "onpolicy" else
                       round(self.update_per_step * collected["n/st"]))
            self._bump("n_grad_steps", int(getattr(self.policy, "gradient_steps", 0)) - g0)
            self.logger.write_without_reset(self.env_step)

    def test_step(self) -> Dict[str, Any]:
        col = self.test_collector
        col.reset_env()
        col.reset_buffer()
        self.policy.eval()
        res = col.collect(n_episode=self.episode_per_test)
        self.logger.store(**{"test/reward": res["rew"], "test/cost": res["cost"], "test/length": int(res["len"])})
        self._bump("test_n_ep", res["n/ep"]); self._bump("test_sum_rew", res["rew"] * res["n/ep"])
        self._bump("test_sum_cost", res["total_cost"])
        return res

    def _close_epoch(self) -> Dict[str, Any]:
        """log the timing info, checkpoint, track the best result, evaluate the stop rule, flush the logger"""
        info = self.gather_update_info()
        self.logger.store(tab="update", **info)
        if self.epoch % self.save_model_interval == 0:
            self.logger.save_checkpoint()
        if self.perf_is_better(test=True):
            self.logger.save_checkpoint(suffix="best")
        stop_here = self.stop_fn is not None and self.stop_fn(self._best.reward, self._best.cost)
        # ---- the one exchange step of the multi-GPU layout: the epoch vector, all-reduced over the ranks.
        # The loss means come from this rank's logger (per-step rows of the epoch), weighted by its optimiser steps.
        gsteps = self._acc.get("n_grad_steps", 0.0)
        self._acc["sum_loss_total"] = self.logger.get_mean("loss/total") * gsteps
        self._acc["sum_kl"] = self.logger.get_mean("loss/kl") * gsteps
        self._acc["stop"] = 1.0 if stop_here else 0.0
        eng = getattr(self.policy, "engine", None)
        lib_world = eng.comm_info()[1] if eng is not None and hasattr(eng, "comm_info") else 1
        self.job_stats = parallel.reduce_epoch(self._acc, eng if lib_world > 1 else None)
        if lib_world > 1 and self.world == 1:          # ranks joined through the library only (no torch process group)
            self.rank, self.world = eng.comm_info()
        if self.world > 1:
            # independent agents, one collective per epoch: every rank must enter it the same number of times, so the
            # job stops when EVERY rank's stop rule has fired (a rank that is done early keeps training until then)
            stop_here = self.job_stats["job/all_stop"] > 0
            if self.rank == 0:
                self.logger.store(**self.job_stats)
        if stop_here:
            self.stop_fn_flag = True
            self.logger.print("Early stop due to the stop_fn met.", "red")
        self._epoch_stats = self.logger.stats_mean
        self.logger.write(self.env_step, display=self.verbose)
        info.update(best_reward=self._best.reward, best_cost=self._best.cost)
        if self.world > 1:
            info.update(self.job_stats)
        return info

    def _bump(self, key: str, value: float) -> None:
        self._acc[key] = self._acc.get(key, 0.0) + float(value)

    # ------------------------------------------------------------------ pieces subclasses and callers use
    def train_step(self) -> Dict[str, Any]:
        res = self.train_collector.collect(self.episode_per_collect)
        self.env_step += int(res["n/st"])
        self.cum_episode += int(res["n/ep"])
        self.cum_cost += res["total_cost"]
        self.logger.store(**{"update/episode": self.cum_episode, "update/cum_cost": self.cum_cost,
                             "train/reward": res["rew"], "train/cost": res["cost"], "train/length": int(res["len"])})
        self._bump("n_st", res["n/st"]); self._bump("n_ep", res["n/ep"])
        self._bump("sum_rew", res["rew"] * res["n/ep"]); self._bump("sum_cost", res["total_cost"])
        self._bump("sum_len", res["len"] * res["n/ep"])
        return res

    def perf_is_better(self, test: bool = True) -> bool:
        tab = "test" if (test and self.test_collector is not None) else "train"
        return self._best.offer(self.logger.get_mean(tab + "/reward"), self.logger.get_mean(tab + "/cost"))

    @abstractmethod
    def policy_update_fn(self, result: Dict[str, Any]) -> None:
        """pre_update_fn -> update(s) -> post_update_fn of the concrete learning type"""

    def gather_update_info(self) -> Dict[str, Any]:
        total = self._clock.elapsed()
        train_col = self.train_collector
        test_s = self.test_collector.collect_time if self.test_collector is not None else 0.0
        info: Dict[str, Any] = {"duration": total}
        if test_s > 0:
            info["test_time"] = test_s
            info["test_speed"] = self.test_collector.collect_step / test_s
        info["train_collector_time"] = train_col.collect_time
        info["train_model_time"] = max(0.0, total - train_col.collect_time - test_s)
        info["train_speed"] = train_col.collect_step / max(total - test_s, 1e-9)
        info["policy_update_time"] = self._clock.in_update
        info["remaining_epoch"] = self.max_epoch - self.epoch
        return info
