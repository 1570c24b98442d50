"""Epoch iterator: collect -> policy_update_fn -> test -> checkpoint/log, with the timing info
of fsrl/trainer/base_trainer.py:181-356 (`train_speed` = env-steps/s, `train_model_time`, ...).
The caller of the hot path; plain Python."""
import time
from abc import ABC, abstractmethod
from collections import deque
from typing import Any, Callable, Dict, Optional, Tuple, Union

import numpy as np

from fsrl_amd.utils.logger import BaseLogger, DummyLogger


class BaseTrainer(ABC):
    def __init__(self, learning_type: str, policy, train_collector, test_collector=None,
                 max_epoch: int = 100, batch_size: int = 512, cost_limit: float = np.inf,
                 step_per_epoch: Optional[int] = None, repeat_per_collect: Optional[int] = None,
                 update_per_step: Union[int, float] = 1, save_model_interval: int = 1,
                 episode_per_test: Optional[int] = None, episode_per_collect: int = 1,
                 stop_fn: Optional[Callable[[float, float], bool]] = None, resume_from_log: bool = False,
                 logger: BaseLogger = None, verbose: bool = True, show_progress: bool = True):
        self.learning_type = learning_type
        self.policy = policy
        self.train_collector, self.test_collector = train_collector, test_collector
        self.logger = logger if logger is not None else DummyLogger()
        self.cost_limit = cost_limit
        self.start_time = time.time()
        self.best_perf_rew, self.best_perf_cost = -np.inf, np.inf
        self.start_epoch = 0
        self.env_step, self.cum_cost, self.cum_episode = 0, 0, 0
        self.max_epoch, self.step_per_epoch = max_epoch, step_per_epoch
        self.episode_per_collect, self.episode_per_test = episode_per_collect, episode_per_test
        self.update_per_step, self.save_model_interval = update_per_step, save_model_interval
        self.repeat_per_collect, self.batch_size = repeat_per_collect, batch_size
        self.stop_fn = stop_fn
        self.verbose, self.show_progress, self.resume_from_log = verbose, show_progress, resume_from_log
        self.epoch = self.start_epoch
        self.stop_fn_flag = False
        self.update_time = 0.0   # wall time spent inside policy_update_fn (device path)

    def reset(self) -> None:
        self.env_step = 0
        self.start_time = time.time()
        self.train_collector.reset_stat()
        if self.test_collector is not None:
            assert self.episode_per_test is not None
            self.test_collector.reset_stat()
        self.epoch = self.start_epoch
        self.stop_fn_flag = False

    def __iter__(self):
        self.reset()
        return self

    def __next__(self) -> Tuple[int, Dict, Dict]:
        self.epoch += 1
        if self.epoch > self.max_epoch or self.stop_fn_flag:
            raise StopIteration
        self.policy.train()
        steps_this_epoch = 0
        while steps_this_epoch < self.step_per_epoch:
            stats_train = self.train_step()
            steps_this_epoch += int(stats_train["n/st"])
            t0 = time.time()
            self.policy_update_fn(stats_train)
            self.update_time += time.time() - t0
            self.logger.write_without_reset(self.env_step)
        if self.test_collector is not None:
            self.test_step()
        update_info = self.gather_update_info()
        self.logger.store(tab="update", **update_info)
        if self.epoch % self.save_model_interval == 0:
            self.logger.save_checkpoint()
        if self.perf_is_better(test=True):
            self.logger.save_checkpoint(suffix="best")
        if self.stop_fn and self.stop_fn(self.best_perf_rew, self.best_perf_cost):
            self.stop_fn_flag = True
            self.logger.print("Early stop due to the stop_fn met.", "red")
        epoch_stats = self.logger.stats_mean
        self.logger.write(self.env_step, display=self.verbose)
        update_info.update({"best_reward": self.best_perf_rew, "best_cost": self.best_perf_cost})
        return self.epoch, epoch_stats, update_info

    def perf_is_better(self, test: bool = True) -> bool:
        mode = "test" if test and self.test_collector is not None else "train"
        rew, cost = self.logger.get_mean(mode + "/reward"), self.logger.get_mean(mode + "/cost")
        if self.best_perf_cost > self.cost_limit:
            better = cost <= self.cost_limit or rew > self.best_perf_rew
        else:
            better = cost <= self.cost_limit and rew > self.best_perf_rew
        if better:
            self.best_perf_cost, self.best_perf_rew = cost, rew
        return better

    def test_step(self) -> Dict[str, Any]:
        self.test_collector.reset_env()
        self.test_collector.reset_buffer()
        self.policy.eval()
        stats_test = self.test_collector.collect(n_episode=self.episode_per_test)
        self.logger.store(**{"test/reward": stats_test["rew"], "test/cost": stats_test["cost"],
                             "test/length": int(stats_test["len"])})
        return stats_test

    def train_step(self) -> Dict[str, Any]:
        stats_train = self.train_collector.collect(self.episode_per_collect)
        self.env_step += int(stats_train["n/st"])
        self.cum_cost += stats_train["total_cost"]
        self.cum_episode += int(stats_train["n/ep"])
        self.logger.store(**{"update/episode": self.cum_episode, "update/cum_cost": self.cum_cost,
                             "train/reward": stats_train["rew"], "train/cost": stats_train["cost"],
                             "train/length": int(stats_train["len"])})
        return stats_train

    @abstractmethod
    def policy_update_fn(self, result: Dict[str, Any]) -> None:
        pass

    def run(self) -> Dict[str, Union[float, str]]:
        deque(self, maxlen=0)
        return self.gather_update_info()

    def gather_update_info(self) -> Dict[str, Any]:
        duration = max(0, time.time() - self.start_time)
        model_time = max(0, duration - self.train_collector.collect_time)
        result = {"duration": duration}
        if self.test_collector is not None and self.test_collector.collect_time > 0:
            collect_test = self.test_collector.collect_time
            model_time = max(0, model_time - collect_test)
            result.update({"test_time": collect_test,
                           "test_speed": self.test_collector.collect_step / collect_test})
            train_speed = self.train_collector.collect_step / max(duration - collect_test, 1e-9)
        else:
            train_speed = self.train_collector.collect_step / max(duration, 1e-9)
        result.update({"train_collector_time": self.train_collector.collect_time,
                       "train_model_time": model_time, "train_speed": train_speed,
                       "policy_update_time": self.update_time,
                       "remaining_epoch": self.max_epoch - self.epoch})
        return result
