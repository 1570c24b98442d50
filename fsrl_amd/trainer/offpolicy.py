"""Off-policy trainer: round(update_per_step * n/st) device updates per collect
(fsrl/trainer/offpolicy.py:93-106)."""
from typing import Any, Dict

from fsrl_amd.trainer.base_trainer import BaseTrainer


class OffpolicyTrainer(BaseTrainer):
    def __init__(self, policy, train_collector, test_collector=None, max_epoch: int = 100,
                 batch_size: int = 512, cost_limit: float = float("inf"), step_per_epoch=None,
                 update_per_step: float = 1, save_model_interval: int = 1, episode_per_test=None,
                 episode_per_collect: int = 1, stop_fn=None, resume_from_log: bool = False, logger=None,
                 verbose: bool = True, show_progress: bool = True):
        super().__init__(learning_type="offpolicy", policy=policy, train_collector=train_collector,
                         test_collector=test_collector, max_epoch=max_epoch, batch_size=batch_size,
                         cost_limit=cost_limit, step_per_epoch=step_per_epoch, update_per_step=update_per_step,
                         save_model_interval=save_model_interval, episode_per_test=episode_per_test,
                         episode_per_collect=episode_per_collect, stop_fn=stop_fn,
                         resume_from_log=resume_from_log, logger=logger, verbose=verbose,
                         show_progress=show_progress)

    def policy_update_fn(self, stats_train: Dict[str, Any]) -> None:
        self.policy.pre_update_fn(stats_train=stats_train, batch_size=self.batch_size,
                                  buffer=self.train_collector.buffer, update_per_step=self.update_per_step)
        for _ in range(round(self.update_per_step * stats_train["n/st"])):
            self.policy.update(self.batch_size, self.train_collector.buffer)
        self.policy.post_update_fn(stats_train=stats_train)
