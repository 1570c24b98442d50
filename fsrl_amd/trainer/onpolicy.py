"""On-policy trainer: one device update per collect (fsrl/trainer/onpolicy.py:92-109)."""
from typing import Any, Dict

from fsrl_amd.trainer.base_trainer import BaseTrainer


class OnpolicyTrainer(BaseTrainer):
    def __init__(self, policy, train_collector, test_collector=None, max_epoch: int = 100,
                 batch_size: int = 512, cost_limit: float = float("inf"), step_per_epoch=None,
                 repeat_per_collect=None, save_model_interval: int = 1, episode_per_test=None,
                 episode_per_collect: int = 1, stop_fn=None, resume_from_log: bool = False,
                 logger=None, verbose: bool = True, show_progress: bool = True):
        super().__init__(learning_type="onpolicy", policy=policy, train_collector=train_collector,
                         test_collector=test_collector, max_epoch=max_epoch, batch_size=batch_size,
                         cost_limit=cost_limit, step_per_epoch=step_per_epoch,
                         repeat_per_collect=repeat_per_collect, save_model_interval=save_model_interval,
                         episode_per_test=episode_per_test, episode_per_collect=episode_per_collect,
                         stop_fn=stop_fn, resume_from_log=resume_from_log, logger=logger,
                         verbose=verbose, show_progress=show_progress)

    def policy_update_fn(self, stats_train: Dict[str, Any]) -> None:
        self.policy.pre_update_fn(stats_train=stats_train, batch_size=self.batch_size,
                                  buffer=self.train_collector.buffer)
        # first argument 0: consume every stored transition (on-policy)
        self.policy.update(0, self.train_collector.buffer, batch_size=self.batch_size,
                           repeat=self.repeat_per_collect)
        self.policy.post_update_fn(stats_train=stats_train)
        self.train_collector.reset_buffer(keep_statistics=True)
