"""Agent base classes: `evaluate`, `state_dict` and the on-policy `learn` loop with the keyword
arguments of fsrl/agent/base_agent.py:53-93,225-324.  The replay buffer created here is the
HIP-resident store proxy bound to the policy's engine."""
from abc import ABC, abstractmethod
from typing import Optional, Tuple

from fsrl_amd.data import FastCollector, HipVectorReplayBuffer
from fsrl_amd.trainer import OnpolicyTrainer
from fsrl_amd.utils.logger import BaseLogger, DummyLogger


class BaseAgent(ABC):
    name = "BaseAgent"

    @abstractmethod
    def __init__(self, *args, **kwargs) -> None:
        self.policy = None
        self.logger: BaseLogger = DummyLogger()
        self.cost_limit = 0

    @abstractmethod
    def learn(self, *args, **kwargs):
        raise NotImplementedError

    def evaluate(self, test_envs, state_dict: Optional[dict] = None, eval_episodes: int = 10,
                 render: bool = False, train_mode: bool = False) -> Tuple[float, float, float]:
        if state_dict is not None:
            self.policy.load_state_dict(state_dict)
        self.policy.train() if train_mode else self.policy.eval()
        result = FastCollector(self.policy, test_envs).collect(n_episode=eval_episodes, render=render)
        return result["rew"], result["len"], result["cost"]

    @property
    def state_dict(self):
        return self.policy.state_dict()


class OnpolicyAgent(BaseAgent):
    name = "OnpolicyAgent"

    def __init__(self) -> None:
        pass

    def learn(self, train_envs, test_envs=None, epoch: int = 300, episode_per_collect: int = 20,
              step_per_epoch: int = 10000, repeat_per_collect: int = 4, buffer_size: Optional[int] = None,
              testing_num: int = 2, batch_size: int = 512, reward_threshold: float = 450,
              save_interval: int = 4, resume: bool = False, save_ckpt: bool = True,
              verbose: bool = True, show_progress: bool = True, device_actor: bool = False):
        assert self.policy is not None, "The policy is not initialized"
        self.policy.train()
        eng = self.policy.engine
        from fsrl_amd.env.venv import as_vector_env
        train_envs = as_vector_env(train_envs)                 # a single env: one sub-buffer (base_agent.py:160-163)
        assert eng.cfg.env_num >= len(train_envs), \
            f"agent built for {eng.cfg.env_num} env sub-buffers, got {len(train_envs)} envs (pass training_num)"
        # VectorReplayBuffer(buffer_size, len(train_envs)) of the reference (base_agent.py:279): the store is re-cut to
        # that geometry.  None = the size the agent was built with (its `buffer_size`, default 100 000 like the
        # reference's learn()); an explicit size must fit that allocation (AssertionError otherwise).
        buffer = HipVectorReplayBuffer(eng, buffer_size, len(train_envs))
        # device_actor=True: actor + noise on the MI355X (library RNG); over a worker-process env the collector then also
        # overlaps the actor with the env workers when that pays (split_phase="auto")
        train_collector = FastCollector(self.policy, train_envs, buffer, exploration_noise=True, device_actor=device_actor,
                                        split_phase="auto" if device_actor else False)
        test_collector = FastCollector(self.policy, test_envs) if test_envs is not None else None

        def stop_fn(reward, cost):
            return reward > reward_threshold and cost < self.cost_limit

        if save_ckpt:
            self.logger.setup_checkpoint_fn(lambda: {"model": self.state_dict})
        trainer = OnpolicyTrainer(policy=self.policy, train_collector=train_collector,
                                  test_collector=test_collector, max_epoch=epoch, batch_size=batch_size,
                                  cost_limit=self.cost_limit, step_per_epoch=step_per_epoch,
                                  repeat_per_collect=repeat_per_collect, episode_per_test=testing_num,
                                  episode_per_collect=episode_per_collect, stop_fn=stop_fn,
                                  logger=self.logger, resume_from_log=resume,
                                  save_model_interval=save_interval, verbose=verbose,
                                  show_progress=show_progress)
        ep, stat, info = 0, {}, {}
        for ep, stat, info in trainer:
            self.logger.store(tab="train", cost_limit=self.cost_limit)
            if verbose:
                print(f"Epoch: {ep}", info)
        return ep, stat, info
