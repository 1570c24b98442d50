"""SACLagAgent + the off-policy learn loop: keyword arguments and defaults of
fsrl/agent/sac_lag_agent.py:75-203 and fsrl/agent/base_agent.py:108-209."""
from typing import Optional, Tuple

import numpy as np
import torch

from fsrl_amd.agent._nets import adam, offpolicy_nets
from fsrl_amd.agent.base_agent import BaseAgent
from fsrl_amd.data import FastCollector, HipVectorReplayBuffer
from fsrl_amd.policy.sac_lag import SACLagrangian
from fsrl_amd.trainer.offpolicy import OffpolicyTrainer
from fsrl_amd.utils.exp_util import seed_all
from fsrl_amd.utils.logger import DummyLogger


class OffpolicyAgent(BaseAgent):
    name = "OffpolicyAgent"

    def __init__(self) -> None:
        pass

    def learn(self, train_envs, test_envs=None, epoch: int = 300, episode_per_collect: int = 5,
              step_per_epoch: int = 3000, update_per_step: float = 0.1, buffer_size: Optional[int] = None,
              testing_num: int = 2, batch_size: int = 256, reward_threshold: float = 450,
              save_interval: int = 4, resume: bool = False, save_ckpt: bool = True, verbose: bool = True,
              show_progress: bool = True, device_actor: bool = False):
        assert self.policy is not None, "The policy is not initialized"
        self.policy.train()
        eng = self.policy.engine
        from fsrl_amd.env.venv import as_vector_env
        train_envs = as_vector_env(train_envs)                 # a single env: one sub-buffer (base_agent.py:160-163)
        assert eng.cfg.env_num >= len(train_envs)
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
        trainer = OffpolicyTrainer(policy=self.policy, train_collector=train_collector,
                                   test_collector=test_collector, max_epoch=epoch, batch_size=batch_size,
                                   cost_limit=self.cost_limit, step_per_epoch=step_per_epoch,
                                   update_per_step=update_per_step, episode_per_test=testing_num,
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


class SACLagAgent(OffpolicyAgent):
    name = "SACLagAgent"

    def __init__(self, env, logger=None, cost_limit: float = 10, device: str = "cuda:0", thread: int = 4,
                 seed: int = 10, actor_lr: float = 5e-4, critic_lr: float = 1e-3,
                 hidden_sizes: Tuple[int, ...] = (128, 128), auto_alpha: bool = True, alpha_lr: float = 3e-4,
                 alpha: float = 0.002, tau: float = 0.05, n_step: int = 2, use_lagrangian: bool = True,
                 lagrangian_pid: Tuple[float, ...] = (0.05, 0.0005, 0.1), rescaling: bool = True,
                 gamma: float = 0.99, conditioned_sigma: bool = True, unbounded: bool = True,
                 last_layer_scale: bool = False, deterministic_eval: bool = False, action_scaling: bool = True,
                 action_bound_method: str = "clip", lr_scheduler=None, training_num: int = 10,
                 buffer_size: int = 100000, reference_rng: bool = False) -> None:
        super().__init__()
        self.logger = logger if logger is not None else DummyLogger()
        self.cost_limit = cost_limit
        assert np.isscalar(cost_limit) and conditioned_sigma and unbounded, \
            "the HIP SAC path: one cost, state-conditioned sigma, unbounded mean (the reference defaults)"
        seed_all(seed)
        torch.set_num_threads(thread)
        actor, critics = offpolicy_nets(env, hidden_sizes, "double", unbounded=True, last_layer_scale=last_layer_scale)
        actor_optim, critic_optim = adam(actor, actor_lr), adam(critics, critic_lr)
        if auto_alpha:
            target_entropy = -float(np.prod(env.action_space.shape))
            log_alpha = torch.zeros(1, requires_grad=True)
            alpha = (target_entropy, log_alpha, torch.optim.Adam([log_alpha], lr=alpha_lr))
        self.policy = SACLagrangian(actor=actor, critics=critics, actor_optim=actor_optim,
                                    critic_optim=critic_optim, logger=self.logger, alpha=alpha, tau=tau,
                                    gamma=gamma, exploration_noise=None, n_step=n_step,
                                    use_lagrangian=use_lagrangian, lagrangian_pid=lagrangian_pid,
                                    cost_limit=cost_limit, rescaling=rescaling, reward_normalization=False,
                                    deterministic_eval=deterministic_eval, action_scaling=action_scaling,
                                    action_bound_method=action_bound_method,
                                    observation_space=env.observation_space, action_space=env.action_space,
                                    lr_scheduler=lr_scheduler, device=device, env_num=training_num,
                                    buffer_size=buffer_size, reference_rng=reference_rng, seed=seed)
