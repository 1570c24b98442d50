"""DDPGLagAgent: keyword arguments and defaults of fsrl/agent/ddpg_lag_agent.py:62-173."""
from typing import Tuple

import numpy as np
import torch

from fsrl_amd.agent._nets import adam, offpolicy_nets
from fsrl_amd.agent.sac_lag_agent import OffpolicyAgent
from fsrl_amd.policy.ddpg_lag import DDPGLagrangian
from fsrl_amd.utils.exp_util import seed_all
from fsrl_amd.utils.logger import DummyLogger
from fsrl_amd.utils.net import GaussianNoise


class DDPGLagAgent(OffpolicyAgent):
    name = "DDPGLagAgent"

    def __init__(self, env, logger=None, cost_limit: float = 10, device: str = "cuda:0", thread: int = 4, seed: int = 10,
                 actor_lr: float = 1e-4, critic_lr: float = 1e-3, hidden_sizes: Tuple[int, ...] = (128, 128),
                 tau: float = 0.05, exploration_noise: float = 0.1, n_step: int = 3, use_lagrangian: bool = True,
                 lagrangian_pid: Tuple[float, ...] = (0.5, 0.001, 0.1), rescaling: bool = True, gamma: float = 0.99,
                 deterministic_eval: bool = True, action_scaling: bool = True, action_bound_method: str = "clip",
                 lr_scheduler=None, training_num: int = 10, buffer_size: int = 100000, reference_rng: bool = False) -> None:
        super().__init__()
        self.logger = logger if logger is not None else DummyLogger()
        self.cost_limit = cost_limit
        assert np.isscalar(cost_limit), "the HIP path supports one cost constraint"
        seed_all(seed)
        torch.set_num_threads(thread)
        actor, critics = offpolicy_nets(env, hidden_sizes, "plain", deterministic=True)
        actor_optim, critic_optim = adam(actor, actor_lr), adam(critics, critic_lr)
        self.policy = DDPGLagrangian(actor=actor, critics=critics, actor_optim=actor_optim, critic_optim=critic_optim,
                                     logger=self.logger, tau=tau, exploration_noise=GaussianNoise(sigma=exploration_noise),
                                     n_step=n_step, use_lagrangian=use_lagrangian, lagrangian_pid=lagrangian_pid,
                                     cost_limit=cost_limit, rescaling=rescaling, gamma=gamma,
                                     reward_normalization=False, deterministic_eval=deterministic_eval,
                                     action_scaling=action_scaling, action_bound_method=action_bound_method,
                                     observation_space=env.observation_space, action_space=env.action_space,
                                     lr_scheduler=lr_scheduler, device=device, env_num=training_num,
                                     buffer_size=buffer_size, reference_rng=reference_rng, seed=seed)
