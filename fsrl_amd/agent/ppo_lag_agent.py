"""PPOLagAgent: builds the networks (orthogonal init, sigma_param = -0.5), the optimiser and the
HIP-backed PPOLagrangian policy with the keyword arguments and defaults of
fsrl/agent/ppo_lag_agent.py:82-200.  `device` selects the GPU the engine runs on;
`training_num` (not in the reference signature) is the number of env sub-buffers the
HIP-resident store is created with (the reference sizes its buffer later, inside learn())."""
from typing import Optional, Tuple

import numpy as np
import torch

from fsrl_amd.agent._nets import adam, independent_normal, onpolicy_nets
from fsrl_amd.agent.base_agent import OnpolicyAgent
from fsrl_amd.policy import PPOLagrangian
from fsrl_amd.utils.exp_util import seed_all
from fsrl_amd.utils.logger import BaseLogger, DummyLogger


class PPOLagAgent(OnpolicyAgent):
    name = "PPOLagAgent"

    def __init__(self, env, logger: BaseLogger = None, cost_limit: float = 10, device: str = "cuda:0",
                 thread: int = 4, seed: int = 10, lr: float = 5e-4,
                 hidden_sizes: Tuple[int, ...] = (128, 128), unbounded: bool = False,
                 last_layer_scale: bool = False,
                 # PPO specific arguments
                 target_kl: float = 0.02, vf_coef: float = 0.25, max_grad_norm: Optional[float] = None,
                 gae_lambda: float = 0.95, eps_clip: float = 0.2, dual_clip: Optional[float] = None,
                 value_clip: bool = False, advantage_normalization: bool = True,
                 recompute_advantage: bool = False,
                 # Lagrangian specific arguments
                 use_lagrangian: bool = True, lagrangian_pid: Tuple = (0.05, 0.0005, 0.1),
                 rescaling: bool = True,
                 # Base policy common arguments
                 gamma: float = 0.99, max_batchsize: int = 99999, reward_normalization: bool = False,
                 deterministic_eval: bool = True, action_scaling: bool = True,
                 action_bound_method: str = "clip", lr_scheduler=None,
                 # store geometry of the HIP engine
                 training_num: int = 20, buffer_size: int = 100000) -> None:
        super().__init__()
        self.logger = logger if logger is not None else DummyLogger()
        self.cost_limit = cost_limit
        cost_dim = 1 if np.isscalar(cost_limit) else len(cost_limit)
        seed_all(seed)
        torch.set_num_threads(thread)
        actor, critic, actor_critic = onpolicy_nets(env, hidden_sizes, 1 + cost_dim, last_layer_scale, unbounded)
        optim = adam(actor_critic, lr)
        dist = independent_normal
        self.policy = PPOLagrangian(
            actor, critic, optim, dist, logger=self.logger, target_kl=target_kl, vf_coef=vf_coef,
            max_grad_norm=max_grad_norm, gae_lambda=gae_lambda, eps_clip=eps_clip, dual_clip=dual_clip,
            value_clip=value_clip, advantage_normalization=advantage_normalization,
            recompute_advantage=recompute_advantage, use_lagrangian=use_lagrangian,
            lagrangian_pid=lagrangian_pid, cost_limit=cost_limit, rescaling=rescaling, gamma=gamma,
            max_batchsize=max_batchsize, reward_normalization=reward_normalization,
            deterministic_eval=deterministic_eval, action_scaling=action_scaling,
            action_bound_method=action_bound_method, observation_space=env.observation_space,
            action_space=env.action_space, lr_scheduler=lr_scheduler, device=device,
            env_num=training_num, buffer_size=buffer_size)
