"""CPOAgent and TRPOLagAgent: keyword arguments and defaults of fsrl/agent/cpo_agent.py:69-181 and
fsrl/agent/trpo_lag_agent.py:75-190 over the HIP-backed policies."""
from typing import Optional, Tuple

import numpy as np
import torch

from fsrl_amd.agent._nets import adam, independent_normal, onpolicy_nets
from fsrl_amd.agent.base_agent import OnpolicyAgent
from fsrl_amd.policy.cpo import CPO
from fsrl_amd.policy.trpo_lag import TRPOLagrangian
from fsrl_amd.utils.exp_util import seed_all
from fsrl_amd.utils.logger import DummyLogger


class CPOAgent(OnpolicyAgent):
    name = "CPOAgent"

    def __init__(self, env, logger=None, cost_limit: float = 10, device: str = "cuda:0", thread: int = 4,
                 seed: int = 10, lr: float = 1e-3, hidden_sizes: Tuple[int, ...] = (128, 128),
                 unbounded: bool = False, last_layer_scale: bool = False, target_kl: float = 0.01,
                 backtrack_coeff: float = 0.8, damping_coeff: float = 0.1, max_backtracks: int = 10,
                 optim_critic_iters: int = 10, l2_reg: float = 0.001, gae_lambda: float = 0.95,
                 advantage_normalization: bool = True, gamma: float = 0.99, max_batchsize: int = 99999,
                 reward_normalization: bool = False, deterministic_eval: bool = True,
                 action_scaling: bool = True, action_bound_method: str = "clip", lr_scheduler=None,
                 training_num: int = 20, buffer_size: int = 100000) -> None:
        super().__init__()
        self.logger = logger if logger is not None else DummyLogger()
        self.cost_limit = cost_limit
        if not np.isscalar(cost_limit):
            raise RuntimeError("CPO does not support multiple costs.")
        seed_all(seed)
        torch.set_num_threads(thread)
        actor, critic, _ = onpolicy_nets(env, hidden_sizes, 2, last_layer_scale, unbounded)
        optim = adam(critic, lr)
        self.policy = CPO(actor, critic, optim, independent_normal, logger=self.logger, target_kl=target_kl,
                          backtrack_coeff=backtrack_coeff, damping_coeff=damping_coeff,
                          max_backtracks=max_backtracks, optim_critic_iters=optim_critic_iters,
                          l2_reg=l2_reg, gae_lambda=gae_lambda,
                          advantage_normalization=advantage_normalization, cost_limit=cost_limit,
                          gamma=gamma, max_batchsize=max_batchsize,
                          reward_normalization=reward_normalization, deterministic_eval=deterministic_eval,
                          action_scaling=action_scaling, action_bound_method=action_bound_method,
                          observation_space=env.observation_space, action_space=env.action_space,
                          lr_scheduler=lr_scheduler, device=device, env_num=training_num,
                          buffer_size=buffer_size)


class TRPOLagAgent(OnpolicyAgent):
    name = "TRPOLagAgent"

    def __init__(self, env, logger=None, cost_limit: float = 10, device: str = "cuda:0", thread: int = 4,
                 seed: int = 10, lr: float = 5e-4, hidden_sizes: Tuple[int, ...] = (128, 128),
                 unbounded: bool = False, last_layer_scale: bool = False, target_kl: float = 0.001,
                 backtrack_coeff: float = 0.8, max_backtracks: int = 10, optim_critic_iters: int = 20,
                 gae_lambda: float = 0.95, advantage_normalization: bool = True,
                 use_lagrangian: bool = True, lagrangian_pid: Tuple = (0.05, 0.0005, 0.1),
                 rescaling: bool = True, gamma: float = 0.99, max_batchsize: int = 99999,
                 reward_normalization: bool = False, deterministic_eval: bool = True,
                 action_scaling: bool = True, action_bound_method: str = "clip", lr_scheduler=None,
                 training_num: int = 20, buffer_size: int = 100000) -> None:
        super().__init__()
        self.logger = logger if logger is not None else DummyLogger()
        self.cost_limit = cost_limit
        assert np.isscalar(cost_limit)
        seed_all(seed)
        torch.set_num_threads(thread)
        actor, critic, ac = onpolicy_nets(env, hidden_sizes, 2, last_layer_scale, unbounded)
        optim = adam(ac, lr)
        self.policy = TRPOLagrangian(actor, critic, optim, independent_normal, logger=self.logger, target_kl=target_kl,
                                     backtrack_coeff=backtrack_coeff, max_backtracks=max_backtracks,
                                     optim_critic_iters=optim_critic_iters, gae_lambda=gae_lambda,
                                     advantage_normalization=advantage_normalization,
                                     use_lagrangian=use_lagrangian, lagrangian_pid=lagrangian_pid,
                                     cost_limit=cost_limit, rescaling=rescaling, gamma=gamma,
                                     max_batchsize=max_batchsize, reward_normalization=reward_normalization,
                                     deterministic_eval=deterministic_eval, action_scaling=action_scaling,
                                     action_bound_method=action_bound_method,
                                     observation_space=env.observation_space,
                                     action_space=env.action_space, lr_scheduler=lr_scheduler, device=device,
                                     env_num=training_num, buffer_size=buffer_size)


class FOCOPSAgent(OnpolicyAgent):
    """Keyword arguments and defaults of fsrl/agent/focops_agent.py:79-214."""
    name = "FOCOPSAgent"

    def __init__(self, env, logger=None, cost_limit: float = 10, device: str = "cuda:0", thread: int = 4, seed: int = 10,
                 actor_lr: float = 5e-4, critic_lr: float = 1e-3, hidden_sizes: Tuple[int, ...] = (128, 128),
                 unbounded: bool = False, last_layer_scale: bool = False, auto_nu: bool = True, nu: float = 0.01,
                 nu_max: float = 2.0, nu_lr: float = 1e-2, l2_reg: float = 1e-3, delta: float = 0.02, eta: float = 0.02,
                 tem_lambda: float = 0.95, gae_lambda: float = 0.95, max_grad_norm: Optional[float] = 0.5,
                 advantage_normalization: bool = True, recompute_advantage: bool = False, gamma: float = 0.99,
                 max_batchsize: int = 100000, reward_normalization: bool = False, deterministic_eval: bool = True,
                 action_scaling: bool = True, action_bound_method: str = "clip", lr_scheduler=None, training_num: int = 20,
                 buffer_size: int = 100000) -> None:
        super().__init__()
        from fsrl_amd.policy.focops import FOCOPS
        self.logger = logger if logger is not None else DummyLogger()
        self.cost_limit = cost_limit
        assert np.isscalar(cost_limit) and auto_nu
        seed_all(seed)
        torch.set_num_threads(thread)
        actor, critic, _ = onpolicy_nets(env, hidden_sizes, 2, last_layer_scale, unbounded)
        actor_optim, critic_optim = adam(actor, actor_lr), adam(critic, critic_lr)
        self.policy = FOCOPS(actor, critic, actor_optim, critic_optim, independent_normal, logger=self.logger, cost_limit=cost_limit,
                             nu=(nu_max, nu_lr, torch.zeros(1)), l2_reg=l2_reg, delta=delta, eta=eta, tem_lambda=tem_lambda,
                             gae_lambda=gae_lambda, max_grad_norm=max_grad_norm,
                             advantage_normalization=advantage_normalization, recompute_advantage=recompute_advantage,
                             gamma=gamma, max_batchsize=max_batchsize, reward_normalization=reward_normalization,
                             deterministic_eval=deterministic_eval, action_scaling=action_scaling,
                             action_bound_method=action_bound_method, observation_space=env.observation_space,
                             action_space=env.action_space, lr_scheduler=lr_scheduler, device=device,
                             env_num=training_num, buffer_size=buffer_size)
