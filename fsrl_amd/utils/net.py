"""Host mirrors of the networks the reference instantiates for the on-policy agents
(tianshou-0.5 `Net` + `ActorProb` / `Critic`, fsrl/agent/ppo_lag_agent.py:136-145).

They exist to (1) own the parameters under the SAME names, shapes and `parameters()` order as
the reference, so `policy.state_dict()` is interchangeable, and (2) serve collector-time actor
inference on the host CPUs.  The training math runs in libfsrl_hip, not here.
"""
from typing import Sequence

import numpy as np
import torch
from torch import nn


class MLP(nn.Module):
    def __init__(self, input_dim: int, output_dim: int = 0, hidden_sizes: Sequence[int] = ()):
        super().__init__()
        sizes = [input_dim] + list(hidden_sizes)
        layers = []
        for i, o in zip(sizes[:-1], sizes[1:]):
            layers += [nn.Linear(i, o), nn.ReLU()]
        if output_dim > 0:
            layers += [nn.Linear(sizes[-1], output_dim)]
        self.output_dim = output_dim or sizes[-1]
        self.model = nn.Sequential(*layers)

    def forward(self, obs):
        return self.model(torch.as_tensor(obs, dtype=torch.float32).flatten(1))


class Net(nn.Module):
    def __init__(self, state_shape, hidden_sizes: Sequence[int] = (), device="cpu"):
        super().__init__()
        self.model = MLP(int(np.prod(state_shape)), 0, hidden_sizes)
        self.output_dim = self.model.output_dim

    def forward(self, obs, state=None, info={}):
        return self.model(obs), state


class ActorProb(nn.Module):
    def __init__(self, preprocess_net, action_shape, max_action=1.0, device="cpu", unbounded=False):
        super().__init__()
        self.preprocess = preprocess_net
        self.output_dim = int(np.prod(action_shape))
        self.mu = MLP(preprocess_net.output_dim, self.output_dim)
        self.sigma_param = nn.Parameter(torch.zeros(self.output_dim, 1))
        self._max = max_action
        self._unbounded = unbounded

    def forward(self, obs, state=None, info={}):
        logits, hidden = self.preprocess(obs, state)
        mu = self.mu(logits)
        if not self._unbounded:
            mu = self._max * torch.tanh(mu)
        sigma = (self.sigma_param.view(1, -1) + torch.zeros_like(mu)).exp()
        return (mu, sigma), state


class Critic(nn.Module):
    def __init__(self, preprocess_net, device="cpu"):
        super().__init__()
        self.preprocess = preprocess_net
        self.last = MLP(preprocess_net.output_dim, 1)

    def forward(self, obs, act=None, info={}):
        obs = torch.as_tensor(obs, dtype=torch.float32).flatten(1)
        if act is not None:
            obs = torch.cat([obs, torch.as_tensor(act, dtype=torch.float32).flatten(1)], dim=1)
        logits, _ = self.preprocess(obs)
        return self.last(logits)


class ActorCritic(nn.Module):
    """Parameter container (fsrl/utils/net/common.py:6-18)."""

    def __init__(self, actor: nn.Module, critics):
        super().__init__()
        self.actor = actor
        self.critics = nn.ModuleList(critics) if isinstance(critics, (list, tuple)) else critics
