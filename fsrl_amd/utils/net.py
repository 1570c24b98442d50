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
    def __init__(self, state_shape, action_shape=0, hidden_sizes: Sequence[int] = (), device="cpu",
                 concat: bool = False):
        super().__init__()
        in_dim = int(np.prod(state_shape)) + (int(np.prod(action_shape)) if concat else 0)
        self.model = MLP(in_dim, 0, hidden_sizes)
        self.output_dim = self.model.output_dim

    def forward(self, obs, state=None, info={}):
        return self.model(obs), state


def mlp_geometry(preprocess_net):
    """(input_dim, hidden_sizes) of a preprocess Net, the way the HIP engine needs it: its Linear + ReLU hidden layers.  Two
    layers of at most 256 units run on the fused kernels (64 / 128 / 256 wide, narrower layers zero-padded); any other tuple
    makes a layered context (include/fsrl_hip.h fsrl_config.n_hidden)."""
    lin = [m for m in preprocess_net.model.model if isinstance(m, nn.Linear)]
    widths = tuple(int(m.out_features) for m in lin)
    if not 1 <= len(lin) <= 8:
        raise ValueError(f"the HIP path runs MLPs with 1 to 8 hidden layers, got hidden_sizes={widths}")
    return int(lin[0].in_features), widths


SIGMA_MIN, SIGMA_MAX = -20, 2


class ActorProb(nn.Module):
    def __init__(self, preprocess_net, action_shape, max_action=1.0, device="cpu", unbounded=False,
                 conditioned_sigma=False):
        super().__init__()
        self.preprocess = preprocess_net
        self.output_dim = int(np.prod(action_shape))
        self.mu = MLP(preprocess_net.output_dim, self.output_dim)
        self._c_sigma = conditioned_sigma
        if conditioned_sigma:
            self.sigma = MLP(preprocess_net.output_dim, self.output_dim)
        else:
            self.sigma_param = nn.Parameter(torch.zeros(self.output_dim, 1))
        self._max = max_action
        self._unbounded = unbounded

    def forward(self, obs, state=None, info={}):
        logits, hidden = self.preprocess(obs, state)
        mu = self.mu(logits)
        if not self._unbounded:
            mu = self._max * torch.tanh(mu)
        if self._c_sigma:
            sigma = torch.clamp(self.sigma(logits), min=SIGMA_MIN, max=SIGMA_MAX).exp()
        else:
            sigma = (self.sigma_param.view(1, -1) + torch.zeros_like(mu)).exp()
        return (mu, sigma), state


class DoubleCritic(nn.Module):
    """Two Q MLPs on concat(obs, act) (fsrl/utils/net/continuous.py:13-101): same attribute names
    (preprocess1/2, last1/2) so the state_dict keys match the reference."""

    def __init__(self, preprocess_net1, preprocess_net2, device="cpu"):
        super().__init__()
        from copy import deepcopy
        self.preprocess1, self.preprocess2 = preprocess_net1, preprocess_net2
        self.last1 = MLP(preprocess_net1.output_dim, 1)
        self.last2 = deepcopy(self.last1)

    def forward(self, obs, act=None, info={}):
        obs = torch.as_tensor(obs, dtype=torch.float32).flatten(1)
        if act is not None:
            obs = torch.cat([obs, torch.as_tensor(act, dtype=torch.float32).flatten(1)], dim=1)
        return [self.last1(self.preprocess1(obs)[0]), self.last2(self.preprocess2(obs)[0])]

    def predict(self, obs, act=None, info={}):
        q = self(obs, act, info)
        return torch.min(q[0], q[1]), q


class Actor(nn.Module):
    """Deterministic actor max_action * tanh(MLP(obs)) (tianshou-0.5 continuous.Actor as FSRL builds it,
    fsrl/agent/ddpg_lag_agent.py:106-107): attribute names preprocess / last like the reference's state_dict."""

    def __init__(self, preprocess_net, action_shape, max_action=1.0, device="cpu"):
        super().__init__()
        self.preprocess = preprocess_net
        self.output_dim = int(np.prod(action_shape))
        self.last = MLP(preprocess_net.output_dim, self.output_dim)
        self._max = max_action

    def forward(self, obs, state=None, info={}):
        logits, hidden = self.preprocess(obs, state)
        return self._max * torch.tanh(self.last(logits)), hidden


class GaussianNoise:
    """Exploration noise N(mu, sigma^2) from numpy's global RNG (tianshou.exploration.GaussianNoise)."""

    def __init__(self, mu: float = 0.0, sigma: float = 1.0):
        self._mu, self._sigma = mu, sigma

    def __call__(self, size):
        return np.random.normal(self._mu, self._sigma, size)


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


class SingleCritic(Critic):
    """tianshou Critic whose forward returns a one-element list, like DoubleCritic's pair
    (fsrl/utils/net/continuous.py:103-160); attribute names preprocess / last as in the reference's state_dict."""

    def forward(self, obs, act=None, info={}):
        return [super().forward(obs, act, info)]

    def predict(self, obs, act=None, info={}):
        q = self(obs, act, info)[0]
        return q, [q]


class ActorCritic(nn.Module):
    """Parameter container (fsrl/utils/net/common.py:6-18)."""

    def __init__(self, actor: nn.Module, critics):
        super().__init__()
        self.actor = actor
        self.critics = nn.ModuleList(critics) if isinstance(critics, (list, tuple)) else critics
