"""Network construction shared by the agents: the tianshou-shaped modules of fsrl_amd/utils/net.py, built and initialised
the way the reference agents do it (orthogonal weights, zero biases, optional 0.01 scaling of the mean head; e.g.
fsrl/agent/ppo_lag_agent.py:136-160, fsrl/agent/sac_lag_agent.py:126-176).  Construction order = torch RNG order:
actor first, then the critics, then one orthogonal_ per Linear in ActorCritic(actor, critics).modules() order."""
from typing import List, Sequence

import torch
from torch import nn
from torch.distributions import Independent, Normal

from fsrl_amd.utils.net import Actor, ActorCritic, ActorProb, Critic, DoubleCritic, Net, SingleCritic


def independent_normal(*logits):
    """dist_fn of every Gaussian policy here: Independent(Normal(mu, sigma), 1)"""
    return Independent(Normal(*logits), 1)


def _shapes(env):
    return env.observation_space.shape, env.action_space.shape, float(env.action_space.high[0])


def _initialise(actor: nn.Module, critics: Sequence[nn.Module], last_layer_scale: bool) -> ActorCritic:
    both = ActorCritic(actor, list(critics))
    for layer in (m for m in both.modules() if isinstance(m, nn.Linear)):
        nn.init.orthogonal_(layer.weight)
        nn.init.zeros_(layer.bias)
    if last_layer_scale:                          # near-zero initial means (arXiv 2006.05990, fig. 24)
        for layer in (m for m in actor.mu.modules() if isinstance(m, nn.Linear)):
            nn.init.zeros_(layer.bias)
            layer.weight.data.mul_(0.01)
    return both


def onpolicy_nets(env, hidden_sizes, n_critics: int, last_layer_scale: bool = False, unbounded: bool = False):
    """Gaussian actor with a free sigma_param (-0.5) and `n_critics` state-value critics (PPO-Lag, CPO, TRPO-Lag, FOCOPS)."""
    obs_shape, act_shape, amax = _shapes(env)
    actor = ActorProb(Net(obs_shape, hidden_sizes=hidden_sizes), act_shape, max_action=amax, unbounded=unbounded)
    critics = [Critic(Net(obs_shape, hidden_sizes=hidden_sizes)) for _ in range(n_critics)]
    nn.init.constant_(actor.sigma_param, -0.5)
    return actor, critics, _initialise(actor, critics, last_layer_scale)


def _q_net(env, hidden_sizes):
    obs_shape, act_shape, _ = _shapes(env)
    return Net(obs_shape, act_shape, hidden_sizes=hidden_sizes, concat=True)


def offpolicy_nets(env, hidden_sizes, critic: str, n_critics: int = 2, unbounded: bool = True, deterministic: bool = False,
                   last_layer_scale: bool = False):
    """Actor + `n_critics` action-value critics of the replay agents.  critic: "double" (SAC-Lag; CVPO with
    double_critic), "single" (CVPO default: one Q in a list), "plain" (DDPG-Lag: tianshou Critic).  The actor is the
    deterministic max_action * tanh(MLP) for DDPG-Lag, else a Gaussian with a state-conditioned sigma head."""
    obs_shape, act_shape, amax = _shapes(env)
    if deterministic:
        actor = Actor(Net(obs_shape, hidden_sizes=hidden_sizes), act_shape, max_action=amax)
    else:
        actor = ActorProb(Net(obs_shape, hidden_sizes=hidden_sizes), act_shape, max_action=amax, conditioned_sigma=True,
                          unbounded=unbounded)
    make = {"double": lambda: DoubleCritic(_q_net(env, hidden_sizes), _q_net(env, hidden_sizes)),
            "single": lambda: SingleCritic(_q_net(env, hidden_sizes)),
            "plain": lambda: Critic(_q_net(env, hidden_sizes))}[critic]
    critics: List[nn.Module] = [make() for _ in range(n_critics)]
    both = ActorCritic(actor, critics)
    for layer in (m for m in both.modules() if isinstance(m, nn.Linear)):
        nn.init.orthogonal_(layer.weight)
        nn.init.zeros_(layer.bias)
    if last_layer_scale and not deterministic:
        for layer in (m for m in actor.mu.modules() if isinstance(m, nn.Linear)):
            nn.init.zeros_(layer.bias)
            layer.weight.data.mul_(0.01)
    return actor, critics


def adam(modules, lr: float) -> torch.optim.Adam:
    mods = modules if isinstance(modules, nn.Module) else nn.ModuleList(list(modules))
    return torch.optim.Adam(mods.parameters(), lr=lr)
