#!/usr/bin/env python
"""G8b: key order / shapes of SACLagrangian.state_dict(), DDPGLagrangian.state_dict() and CVPO.state_dict() as the
UNMODIFIED reference builds them (sac_lag_agent.py:126-176, ddpg_lag_agent.py:106-160, cvpo_agent.py:143-231); merged into
state_dict_manifest.json.  Build container only.

    python tests/golden/gen_manifest_offpolicy.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from fsrl.policy import CVPO, DDPGLagrangian, SACLagrangian  # noqa: E402
from fsrl.utils.net.continuous import DoubleCritic, SingleCritic  # noqa: E402
from torch.distributions import Independent, Normal  # noqa: E402
from torch import nn  # noqa: E402

from ref_shim import Actor, ActorProb, Critic, Net, _Box  # noqa: E402

Do, Da, hidden = 6, 3, (64, 64)
spaces = dict(observation_space=_Box(-np.inf, np.inf, (Do, )), action_space=_Box(-1, 1, (Da, )))


def manifest(policy):
    return [[k, list(v.shape) if torch.is_tensor(v) else None] for k, v in policy.state_dict().items()]


actor = ActorProb(Net((Do, ), hidden_sizes=hidden), (Da, ), max_action=1.0, conditioned_sigma=True, unbounded=True)
critics = [DoubleCritic(Net((Do, ), (Da, ), hidden_sizes=hidden, concat=True), Net((Do, ), (Da, ), hidden_sizes=hidden, concat=True))
           for _ in range(2)]
log_alpha = torch.zeros(1, requires_grad=True)
sac = SACLagrangian(actor=actor, critics=critics, actor_optim=torch.optim.Adam(actor.parameters(), lr=1e-3),
                    critic_optim=torch.optim.Adam(nn.ModuleList(critics).parameters(), lr=1e-3),
                    alpha=(-float(Da), log_alpha, torch.optim.Adam([log_alpha], lr=1e-3)), cost_limit=10.0, **spaces)
actor2 = Actor(Net((Do, ), hidden_sizes=hidden), (Da, ), max_action=1.0)
critics2 = [Critic(Net((Do, ), (Da, ), hidden_sizes=hidden, concat=True)) for _ in range(2)]
ddpg = DDPGLagrangian(actor=actor2, critics=critics2, actor_optim=torch.optim.Adam(actor2.parameters(), lr=1e-3),
                      critic_optim=torch.optim.Adam(nn.ModuleList(critics2).parameters(), lr=1e-3), cost_limit=10.0,
                      exploration_noise=None, **spaces)
actor3 = ActorProb(Net((Do, ), hidden_sizes=hidden), (Da, ), max_action=1.0, conditioned_sigma=True, unbounded=False)
critics3 = [SingleCritic(Net((Do, ), (Da, ), hidden_sizes=hidden, concat=True)) for _ in range(2)]
cvpo = CVPO(actor=actor3, critics=critics3, actor_optim=torch.optim.Adam(actor3.parameters(), lr=1e-3),
            critic_optim=torch.optim.Adam(nn.ModuleList(critics3).parameters(), lr=1e-3), action_space=spaces["action_space"],
            dist_fn=lambda *l: Independent(Normal(*l), 1), max_episode_steps=100, cost_limit=10.0)
path = os.path.join(HERE, "state_dict_manifest.json")
man = json.load(open(path))
man["sac_lag_64x64_obs6_act3"] = manifest(sac)
man["ddpg_lag_64x64_obs6_act3"] = manifest(ddpg)
man["cvpo_64x64_obs6_act3"] = manifest(cvpo)
json.dump(man, open(path, "w"), indent=0)
print({k: len(v) for k, v in man.items()})
