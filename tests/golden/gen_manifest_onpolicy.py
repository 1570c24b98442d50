#!/usr/bin/env python
"""G8c: key order / shapes of CPO.state_dict(), TRPOLagrangian.state_dict() and FOCOPS.state_dict() as the UNMODIFIED
reference builds them (cpo_agent.py / trpo_lag_agent.py / focops_agent.py network sections); merged into
state_dict_manifest.json.  Build container only.

    python tests/golden/gen_manifest_onpolicy.py
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
from fsrl.policy import CPO, FOCOPS, TRPOLagrangian  # noqa: E402
from fsrl.utils.net.common import ActorCritic  # noqa: E402
from torch import nn  # noqa: E402
from torch.distributions import Independent, Normal  # noqa: E402

from gen_golden import CaptureLogger  # noqa: E402
from ref_shim import ActorProb, Critic, Net, _Box  # noqa: E402

Do, Da, hidden = 6, 3, (64, 64)
spaces = dict(observation_space=_Box(-np.inf, np.inf, (Do, )), action_space=_Box(-1, 1, (Da, )))


def nets():
    actor = ActorProb(Net((Do, ), hidden_sizes=hidden), (Da, ), max_action=1.0, unbounded=False)
    critics = [Critic(Net((Do, ), hidden_sizes=hidden)) for _ in range(2)]
    return actor, critics


def dist(*logits):
    return Independent(Normal(*logits), 1)


def manifest(policy):
    return [[k, list(v.shape) if torch.is_tensor(v) else None] for k, v in policy.state_dict().items()]


a, c = nets()
cpo = CPO(a, c, torch.optim.Adam(nn.ModuleList(c).parameters(), lr=1e-3), dist, logger=CaptureLogger(), cost_limit=10.0,
          **spaces)
a, c = nets()
trpo = TRPOLagrangian(a, c, torch.optim.Adam(ActorCritic(a, c).parameters(), lr=5e-4), dist, logger=CaptureLogger(),
                      cost_limit=10.0, **spaces)
a, c = nets()
focops = FOCOPS(a, c, torch.optim.Adam(a.parameters(), lr=5e-4), torch.optim.Adam(nn.ModuleList(c).parameters(), lr=1e-3),
                dist, logger=CaptureLogger(), cost_limit=10.0, **spaces)
path = os.path.join(HERE, "state_dict_manifest.json")
man = json.load(open(path))
man["cpo_64x64_obs6_act3"] = manifest(cpo)
man["trpo_lag_64x64_obs6_act3"] = manifest(trpo)
man["focops_64x64_obs6_act3"] = manifest(focops)
json.dump(man, open(path, "w"), indent=0)
print({k: len(v) for k, v in man.items()})
