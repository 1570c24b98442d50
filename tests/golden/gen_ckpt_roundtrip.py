#!/usr/bin/env python
"""f3: load the HIP-TRAINED checkpoints (tests/golden/hip_ckpt_{ppo,sac}.pt, written on an MI355X by
tools/make_hip_checkpoint.py through policy.state_dict()) into the UNMODIFIED reference policies with
load_state_dict(strict=True) -- what fsrl/utils/exp_util.py:60-84 + agent.evaluate(state_dict=...) (base_agent.py:75-76) do
-- and record the reference's forward() outputs on the probe observations.  Build container only.

    python tests/golden/gen_ckpt_roundtrip.py [out.npz]      # default: tests/golden/hip_ckpt_ref_forward.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from fsrl.policy import PPOLagrangian, SACLagrangian  # noqa: E402
from fsrl.utils.net.common import ActorCritic  # noqa: E402
from fsrl.utils.net.continuous import DoubleCritic  # noqa: E402
from tianshou.data import Batch  # noqa: E402  (the shim's)
from torch import nn  # noqa: E402
from torch.distributions import Independent, Normal  # noqa: E402

from ref_shim import ActorProb, Critic, Net, _Box  # noqa: E402


def ref_ppo(net):
    Do, Da, h = net["obs_dim"], net["act_dim"], tuple(net["hidden"])
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), max_action=1.0, unbounded=False)
    critics = [Critic(Net((Do, ), hidden_sizes=h)) for _ in range(2)]
    optim = torch.optim.Adam(ActorCritic(actor, critics).parameters(), lr=5e-4)
    return PPOLagrangian(actor, critics, optim, lambda *l: Independent(Normal(*l), 1), cost_limit=10.0,
                         observation_space=_Box(-np.inf, np.inf, (Do, )), action_space=_Box(-1, 1, (Da, )))


def ref_sac(net):
    Do, Da, h = net["obs_dim"], net["act_dim"], tuple(net["hidden"])
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), max_action=1.0, conditioned_sigma=True, unbounded=True)
    critics = [DoubleCritic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True), Net((Do, ), (Da, ), hidden_sizes=h, concat=True))
               for _ in range(2)]
    la = torch.zeros(1, requires_grad=True)
    return SACLagrangian(actor=actor, critics=critics, actor_optim=torch.optim.Adam(actor.parameters(), lr=1e-3),
                         critic_optim=torch.optim.Adam(nn.ModuleList(critics).parameters(), lr=1e-3),
                         alpha=(-float(Da), la, torch.optim.Adam([la], lr=1e-3)), cost_limit=10.0,
                         observation_space=_Box(-np.inf, np.inf, (Do, )), action_space=_Box(-1, 1, (Da, )))


def main():
    out = {}
    ck = torch.load(os.path.join(HERE, "hip_ckpt_ppo.pt"), weights_only=False)
    pol = ref_ppo(ck["net"])
    missing = pol.load_state_dict(ck["model"], strict=True)
    pol.eval()
    with torch.no_grad():
        res = pol(Batch(obs=ck["probe_obs"], info={}))
        out["ppo_mu"], out["ppo_sigma"] = res.logits[0].numpy(), res.logits[1].numpy()
        out["ppo_act_eval"] = res.act.numpy()
        out["ppo_values"] = np.stack([c(ck["probe_obs"]).flatten().numpy() for c in pol.critics])
    out["ppo_lagrangian"] = np.array([o.get_lag() for o in pol.lag_optims])
    print("ppo: strict load ok", missing, "max |mu_ref - mu_device| =", np.abs(out["ppo_mu"] - ck["device_mu"]).max())
    ck = torch.load(os.path.join(HERE, "hip_ckpt_sac.pt"), weights_only=False)
    pol = ref_sac(ck["net"])
    missing = pol.load_state_dict(ck["model"], strict=True)
    pol.eval()
    with torch.no_grad():
        (mu, sigma), _ = pol.actor(ck["probe_obs"])
        out["sac_mu"], out["sac_sigma"] = mu.numpy(), sigma.numpy()
        out["sac_q"] = np.array([[x.flatten().numpy() for x in c(ck["probe_obs"], ck["probe_act"])] for c in pol.critics])
        out["sac_q_old"] = np.array([[x.flatten().numpy() for x in c(ck["probe_obs"], ck["probe_act"])]
                                     for c in pol.critics_old])
    print("sac: strict load ok", missing, "max |mu_ref - mu_device| =", np.abs(out["sac_mu"] - ck["device_mu"]).max(),
          " max |q_ref - q_host_mirror| =", np.abs(out["sac_q"] - ck["host_q"]).max())
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "hip_ckpt_ref_forward.npz")
    np.savez_compressed(dst, **out)


if __name__ == "__main__":
    main()
