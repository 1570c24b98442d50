"""Flat parameter layout shared by the oracle and the HIP library (include/fsrl_hip.h).

The flat float32 vector is the concatenation, in `torch.nn.Module.parameters()` order, of
the reference's networks as instantiated at fsrl/agent/ppo_lag_agent.py:136-153 (tianshou
0.5 `Net` + `ActorProb` / `Critic`):

  gaussian actor (state-independent sigma), hidden_sizes = (H1, ..., HL):
      sigma_param[Da]  W1[H1,Do] b1[H1]  W2[H2,H1] b2[H2] ... W{L+1}[Da,HL] b{L+1}[Da]      (the last pair is `mu`)
  V critic:
      W1[H1,Do] b1[H1]  W2[H2,H1] b2[H2] ... W{L+1}[1,HL] b{L+1}[1]                          (the last pair is `last`)
  on-policy policy (PPO-Lag / TRPO-Lag / CPO):  actor ++ critic_0 (reward) ++ critic_1 (cost) ...

All weights are row-major [out_features, in_features] exactly like `nn.Linear.weight`.
"""
from collections import OrderedDict

import numpy as np


def _trunk_spec(obs_dim, out_dim, hidden):
    """W1 b1 ... W{L+1} b{L+1}: L hidden Linear + ReLU layers of the given widths, then the head (tianshou 0.5 `Net` with
    `hidden_sizes` of any length + the `mu` / `last` MLP, fsrl/agent/ppo_lag_agent.py:91,136-145)."""
    sizes = [int(obs_dim)] + [int(h) for h in hidden] + [int(out_dim)]
    items = []
    for l in range(1, len(sizes)):
        items += [(f"W{l}", (sizes[l], sizes[l - 1])), (f"b{l}", (sizes[l], ))]
    return items


def gauss_actor_spec(obs_dim, act_dim, hidden):
    return OrderedDict([("sigma_param", (act_dim, ))] + _trunk_spec(obs_dim, act_dim, hidden))


def v_critic_spec(obs_dim, hidden):
    return OrderedDict(_trunk_spec(obs_dim, 1, hidden))


def spec_size(spec):
    return int(sum(int(np.prod(s)) for s in spec.values()))


def onpolicy_specs(obs_dim, act_dim, hidden, n_critics=2):
    """[actor, critic_0, ..., critic_{C-1}] specs."""
    return [gauss_actor_spec(obs_dim, act_dim, hidden)] + \
        [v_critic_spec(obs_dim, hidden) for _ in range(n_critics)]


def views(flat, spec, offset=0):
    """Named views into a flat tensor/array (no copy)."""
    out = OrderedDict()
    for name, shape in spec.items():
        n = int(np.prod(shape))
        out[name] = flat[offset:offset + n].reshape(shape)
        offset += n
    return out, offset


def state_dict_keys_onpolicy(n_critics=2):
    """Key order of `policy.state_dict()` for the on-policy policies, as emitted by the
    reference (actor.*, critics.*, then the `_actor_critic.*` aliases; `_extra_state` for
    the Lagrangian policies -- fsrl/policy/lagrangian_base.py:122-143).  Returns
    [(key, (net_index, param_name))]."""
    keys = [("_extra_state", None)]  # root module's extra state is emitted first

    def net_keys(prefix, net, is_actor):
        ks = []
        if is_actor:
            ks.append((prefix + "sigma_param", (net, "sigma_param")))
        pre = prefix + "preprocess.model.model."
        ks += [(pre + "0.weight", (net, "W1")), (pre + "0.bias", (net, "b1")),
               (pre + "2.weight", (net, "W2")), (pre + "2.bias", (net, "b2"))]
        head = prefix + ("mu" if is_actor else "last") + ".model.0."
        ks += [(head + "weight", (net, "W3")), (head + "bias", (net, "b3"))]
        return ks

    for root in ("", "_actor_critic."):
        keys += net_keys(root + "actor.", 0, True)
        for c in range(n_critics):
            keys += net_keys(root + f"critics.{c}.", 1 + c, False)
    return keys
