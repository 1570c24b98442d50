"""Host-side logic of the layered contexts (hidden_sizes of any depth / width; include/fsrl_hip.h fsrl_config.n_hidden): the
configuration hand-over, the network geometry the policies read off the torch mirrors, and the oracle's flat layouts.  No GPU."""
import numpy as np
import pytest
import torch


def test_engine_config_hands_hidden_sizes_over_as_a_tuple():
    from fsrl_amd.engine import EngineConfig
    c = EngineConfig(obs_dim=7, act_dim=3, hidden_sizes=(64, 48, 32)).to_c()
    assert (c.hidden, c.hidden1, c.hidden2, c.n_hidden) == (0, 0, 0, 3) and list(c.hidden_sizes)[:4] == [64, 48, 32, 0]
    assert c.force_layered == 0
    c = EngineConfig(hidden_sizes=(100, 50), force_layered=True).to_c()          # two layers <= 256: fused unless forced
    assert c.n_hidden == 2 and list(c.hidden_sizes)[:2] == [100, 50] and c.force_layered == 1
    c = EngineConfig(hidden=256).to_c()                                            # the old spelling: two layers of `hidden`
    assert (c.hidden, c.n_hidden) == (256, 0)
    for bad in ((), (8, ) * 9):
        with pytest.raises(ValueError, match="1 to 8 hidden layers"):
            EngineConfig(hidden_sizes=bad).to_c()


@pytest.mark.parametrize("hidden", [(96, ), (64, 48, 32), (300, 260), (40, 72, 72, 24)])
def test_mlp_geometry_and_parameter_order_match_the_oracle_layout(hidden):
    """The flat vector the policies push (torch parameters() order of ActorProb + critics) is the oracle's / the library's
    layout for any depth: sigma_param, W1, b1, ..., W{L+1}, b{L+1} per network (oracle/layout.py)."""
    from fsrl_amd.utils.net import ActorProb, Critic, Net, mlp_geometry
    from oracle import layout
    Do, Da = 9, 3
    actor = ActorProb(Net((Do, ), hidden_sizes=hidden), (Da, ))
    critics = [Critic(Net((Do, ), hidden_sizes=hidden)) for _ in range(2)]
    assert mlp_geometry(actor.preprocess) == (Do, tuple(hidden))
    specs = layout.onpolicy_specs(Do, Da, hidden, 2)
    shapes = [tuple(p.shape) for m in [actor] + critics for p in m.parameters()]
    want = [s if len(s) > 1 or k != "sigma_param" else (s[0], 1) for spec in specs for k, s in spec.items()]
    assert shapes == want
    assert sum(int(np.prod(s)) for s in shapes) == sum(layout.spec_size(s) for s in specs)


def test_mlp_geometry_refuses_more_than_eight_layers():
    from fsrl_amd.utils.net import Net, mlp_geometry
    with pytest.raises(ValueError, match="1 to 8 hidden layers"):
        mlp_geometry(Net((4, ), hidden_sizes=(8, ) * 9))


@pytest.mark.parametrize("hidden", [(48, 64, 40), (272, )])
def test_replay_oracles_take_any_depth(hidden):
    """SAC / DDPG / CVPO oracles: parameter counts of the generalised specs against torch mirrors of the same networks."""
    from fsrl_amd.utils.net import Actor, ActorProb, Critic, DoubleCritic, Net, SingleCritic
    from oracle.cvpo import single_critic_spec
    from oracle.ddpg_lag import mlp_spec
    from oracle.sac_lag import actor_spec, double_critic_spec
    Do, Da = 6, 3
    n = lambda spec: sum(int(np.prod(s)) for s in spec.values())  # noqa: E731
    cnt = lambda m: sum(p.numel() for p in m.parameters())  # noqa: E731
    a = ActorProb(Net((Do, ), hidden_sizes=hidden), (Da, ), conditioned_sigma=True, unbounded=True)
    assert n(actor_spec(Do, Da, hidden)) == cnt(a)
    dc = DoubleCritic(Net((Do, ), (Da, ), hidden_sizes=hidden, concat=True), Net((Do, ), (Da, ), hidden_sizes=hidden, concat=True))
    assert n(double_critic_spec(Do, Da, hidden)) == cnt(dc)
    assert n(mlp_spec(Do, Da, hidden)) == cnt(Actor(Net((Do, ), hidden_sizes=hidden), (Da, )))
    assert n(mlp_spec(Do + Da, 1, hidden)) == cnt(Critic(Net((Do, ), (Da, ), hidden_sizes=hidden, concat=True)))
    assert n(single_critic_spec(Do, Da, hidden)) == cnt(SingleCritic(Net((Do, ), (Da, ), hidden_sizes=hidden, concat=True)))
    assert [k for k in double_critic_spec(Do, Da, hidden)][-1] == f"b{len(hidden) + 1}_2"
    torch.manual_seed(0)
