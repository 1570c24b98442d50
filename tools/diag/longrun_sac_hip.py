"""Diagnosis twin of the reference long run (tests/golden/longrun_ref_sac.py protocol): SAC-Lagrangian through the facade on
the synthetic env, 20 cycles x (10 episodes of 100 steps, 200 updates of batch 256); prints cycle, reward, cost, lambda, alpha."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from test_gpu_loop import _rollout
from fsrl_amd.data import HipVectorReplayBuffer
from fsrl_amd.env import Box, SyntheticSafetyVectorEnv
from fsrl_amd.policy import SACLagrangian
from fsrl_amd.utils.net import ActorCritic, ActorProb, DoubleCritic, Net

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 20
Do, Da, h, E, ep_len, B, upc = 8, 2, (64, 64), 10, 100, 256, 200
random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), conditioned_sigma=True, unbounded=True)
critics = [DoubleCritic(Net((Do, ), (Da, ), hidden_sizes=h, concat=True), Net((Do, ), (Da, ), hidden_sizes=h, concat=True))
           for _ in range(2)]
for m in ActorCritic(actor, critics).modules():
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.orthogonal_(m.weight); torch.nn.init.zeros_(m.bias)
la = torch.zeros(1, requires_grad=True)
pol = SACLagrangian(actor, critics, torch.optim.Adam(actor.parameters(), lr=5e-4),
                    torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=1e-3), logger=None,
                    alpha=(-float(Da), la, torch.optim.Adam([la], lr=3e-4)), tau=0.05, n_step=2, cost_limit=20.0, gamma=0.99,
                    observation_space=Box(-np.inf, np.inf, (Do, )), action_space=Box(-1, 1, (Da, )), device=0, env_num=E,
                    buffer_size=50000, reference_rng=("--refrng" in sys.argv))
pol.train()
env = SyntheticSafetyVectorEnv(env_num=E, obs_dim=Do, act_dim=Da, episode_len=ep_len, seed=0)
buf = HipVectorReplayBuffer(pol.engine, 50000, E)
t0 = time.time()
for c in range(cycles):
    st = _rollout(pol, env, buf, noise=True)
    pol.pre_update_fn(stats_train={"cost": st["cost"]})
    for _ in range(upc):
        pol.update(B, buf)
    alpha = float(pol.engine.sac_get_params(0)[1]) if hasattr(pol.engine, "sac_get_params") else float("nan")
    print(c, round(st["reward"], 1), round(st["cost"], 1), round(pol.lag_optims[0].get_lag(), 3), round(alpha, 4), round(time.time() - t0), flush=True)
