#!/usr/bin/env python
"""Diagnosis (build container only, not a fixture): the UNMODIFIED reference SACLagrangian trained for 20 cycles x (10 episodes
of 100 steps, 200 updates of batch 256) on the synthetic env; prints cycle, reward, cost, lambda, alpha, seconds.  Its HIP twin
is tools/diag/longrun_sac_hip.py; both logs side by side: profiles/r02_longrun_sac.md.
    python tests/golden/longrun_ref_sac.py <seed>"""
import sys, os, json, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np, torch
import ref_shim
ref_shim.install()
from gen_golden import CaptureLogger, seed_all
from gen_golden_loop import rollout
from ref_shim import Batch, VectorReplayBuffer, ActorProb, Net, _Box
from fsrl_amd.env.synthetic import SyntheticSafetyVectorEnv
from fsrl.policy import SACLagrangian
from fsrl.utils.net.common import ActorCritic
from fsrl.utils.net.continuous import DoubleCritic
from torch import nn
torch.set_num_threads(4)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
obs_dim, act_dim, hidden, env_num, ep_len, cycles, B, upc = 8, 2, (64, 64), 10, 100, 20, 256, 200
seed_all(seed)
actor = ActorProb(Net((obs_dim, ), hidden_sizes=hidden), (act_dim, ), max_action=1.0, conditioned_sigma=True, unbounded=True)
critics = [DoubleCritic(Net((obs_dim, ), (act_dim, ), hidden_sizes=hidden, concat=True),
                        Net((obs_dim, ), (act_dim, ), hidden_sizes=hidden, concat=True)) for _ in range(2)]
for m in ActorCritic(actor, critics).modules():
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.orthogonal_(m.weight); torch.nn.init.zeros_(m.bias)
log_alpha = torch.zeros(1, requires_grad=True)
logger = CaptureLogger()
policy = SACLagrangian(actor=actor, critics=critics, actor_optim=torch.optim.Adam(actor.parameters(), lr=5e-4),
                       critic_optim=torch.optim.Adam(nn.ModuleList(critics).parameters(), lr=1e-3), logger=logger,
                       alpha=(-float(act_dim), log_alpha, torch.optim.Adam([log_alpha], lr=3e-4)), n_step=2,
                       cost_limit=20.0, observation_space=_Box(-np.inf, np.inf, (obs_dim, )), action_space=_Box(-1, 1, (act_dim, )))
policy.train()
env = SyntheticSafetyVectorEnv(env_num=env_num, obs_dim=obs_dim, act_dim=act_dim, episode_len=ep_len, seed=0)
buf = VectorReplayBuffer(50000, env_num)
t0 = time.time()
for c in range(cycles):
    st = rollout(policy, env, buf, noise=True)
    policy.pre_update_fn(stats_train={"cost": st["cost"]})
    for _ in range(upc):
        policy.update(B, buf)
    logger.rows.clear()
    print(c, round(st["reward"], 1), round(st["cost"], 1), round(policy.lag_optims[0].get_lag(), 3), round(float(policy._alpha), 4), round(time.time() - t0), flush=True)
