import json, random, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
from torch.distributions import Independent, Normal
from helpers import load_npz
from test_gpu_loop import _rollout, _Cap
from fsrl_amd.data import HipVectorReplayBuffer
from fsrl_amd.env import Box, SyntheticSafetyVectorEnv
from fsrl_amd.policy import CPO, TRPOLagrangian
from fsrl_amd.utils.net import ActorProb, Critic, Net
for kind in ("cpo", "trpo"):
    g = load_npz(f"loop_{kind}.npz"); cfg = json.loads(str(g["cfg_json"]))
    Do, Da, h, E = cfg["obs_dim"], cfg["act_dim"], tuple(cfg["hidden"]), cfg["env_num"]
    actor = ActorProb(Net((Do, ), hidden_sizes=h), (Da, ), max_action=1.0)
    critics = [Critic(Net((Do, ), hidden_sizes=h)) for _ in range(2)]
    log = _Cap()
    cls = CPO if kind == "cpo" else TRPOLagrangian
    pol = cls(actor, critics, torch.optim.Adam(torch.nn.ModuleList(critics).parameters(), lr=cfg["lr"]),
              lambda *l: Independent(Normal(*l), 1), logger=log, cost_limit=cfg["cost_limit"], optim_critic_iters=cfg["optim_critic_iters"],
              observation_space=Box(-np.inf, np.inf, (Do, )), action_space=Box(-1, 1, (Da, )), device=0, env_num=E,
              buffer_size=E * cfg["ep_len"] * 2, reference_rng=True)
    pol.engine.set_params(g["theta0"]); pol._pull_params(); pol.train()
    env = SyntheticSafetyVectorEnv(env_num=E, obs_dim=Do, act_dim=Da, episode_len=cfg["ep_len"], seed=cfg["seed"] + 11)
    buf = HipVectorReplayBuffer(pol.engine, E * cfg["ep_len"] * 2, E)
    seed = cfg["seed"] + 7
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    st = _rollout(pol, env, buf)
    pol.pre_update_fn(stats_train={"cost": st["cost"]})
    pol.update(0, buf, batch_size=99999, repeat=cfg["repeat"])
    rows = [r for r in log.rows if "update/gradient_steps" not in r]
    keys = [str(k) for k in g["stat_keys"]]
    per = len(rows) // cfg["repeat"]
    np.set_printoptions(linewidth=220, precision=5)
    for i in range(cfg["repeat"]):
        m = {}
        for r in rows[i * per:(i + 1) * per]: m.update(r)
        print(kind, "repeat", i, "got ", np.array([m[k] for k in keys]))
        print(kind, "repeat", i, "want", g["first_update_rows"][i])
    d = np.abs(pol.engine.get_params() - g["theta_after_first"])
    print(kind, "theta after first update: mean/max diff", d.mean(), d.max(), "ls evals", pol.engine.tr_linesearch_evals())
