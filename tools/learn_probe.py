import sys, tempfile
sys.path.insert(0, '.')
import numpy as np
from fsrl_amd.agent import PPOLagAgent, SACLagAgent, CPOAgent
from fsrl_amd.env import SyntheticSafetyVectorEnv
from fsrl_amd.utils import BaseLogger
for name, cls, kw, lk in (("ppol", PPOLagAgent, dict(), dict(repeat_per_collect=4, batch_size=256)),
                          ("cpo", CPOAgent, dict(), dict(repeat_per_collect=2, batch_size=99999)),
                          ("sacl", SACLagAgent, dict(buffer_size=50000), dict(update_per_step=0.2, batch_size=256))):
    env = SyntheticSafetyVectorEnv(env_num=10, episode_len=100, seed=0)
    test = SyntheticSafetyVectorEnv(env_num=2, episode_len=100, seed=5)
    agent = cls(env, BaseLogger(tempfile.mkdtemp(), name=name), cost_limit=20, device="cuda:0", seed=1, hidden_sizes=(64, 64),
                training_num=10, **kw)
    r0 = agent.evaluate(test, eval_episodes=4)
    hist = []
    for ep in range(1):
        out = agent.learn(env, None, epoch=20, episode_per_collect=10, step_per_epoch=2000, verbose=False, save_ckpt=False,
                          device_actor=True, **lk)
    r1 = agent.evaluate(test, eval_episodes=4)
    print(name, "before", [round(x, 1) for x in r0], "after", [round(x, 1) for x in r1], {k: round(v, 2) for k, v in out[1].items() if k in ("train/reward", "train/cost", "loss/lagrangian")})
