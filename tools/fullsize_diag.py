"""Print HIP-vs-f64 and f32-oracle-vs-f64 deviations for the full-size single-pass update."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.ppo_lag import OnPolicyData, PPOLagConfig, PPOLagOracle
from fsrl_amd.engine import Engine, EngineConfig
rng = np.random.default_rng(11)
env_num, ep, n_ep = 20, 300, int(os.environ.get("NEP", 67))
HID = int(os.environ.get("HID", 256)); CLIP = float(os.environ.get("CLIP", 0.5)) or None
LR = float(os.environ.get("LR", 5e-4)); NORM = int(os.environ.get("NORM", 1))
print("HID", HID, "CLIP", CLIP, "LR", LR, "NEP", n_ep, "NORM", NORM)
eng = Engine(EngineConfig(obs_dim=8, act_dim=2, hidden=HID, env_num=env_num, max_grad_norm=CLIP, target_kl=None, lr=LR, norm_adv=bool(NORM)))
ocfg = PPOLagConfig(obs_dim=8, act_dim=2, hidden=(HID, HID), max_grad_norm=CLIP, target_kl=1e9, lr=LR, advantage_normalization=bool(NORM))
o = PPOLagOracle(ocfg); o64 = PPOLagOracle(ocfg, dtype=torch.float64)
torch.manual_seed(5)
theta = (0.1 * torch.randn(o.n_params)).numpy()
o.set_params(theta); o64.set_params(theta); eng.set_params(theta)
per_env = [n_ep // env_num + (1 if e < n_ep % env_num else 0) for e in range(env_num)]
cols = {k: [] for k in ("obs", "act", "rew", "cost", "term", "trunc", "obs_next")}
for e in range(env_num):
    T = per_env[e] * ep
    obs = rng.standard_normal((T + 1, 8)).astype(np.float32)
    act = (0.3 * rng.standard_normal((T, 2))).astype(np.float32)
    rew = rng.normal(0.5, 0.5, T); cost = (rng.random(T) < 0.1).astype(np.float64)
    trunc = np.zeros(T, bool); trunc[ep - 1::ep] = True; term = np.zeros(T, bool)
    for t in range(T):
        eng.push([e], obs[t:t+1], act[t:t+1], rew[t:t+1], cost[t:t+1], term[t:t+1], trunc[t:t+1], obs[t+1:t+2])
    for k, v in zip(cols, (obs[:-1], act, rew, cost, term, trunc, obs[1:])): cols[k].append(v)
cat = {k: np.concatenate(v) for k, v in cols.items()}
data = OnPolicyData(obs=cat["obs"], act=cat["act"], rew=cat["rew"], cost=cat["cost"], terminated=cat["term"],
                    truncated=cat["trunc"], obs_next=cat["obs_next"], end_flag=cat["term"] | cat["trunc"])
lag = np.array([0.75]); perm = rng.permutation(len(data)); resc = 1/1.75
torch.set_num_threads(4)
pb, ostats, _ = o.update(data, lag, resc, 256, 1, perms=[perm])
_, xstats, _ = o64.update(data, lag, resc, 256, 1, perms=[perm])
stats, _ = eng.ppo_update(lag, resc, 256, 1, perms=[perm])
np.set_printoptions(precision=3, linewidth=200)
print("scale      ", np.abs(xstats).max(0))
print("hip-vs-f64 ", np.abs(stats - xstats).max(0))
print("f32-vs-f64 ", np.abs(ostats - xstats).max(0))
print("hip-vs-f32 ", np.abs(stats - ostats).max(0))
print("grad-norm col? total loss per step hip-f64:", np.abs(stats[:, 9] - xstats[:, 9])[:12])
for s in (0, 1, 5, 20, 50, len(stats) - 1):
    print("step", s, "hip-f64", np.abs(stats[s] - xstats[s]).max(), "f32-f64", np.abs(ostats[s] - xstats[s]).max())
print("theta hip-f64", np.abs(eng.get_params() - o64.get_params()).max(), "f32-f64", np.abs(o.get_params() - o64.get_params()).max())
