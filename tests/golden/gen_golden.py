#!/usr/bin/env python
"""Generate golden vectors from the UNMODIFIED reference code (build container only).

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz

The reference (`/root/reference/fsrl`) is imported under `ref_shim` (stand-ins for the
absent tianshou/numba/gymnasium).  What is recorded are *inputs and outputs only* (data);
no reference source is copied.  The GPU box never runs this script: it has no
`/root/reference`; it consumes the committed `.npz` files.

Vectors:
  G1 gae_*.npz        gae_return          fsrl/policy/base_policy.py:524-540
  G2 nstep_*.npz      nstep_return        fsrl/policy/base_policy.py:543-567
  G3 pid_trace.npz    LagrangianOptimizer fsrl/utils/optim_util.py:28-41
  G4 ppo_*.npz        PPOLagrangian.update (process_fn + learn), fsrl/policy/ppo_lag.py:134-257
  G8 state_dict_manifest.json   key order / shapes of policy.state_dict()
"""
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

from fsrl.policy import PPOLagrangian  # noqa: E402
from fsrl.policy.base_policy import gae_return, nstep_return  # noqa: E402
from fsrl.utils.net.common import ActorCritic  # noqa: E402
from fsrl.utils.optim_util import LagrangianOptimizer  # noqa: E402
from torch.distributions import Independent, Normal  # noqa: E402

from ref_shim import ActorProb, Critic, Net, VectorReplayBuffer, _Box  # noqa: E402


def seed_all(seed):
    # same three RNGs as fsrl/utils/exp_util.py:16-30
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


class CaptureLogger:
    """Duck-typed BaseLogger: keeps every store() call in order."""

    def __init__(self):
        self.rows = []
        self.msgs = []

    def store(self, tab=None, **kw):
        self.rows.append({(tab + "/" + k if tab else k): float(np.mean(v)) for k, v in kw.items()})

    def print(self, msg, *a, **k):
        self.msgs.append(str(msg))


# ----------------------------------------------------------------------------- G1
def gen_gae():
    rng = np.random.default_rng(1)
    out = {}
    cases = [1, 7, 300, 2048, 20000]
    for n in cases:
        v = rng.standard_normal(n).astype(np.float32)
        vn = rng.standard_normal(n).astype(np.float32)
        rew = rng.normal(0.5, 0.5, n)  # float64, like tianshou-stored rew
        end = rng.random(n) < (0.02 if n > 10 else 0.3)
        end[-1] = True  # unfinished tail is forced True (base_policy.py:409-411)
        for gamma, lam in ((0.99, 0.95), (1.0, 1.0), (0.9, 0.0)):
            adv = gae_return(v, vn, rew, end, gamma, lam)
            assert adv.dtype == np.float64
            key = f"n{n}_g{gamma}_l{lam}"
            out[key + "_adv"] = adv
        out[f"n{n}_v"], out[f"n{n}_vn"], out[f"n{n}_rew"], out[f"n{n}_end"] = v, vn, rew, end
    np.savez_compressed(os.path.join(HERE, "gae_cases.npz"), **out)
    print("G1 gae_cases.npz", len(out), "arrays")


# ----------------------------------------------------------------------------- G2
def gen_nstep():
    rng = np.random.default_rng(2)
    out = {}
    size = 500
    metric = rng.normal(0.3, 1.0, size)
    end_flag = rng.random(size) < 0.05
    end_flag[-1] = True
    # buffer.next semantics: i+1 unless end_flag[i]
    nxt = np.where(end_flag, np.arange(size), np.minimum(np.arange(size) + 1, size - 1))
    bsz = 64
    indice = rng.integers(0, size, bsz)
    out["metric"], out["end_flag"], out["indice"] = metric, end_flag, indice
    for n_step in (1, 2, 3, 5):
        idx = [indice]
        for _ in range(n_step - 1):
            idx.append(nxt[idx[-1]])
        idx = np.stack(idx)
        target_q = rng.standard_normal((bsz, 1)).astype(np.float32)
        for gamma in (0.99, 0.9):
            r = nstep_return(metric, end_flag, target_q, idx, gamma, n_step)
            out[f"n{n_step}_g{gamma}_ret"] = r
        out[f"n{n_step}_indices"] = idx
        out[f"n{n_step}_target_q"] = target_q
    np.savez_compressed(os.path.join(HERE, "nstep_cases.npz"), **out)
    print("G2 nstep_cases.npz", len(out), "arrays")


# ----------------------------------------------------------------------------- G3
def gen_pid():
    rng = np.random.default_rng(3)
    costs = np.concatenate([rng.uniform(0, 40, 25), rng.uniform(0, 8, 15), rng.uniform(5, 15, 10)])
    out = {"costs": costs}
    for name, pid, limit in (("default", (0.05, 0.0005, 0.1), 10.0), ("sgd", (0.0, 0.01, 0.0), 5.0)):
        opt = LagrangianOptimizer(pid)
        lag, integ, old = [], [], []
        for c in costs:
            opt.step(c, limit)
            lag.append(opt.get_lag()); integ.append(opt.error_integral); old.append(opt.error_old)
        out[name + "_pid"] = np.array(pid)
        out[name + "_limit"] = np.array(limit)
        out[name + "_lag"] = np.array(lag, np.float64)
        out[name + "_integral"] = np.array(integ, np.float64)
        out[name + "_error_old"] = np.array(old, np.float64)
    np.savez_compressed(os.path.join(HERE, "pid_trace.npz"), **out)
    print("G3 pid_trace.npz")


# ----------------------------------------------------------------------------- G4
def build_ppo(obs_dim, act_dim, hidden, seed, max_action=1.0, last_layer_scale=False,
              logger=None, lr=5e-4, unbounded=False, **ppo_kw):
    """Mirror of fsrl/agent/ppo_lag_agent.py:127-200 using the shim's tianshou nets."""
    seed_all(seed)
    obs_space = _Box(-np.inf, np.inf, (obs_dim, ))
    act_space = _Box(-max_action, max_action, (act_dim, ))
    net = Net((obs_dim, ), hidden_sizes=hidden)
    actor = ActorProb(net, (act_dim, ), max_action=max_action, unbounded=unbounded)
    critic = [Critic(Net((obs_dim, ), hidden_sizes=hidden)) for _ in range(2)]
    torch.nn.init.constant_(actor.sigma_param, -0.5)
    actor_critic = ActorCritic(actor, critic)
    for m in actor_critic.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    if last_layer_scale:
        for m in actor.mu.modules():
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.zeros_(m.bias)
                m.weight.data.copy_(0.01 * m.weight.data)
    optim = torch.optim.Adam(actor_critic.parameters(), lr=lr)

    def dist(*logits):
        return Independent(Normal(*logits), 1)

    policy = PPOLagrangian(actor, critic, optim, dist, logger=logger,
                           observation_space=obs_space, action_space=act_space, **ppo_kw)
    return policy, actor_critic, optim


def fill_buffer(rng, env_num, ep_lens_per_env, obs_dim, act_dim, term_prob=0.3,
                buffer_size=100000, cost_prob=0.1):
    """Synthetic rollouts pushed in lock-step like FastCollector does (fast_collector.py:333).
    ep_lens_per_env[e] = list of episode lengths; the LAST episode of env e may be left
    unfinished by giving a negative length (-k => k steps, no done)."""
    buf = VectorReplayBuffer(buffer_size, env_num)
    plan = []
    for e in range(env_num):
        steps = []
        for L in ep_lens_per_env[e]:
            unfinished = L < 0
            L = abs(L)
            terminated_end = rng.random() < term_prob
            for t in range(L):
                last = (t == L - 1) and not unfinished
                steps.append((last and terminated_end, last and not terminated_end))
        plan.append(steps)
    T = max(len(p) for p in plan)
    cur_obs = rng.standard_normal((env_num, obs_dim)).astype(np.float32)
    for t in range(T):
        ids = [e for e in range(env_num) if t < len(plan[e])]
        k = len(ids)
        obs = cur_obs[ids]
        nxt = rng.standard_normal((k, obs_dim)).astype(np.float32)
        act = (0.3 * rng.standard_normal((k, act_dim))).astype(np.float32)
        rew = rng.normal(0.5, 0.5, k)
        cost = (rng.random(k) < cost_prob).astype(np.float64)      # same draws for every cost_prob
        term = np.array([plan[e][t][0] for e in ids])
        trunc = np.array([plan[e][t][1] for e in ids])
        done = term | trunc
        buf.add({"obs": obs, "act": act, "rew": rew, "terminated": term, "truncated": trunc,
                 "done": done, "obs_next": nxt, "info.cost": cost}, ids)
        cur_obs[ids] = nxt
        for j, e in enumerate(ids):  # episode reset => fresh obs
            if done[j]:
                cur_obs[e] = rng.standard_normal(obs_dim).astype(np.float32)
    return buf


def flat_params(module):
    return torch.cat([p.detach().reshape(-1) for p in module.parameters()]).numpy().copy()


def gen_ppo_case(name, obs_dim, act_dim, hidden, env_num, ep_lens, batch_size, repeat,
                 seed, cost_stat=25.0, cost_limit=10.0, ret_rms0=None, **ppo_kw):
    logger = CaptureLogger()
    policy, actor_critic, optim = build_ppo(obs_dim, act_dim, hidden, seed, logger=logger,
                                            cost_limit=cost_limit, **ppo_kw)
    policy.train()
    rng = np.random.default_rng(seed + 1000)
    buf = fill_buffer(rng, env_num, ep_lens, obs_dim, act_dim)
    out = {}
    theta0 = flat_params(actor_critic)
    out["theta0"] = theta0
    # ---- what the trainer does (fsrl/trainer/onpolicy.py:92-109)
    policy.pre_update_fn(stats_train={"cost": cost_stat})
    out["lagrangian"] = np.array([o.get_lag() for o in policy.lag_optims], np.float64)

    # the batch exactly as update() sees it (base_policy.py:348), for the store test
    batch, indices = buf.sample(0)
    out["indices"] = indices
    for k in ("obs", "act", "rew", "terminated", "truncated", "obs_next"):
        out["buf_" + k] = getattr(batch, k)
    out["buf_cost"] = batch.info.cost
    out["unfinished_index"] = buf.unfinished_index()
    out["env_rows"] = np.array([len(b) for b in buf.buffers])

    if ret_rms0 is not None:      # running return statistics as an earlier update would have left them (fixture input state)
        for rms, (m, v, n) in zip(policy.ret_rms, ret_rms0):
            rms.mean, rms.var, rms.count = float(m), float(v), float(n)
    out["ret_rms0"] = np.array([[r.mean, r.var, r.count] for r in policy.ret_rms], np.float64)
    # process_fn outputs: record them (with reward_normalization the call also updates ret_rms: put the state back)
    import copy
    rms_keep = copy.deepcopy(policy.ret_rms)
    pb = policy.process_fn(batch, buf, indices)
    policy.ret_rms = rms_keep
    out["values"], out["rets"], out["advs"] = (pb.values.numpy().copy(), pb.rets.numpy().copy(),
                                               pb.advs.numpy().copy())
    out["logp_old"] = pb.logp_old.numpy().copy()

    # record the permutations Batch.split draws and a param snapshot after optimizer step 1
    perms = []
    orig_perm = np.random.permutation

    def rec_perm(n):
        p = orig_perm(n)
        perms.append(np.asarray(p).copy())
        return p

    snaps = {}
    orig_step = optim.step
    counter = {"n": 0}

    def rec_step(*a, **k):
        r = orig_step(*a, **k)
        counter["n"] += 1
        if counter["n"] == 1:
            snaps["theta_step1"] = flat_params(actor_critic)
        return r

    optim.step = rec_step
    np.random.permutation = rec_perm
    try:
        seed_all(seed + 7)  # fixes the shuffles; recorded anyway
        policy.update(0, buf, batch_size=batch_size, repeat=repeat)
    finally:
        np.random.permutation = orig_perm
    # shuffled perms only: process_fn/compute_gae use shuffle=False (no permutation call)
    out["perms"] = np.stack(perms)
    out["theta_final"] = flat_params(actor_critic)
    out.update(snaps)
    # per-minibatch stats: logger.store is called 3x per minibatch (ppo_lag.py:245-247)
    rows = [r for r in logger.rows if "update/gradient_steps" not in r]
    assert len(rows) % 3 == 0
    keys = ["loss/rescaling", "loss/lagrangian", "loss/actor_safety", "loss/actor_rew",
            "loss/actor_total", "loss/kl", "loss/vf0", "loss/vf1", "loss/vf_total",
            "loss/total", "loss/entropy"]
    stats = []
    for i in range(0, len(rows), 3):
        merged = {}
        for r in rows[i:i + 3]:
            merged.update(r)
        stats.append([merged[k] for k in keys])
    out["stats"] = np.array(stats, np.float64)
    out["stats_keys"] = np.array(keys)
    out["ret_rms_final"] = np.array([[r.mean, r.var, r.count] for r in policy.ret_rms], np.float64)
    out["gradient_steps"] = np.array(policy.gradient_steps)
    out["early_stop_msgs"] = np.array(len(logger.msgs))
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num,
               batch_size=batch_size, repeat=repeat, seed=seed, cost_stat=cost_stat,
               cost_limit=cost_limit, max_action=1.0)
    cfg.update({k: v for k, v in ppo_kw.items()})
    defaults = dict(target_kl=0.02, vf_coef=0.25, max_grad_norm=None, gae_lambda=0.95,
                    eps_clip=0.2, dual_clip=None, gamma=0.99, lr=5e-4,
                    advantage_normalization=True, lagrangian_pid=(0.05, 0.0005, 0.1),
                    rescaling=True, use_lagrangian=True)
    for k, v in defaults.items():
        cfg.setdefault(k, v)
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"ppo_{name}.npz"), **out)
    print(f"G4 ppo_{name}.npz  N={len(indices)} steps={len(stats)} "
          f"grad_steps={policy.gradient_steps} early_stop={len(logger.msgs)}")
    return policy


def gen_ppo():
    # tiny: ragged episodes, a terminated + truncated mix, unfinished tails, merge_last path
    gen_ppo_case("tiny", obs_dim=5, act_dim=3, hidden=(64, 64), env_num=3,
                 ep_lens=[[40, 37, -13], [55, 50], [30, 30, 30, -7]], batch_size=64, repeat=2,
                 seed=0, max_grad_norm=0.5, target_kl=1e9)
    # C1 shape (128x128, obs 8, act 2), reduced N for fixture size; clip on (ppol_cfg.py:21)
    gen_ppo_case("c1", obs_dim=8, act_dim=2, hidden=(128, 128), env_num=4,
                 ep_lens=[[300, 300], [300, 300], [300, 300], [300, -150]], batch_size=256,
                 repeat=4, seed=1, max_grad_norm=0.5, target_kl=1e9)
    # C2 shape (256x256), agent-default no grad clip (ppo_lag_agent.py:98)
    gen_ppo_case("c2", obs_dim=8, act_dim=2, hidden=(256, 256), env_num=4,
                 ep_lens=[[300, 212], [300, 212], [300, 212], [300, 212]], batch_size=256,
                 repeat=2, seed=2, max_grad_norm=None, target_kl=1e9)
    # KL early stop fires (ppo_lag.py:251-255): large lr => KL blows past 1.5*target_kl
    gen_ppo_case("earlystop", obs_dim=8, act_dim=2, hidden=(64, 64), env_num=2,
                 ep_lens=[[200, 100], [150, 150]], batch_size=64, repeat=6, seed=3,
                 max_grad_norm=0.5, target_kl=0.005, lr=3e-3)
    gen_ppo_recompute()
    # lagrangian off / no adv-norm / dual clip branch
    gen_ppo_case("dualclip", obs_dim=6, act_dim=2, hidden=(64, 64), env_num=2,
                 ep_lens=[[100, 100], [100, -60]], batch_size=128, repeat=2, seed=4,
                 max_grad_norm=None, target_kl=1e9, dual_clip=3.0,
                 advantage_normalization=False)


def gen_ppo_bigbatch():
    # a minibatch whose 16-row tiles exceed the CU count of an MI355X (2 048 rows x 3 networks = 384 tiles > 256): the HIP path runs
    # 32-row tiles there (r6, ppo_fwd_bwd_tall_kernel); 4 600 rows = one minibatch of 2 048 + the merged rest of 2 552 per pass
    gen_ppo_case("bigbatch", obs_dim=8, act_dim=2, hidden=(256, 256), env_num=4,
                 ep_lens=[[600, 550], [600, 550], [600, 550], [600, 550]], batch_size=2048,
                 repeat=2, seed=6, max_grad_norm=0.5, target_kl=1e9)


def gen_ppo_recompute():
    # recompute_advantage (ppo_lag.py:218-221): GAE from the CURRENT critics before passes 2 and 3
    gen_ppo_case("recompute", obs_dim=6, act_dim=2, hidden=(64, 64), env_num=3,
                 ep_lens=[[70, 60, -25], [80, 75], [50, 50, 50]], batch_size=64, repeat=3, seed=5,
                 max_grad_norm=0.5, target_kl=1e9, recompute_advantage=True)


def gen_ppo_options():
    # reward_normalization + value_clip (base_policy.py:430-444, ppo_lag.py:158-164) from a non-trivial running state
    rms0 = [(2.9, 6.5, 1500.0), (0.35, 0.8, 1500.0)]
    gen_ppo_case("rewnorm", obs_dim=6, act_dim=2, hidden=(64, 64), env_num=3,
                 ep_lens=[[70, 60, -25], [80, 75], [50, 50, 50]], batch_size=64, repeat=3, seed=6,
                 max_grad_norm=0.5, target_kl=1e9, reward_normalization=True, value_clip=True, ret_rms0=rms0,
                 eps_clip=0.05, lr=2e-3)      # small clip range + larger steps: rows on both sides of the value clip
    # the same from the initial state (mean 0, var 1, count 0), value clip off
    gen_ppo_case("rewnorm_first", obs_dim=6, act_dim=2, hidden=(64, 64), env_num=3,
                 ep_lens=[[70, 60, -25], [80, 75], [50, 50, 50]], batch_size=64, repeat=2, seed=6,
                 max_grad_norm=0.5, target_kl=1e9, reward_normalization=True)
    # recompute_advantage on top: ret_rms is updated once more before passes 2 and 3, value_clip uses the fresh values
    gen_ppo_case("rewnorm_recompute", obs_dim=6, act_dim=2, hidden=(64, 64), env_num=3,
                 ep_lens=[[70, 60, -25], [80, 75], [50, 50, 50]], batch_size=64, repeat=3, seed=8,
                 max_grad_norm=0.5, target_kl=1e9, reward_normalization=True, value_clip=True, ret_rms0=rms0,
                 recompute_advantage=True)
    # unbounded actor (ActorProb unbounded=True: the mean is the head's raw output)
    gen_ppo_case("unbounded", obs_dim=6, act_dim=2, hidden=(64, 64), env_num=3,
                 ep_lens=[[70, 60, -25], [80, 75], [50, 50, 50]], batch_size=64, repeat=3, seed=9,
                 max_grad_norm=0.5, target_kl=1e9, unbounded=True)


def gen_ppo_widths():
    # hidden_sizes of two DIFFERENT widths, neither one of the kernels' 64 / 128 / 256 (fsrl/agent/ppo_lag_agent.py:91,136:
    # `hidden_sizes: Tuple[int, ...]`): the HIP path zero-pads them
    gen_ppo_case("widths", obs_dim=7, act_dim=3, hidden=(96, 40), env_num=3,
                 ep_lens=[[60, 45, -11], [70, 50], [40, 40, -20]], batch_size=64, repeat=3, seed=41,
                 max_grad_norm=0.5, target_kl=1e9)
    gen_ppo_case("widths_wide", obs_dim=8, act_dim=2, hidden=(100, 200), env_num=2,
                 ep_lens=[[90, 70], [100, -50]], batch_size=128, repeat=2, seed=42,
                 max_grad_norm=None, target_kl=1e9)


def gen_ppo_depths():
    # hidden_sizes that are NOT two layers of at most 256 units (fsrl/agent/ppo_lag_agent.py:91,136: any tuple): the HIP library
    # runs these through its layer-by-layer kernels (include/fsrl_hip.h fsrl_config.n_hidden)
    gen_ppo_case("deep3", obs_dim=7, act_dim=3, hidden=(64, 48, 32), env_num=3,
                 ep_lens=[[60, 45, -11], [70, 50], [40, 40, -20]], batch_size=64, repeat=3, seed=51,
                 max_grad_norm=0.5, target_kl=1e9)
    gen_ppo_case("wide", obs_dim=8, act_dim=2, hidden=(300, 260), env_num=2,
                 ep_lens=[[90, 70], [100, -50]], batch_size=128, repeat=2, seed=52,
                 max_grad_norm=None, target_kl=1e9)
    gen_ppo_case("one_layer", obs_dim=6, act_dim=2, hidden=(96, ), env_num=3,
                 ep_lens=[[70, 60, -25], [80, 75], [50, 50, 50]], batch_size=64, repeat=3, seed=53,
                 max_grad_norm=0.5, target_kl=0.01, lr=3e-3)          # the KL early stop fires
    rms0 = [(2.9, 6.5, 1500.0), (0.35, 0.8, 1500.0)]
    gen_ppo_case("deep4_options", obs_dim=6, act_dim=2, hidden=(40, 72, 72, 24), env_num=3,
                 ep_lens=[[70, 60, -25], [80, 75], [50, 50, 50]], batch_size=64, repeat=3, seed=54,
                 max_grad_norm=0.5, target_kl=1e9, reward_normalization=True, value_clip=True, ret_rms0=rms0,
                 recompute_advantage=True, dual_clip=3.0, eps_clip=0.05, lr=2e-3)


def gen_ppo_full_case(name, env_num, ep_lens, seed=2, hidden=(256, 256), batch_size=256, repeat=4, theta0_from=None,
                      target_kl=1e9, lr=5e-4, keep_pass1=True):
    """BASELINE-size fixtures (configs[1]: 20 envs x 1000 rows; configs[4] per rank: 32 envs x 625 rows), 256x256, grad clip
    0.5 (ppol_cfg.py:21), repeat 4 = 312 steps of the UNMODIFIED PPOLagrangian.update.  20 000 rows of random floats do not
    compress, so the rollout is NOT stored: tests/helpers.synth_rollout(seed) regenerates it (checksum stored), and the file
    keeps theta0, the recorded permutations (uint16), advs / logp_old of process_fn, stats[312, 11], theta after pass 1 and
    after pass 4."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import rollout_checksum, synth_rollout
    obs_dim, act_dim = 8, 2
    logger = CaptureLogger()
    kw = dict(max_grad_norm=0.5, target_kl=target_kl, lr=lr)
    policy, actor_critic, optim = build_ppo(obs_dim, act_dim, hidden, seed, logger=logger, cost_limit=10.0, **kw)
    policy.train()
    steps = synth_rollout(seed + 1000, env_num, ep_lens, obs_dim, act_dim)
    buf = VectorReplayBuffer(100000, env_num)
    for ids, obs, act, rew, cost, term, trunc, nxt in steps:
        buf.add({"obs": obs, "act": act, "rew": rew, "terminated": term, "truncated": trunc, "done": term | trunc,
                 "obs_next": nxt, "info.cost": cost}, ids)
    out = {"rollout_seed": np.array(seed + 1000), "rollout_checksum": rollout_checksum(steps),
           "ep_lens": np.array(ep_lens[0]), "env_num": np.array(env_num)}
    assert all(list(e) == list(ep_lens[0]) for e in ep_lens)
    theta0 = flat_params(actor_critic)
    if theta0_from is None:
        out["theta0"] = theta0
    else:
        assert np.array_equal(theta0, np.load(os.path.join(HERE, theta0_from))["theta0"])
        out["theta0_from"] = np.array(theta0_from)
    policy.pre_update_fn(stats_train={"cost": 25.0})
    out["lagrangian"] = np.array([o.get_lag() for o in policy.lag_optims], np.float64)
    batch, indices = buf.sample(0)
    out["n_rows"] = np.array(len(indices))
    pb = policy.process_fn(batch, buf, indices)
    out["advs"], out["logp_old"] = pb.advs.numpy().copy(), pb.logp_old.numpy().copy()
    perms, snaps, counter = [], {}, {"n": 0}
    orig_perm, orig_step = np.random.permutation, optim.step
    n_per_pass = max(1, len(indices) // batch_size)          # Batch.split(merge_last=True): a remainder joins the last chunk

    def rec_perm(n):
        p = orig_perm(n)
        perms.append(np.asarray(p).copy())
        return p

    def rec_step(*a, **k):
        r = orig_step(*a, **k)
        counter["n"] += 1
        if counter["n"] == n_per_pass:
            snaps["theta_pass1"] = flat_params(actor_critic)
        return r

    optim.step = rec_step
    np.random.permutation = rec_perm
    try:
        seed_all(seed + 7)
        policy.update(0, buf, batch_size=batch_size, repeat=repeat)
    finally:
        np.random.permutation = orig_perm
    assert len(perms) <= repeat and (len(perms) == repeat or target_kl < 1e8) and max(p.max() for p in perms) < 65536
    out["perms"] = np.stack(perms).astype(np.uint16)
    out["theta_final"] = flat_params(actor_critic)
    if keep_pass1:
        out.update(snaps)
    out["passes_run"] = np.array(len(perms))           # < repeat: the KL early stop fired (ppo_lag.py:251-255)
    out["early_stop_msgs"] = np.array(len(logger.msgs))
    rows = [r for r in logger.rows if "update/gradient_steps" not in r]
    keys = ["loss/rescaling", "loss/lagrangian", "loss/actor_safety", "loss/actor_rew", "loss/actor_total", "loss/kl",
            "loss/vf0", "loss/vf1", "loss/vf_total", "loss/total", "loss/entropy"]
    stats = []
    for i in range(0, len(rows), 3):
        merged = {}
        for r in rows[i:i + 3]:
            merged.update(r)
        stats.append([merged[k] for k in keys])
    out["stats"] = np.array(stats, np.float64)
    assert out["stats"].shape == (n_per_pass * len(perms), 11) and ("theta_pass1" in out or not keep_pass1)
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, batch_size=batch_size, repeat=repeat,
               seed=seed, cost_stat=25.0, cost_limit=10.0, max_action=1.0, target_kl=target_kl, vf_coef=0.25, max_grad_norm=0.5,
               gae_lambda=0.95, eps_clip=0.2, dual_clip=None, gamma=0.99, lr=lr, advantage_normalization=True,
               lagrangian_pid=(0.05, 0.0005, 0.1), rescaling=True, use_lagrangian=True)
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"ppo_{name}.npz"), **out)
    print(f"G4 ppo_{name}.npz  N={len(indices)} steps={len(stats)} size={os.path.getsize(os.path.join(HERE, f'ppo_{name}.npz')) / 1e6:.2f} MB")


LR_KLSTOP = float(os.environ.get("LR_KLSTOP", "1.5e-4"))


def gen_ppo_full():
    # configs[1]: 20 envs x 1000 rows, episodes of 250 steps (SURVEY 8d "N = 20 000 variant")
    gen_ppo_full_case("c2full", 20, [[250, 250, 250, 250]] * 20)
    # configs[4], one rank: 32 envs x 625 rows (two episodes of 250 and an unfinished tail of 125 each), same theta0
    gen_ppo_full_case("c5rank", 32, [[250, 250, -125]] * 32, theta0_from="ppo_c2full.npz")


def gen_ppo_full_c0():
    # configs[0] at its full size: 4 envs x 5000 rows (twenty episodes of 250 each), 128x128 -- the reference's own CPU-runnable case
    gen_ppo_full_case("c0full", 4, [[250] * 20] * 4, hidden=(128, 128))


def gen_ppo_full_klstop():
    # configs[1] with the reference's default target_kl = 0.02 (ppo_lag_agent.py:95) and a learning rate at which the pass-mean
    # KL crosses 1.5 x target_kl: the early stop (ppo_lag.py:251-255) fires at full size
    gen_ppo_full_case("c2full_klstop", 20, [[250, 250, 250, 250]] * 20, theta0_from="ppo_c2full.npz", target_kl=0.02, lr=LR_KLSTOP,
                      keep_pass1=False)


def gen_manifest():
    policy, _, _ = build_ppo(8, 2, (128, 128), 0, logger=CaptureLogger(), cost_limit=10.0)
    sd = policy.state_dict()
    man = [[k, list(v.shape) if torch.is_tensor(v) else None] for k, v in sd.items()]
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump({"ppo_lag_128x128_obs8_act2": man}, f, indent=0)
    print("G8 state_dict_manifest.json", len(man), "keys")


if __name__ == "__main__":
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["gae", "nstep", "pid", "ppo", "manifest"]
    for w in which:
        {"gae": gen_gae, "nstep": gen_nstep, "pid": gen_pid, "ppo": gen_ppo, "recompute": gen_ppo_recompute, "bigbatch": gen_ppo_bigbatch, "options": gen_ppo_options,
         "widths": gen_ppo_widths, "depths": gen_ppo_depths, "full": gen_ppo_full, "full_c0": gen_ppo_full_c0, "full_klstop": gen_ppo_full_klstop, "manifest": gen_manifest}[w]()
