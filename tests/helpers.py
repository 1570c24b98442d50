"""Shared helpers for the parity tests (golden-case loading)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def ppo_case(name):
    """-> (cfg dict, golden arrays) for tests/golden/ppo_<name>.npz"""
    g = load_npz(f"ppo_{name}.npz")
    cfg = json.loads(str(g["cfg_json"]))
    return cfg, g


def end_flag_of(g):
    """done | unfinished tail, in sample(0) order (fsrl/policy/base_policy.py:409-411)."""
    end = np.logical_or(g["buf_terminated"], g["buf_truncated"])
    end[np.isin(g["indices"], g["unfinished_index"])] = True
    return end


def oracle_cfg_and_data(cfg, g):
    from oracle.ppo_lag import OnPolicyData, PPOLagConfig
    ocfg = PPOLagConfig(obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"], hidden=tuple(cfg["hidden"]),
                        max_action=cfg["max_action"], gamma=cfg["gamma"],
                        gae_lambda=cfg["gae_lambda"], eps_clip=cfg["eps_clip"],
                        dual_clip=cfg["dual_clip"], vf_coef=cfg["vf_coef"],
                        max_grad_norm=cfg["max_grad_norm"], target_kl=cfg["target_kl"],
                        advantage_normalization=cfg["advantage_normalization"],
                        use_lagrangian=cfg["use_lagrangian"], lr=cfg["lr"],
                        recompute_advantage=bool(cfg.get("recompute_advantage", False)),
                        unbounded=bool(cfg.get("unbounded", False)),
                        reward_normalization=bool(cfg.get("reward_normalization", False)),
                        value_clip=bool(cfg.get("value_clip", False)))
    data = OnPolicyData(obs=g["buf_obs"], act=g["buf_act"], rew=g["buf_rew"], cost=g["buf_cost"],
                        terminated=g["buf_terminated"], truncated=g["buf_truncated"],
                        obs_next=g["buf_obs_next"], end_flag=end_flag_of(g))
    return ocfg, data


def synth_rollout(seed, env_num, ep_lens_per_env, obs_dim, act_dim, term_prob=0.3, cost_prob=0.1):
    """Seeded synthetic rollouts as lock-step vector steps [(ids, obs, act, rew, cost, term, trunc, obs_next)], the way
    FastCollector pushes them (fast_collector.py:333).  ep_lens_per_env[e] = episode lengths of env e; a negative last entry
    -k = k steps left unfinished.  The full-size fixtures (tests/golden/gen_golden.py `full`) store only the seed and a
    checksum of what this function returns; the generator script and the GPU tests both call it."""
    rng = np.random.default_rng(seed)
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
    cur = rng.standard_normal((env_num, obs_dim)).astype(np.float32)
    out = []
    for t in range(T):
        ids = [e for e in range(env_num) if t < len(plan[e])]
        k = len(ids)
        obs = cur[ids].copy()
        nxt = rng.standard_normal((k, obs_dim)).astype(np.float32)
        act = (0.3 * rng.standard_normal((k, act_dim))).astype(np.float32)
        rew = rng.normal(0.5, 0.5, k)
        cost = (rng.random(k) < cost_prob).astype(np.float64)
        term = np.array([plan[e][t][0] for e in ids])
        trunc = np.array([plan[e][t][1] for e in ids])
        out.append((ids, obs, act, rew, cost, term, trunc, nxt))
        cur[ids] = nxt
        for j, e in enumerate(ids):
            if term[j] or trunc[j]:
                cur[e] = rng.standard_normal(obs_dim).astype(np.float32)
    return out


def rollout_checksum(steps):
    """float64 sums of every column of synth_rollout's output: guards the fixture against a drift of numpy's generators"""
    return np.array([sum(float(np.asarray(s[c], np.float64).sum()) for s in steps) for c in range(1, 8)])


def rollout_env_major(steps, env_num):
    """the rows in sample(0) order (env-major, time-ordered inside each env) + end_flag = done | unfinished tail"""
    cols = [[[] for _ in range(env_num)] for _ in range(7)]
    for ids, *rest in steps:
        for j, e in enumerate(ids):
            for c, a in enumerate(rest):
                cols[c][e].append(a[j])
    cat = [np.concatenate([np.asarray(cols[c][e]) for e in range(env_num)]) for c in range(7)]
    obs, act, rew, cost, term, trunc, nxt = cat
    end = term | trunc
    off = np.cumsum([len(cols[0][e]) for e in range(env_num)])
    end[off - 1] = True                       # unfinished tails (base_policy.py:409-411); finished ones are set already
    return dict(obs=obs, act=act, rew=rew, cost=cost, terminated=term, truncated=trunc, obs_next=nxt, end_flag=end)


def ppo_full_case(name):
    """tests/golden/ppo_{c2full,c5rank}.npz: BASELINE-size PPO-Lagrangian updates of the unmodified reference (256x256, clip 0.5,
    N = 20 000, 4 passes).  -> (cfg, golden arrays incl. theta0 and int64 perms, the regenerated rollout steps)"""
    cfg, g = ppo_case(name)
    if "theta0" not in g:
        g["theta0"] = load_npz(str(g["theta0_from"]))["theta0"]
    g["perms"] = g["perms"].astype(np.int64)
    env_num = int(g["env_num"])
    steps = synth_rollout(int(g["rollout_seed"]), env_num, [list(g["ep_lens"])] * env_num, cfg["obs_dim"], cfg["act_dim"])
    assert np.array_equal(rollout_checksum(steps), g["rollout_checksum"]), "numpy's generators no longer reproduce the fixture's rollout"
    return cfg, g, steps


def synth_theta(seed, shapes, bias_scale=0.05, head_scale=1.0):
    """A seeded parameter vector for the full-size fixtures (NOT stored in them: regenerated here, checksum in the fixture):
    `shapes` = the shapes of a module's parameters() in order; matrices ~ N(0, 1 / fan_in), vectors ~ bias_scale * N(0, 1);
    a (Da, 1) tensor is the on-policy actor's sigma_param (= -0.5, ppo_lag_agent.py:147).  head_scale: factor on the head
    matrices (<= 16 output rows): the multi-seed learning-curve fixtures start from small mean actions (0.1), like the agents'
    `last_layer_scale` option (ppo_lag_agent.py:155-161), instead of a saturated tanh."""
    rng = np.random.default_rng(seed)
    parts = []
    for sh in shapes:
        sh = tuple(int(x) for x in sh)
        if len(sh) == 2 and sh[1] == 1 and sh[0] <= 16:
            parts.append(np.full(sh[0], -0.5, np.float32))
        elif len(sh) == 2:
            w = (rng.standard_normal(sh) / np.sqrt(sh[1])).astype(np.float32)
            if sh[0] <= 16 and head_scale != 1.0:
                w = (w * np.float32(head_scale)).astype(np.float32)
            parts.append(w.reshape(-1))
        else:
            parts.append((bias_scale * rng.standard_normal(sh)).astype(np.float32).reshape(-1))
    return np.concatenate(parts)


def theta_checksum(theta):
    th = np.asarray(theta, np.float64)
    return np.array([th.sum(), np.abs(th).sum(), (th * np.arange(1, th.size + 1)).sum()])


def full_rollout_arrays(g, obs_dim, act_dim):
    """the regenerated rollout of a full-size fixture as the arrays the small fixtures store: env-major rows (sample(0) order)
    under the small fixtures' keys (buf_* and st_*), env_rows, slots, indices, unfinished_index"""
    env_num = int(g["env_num"])
    ep_lens = [int(x) for x in g["ep_lens"]]
    steps = synth_rollout(int(g["rollout_seed"]), env_num, [ep_lens] * env_num, obs_dim, act_dim)
    assert np.array_equal(rollout_checksum(steps), g["rollout_checksum"]), "numpy's generators no longer reproduce the fixture's rollout"
    em = rollout_env_major(steps, env_num)
    rows = sum(abs(x) for x in ep_lens)
    sub = int(g["sub_size"])
    slots = np.concatenate([e * sub + np.arange(rows) for e in range(env_num)])
    out = {"env_rows": np.full(env_num, rows), "slots": slots, "indices": slots.copy(),
           "unfinished_index": np.array([e * sub + rows - 1 for e in range(env_num) if ep_lens[-1] < 0], np.int64)}
    for k in ("obs", "act", "rew", "cost", "terminated", "truncated", "obs_next"):
        out["buf_" + k] = out["st_" + k] = em[k]
    return out, steps


def trust_full_case(name):
    """tests/golden/cpo_c3full.npz: BASELINE configs[2] (obs 60, act 2, 256x256, N = 20 000 full batch) through one repeat of the
    unmodified CPO.update; rollout and theta0 regenerated from seeds.  -> the dict the small-fixture tests take"""
    g = load_npz(f"cpo_{name}.npz")
    cfg = json.loads(str(g["cfg_json"]))
    arr, _ = full_rollout_arrays(g, cfg["obs_dim"], cfg["act_dim"])
    g.update(arr)
    shapes = [tuple(s) for s in json.loads(str(g["theta_shapes_json"]))]
    g["theta0"] = synth_theta(int(g["theta_seed"]), shapes)
    assert np.array_equal(theta_checksum(g["theta0"]), g["theta0_checksum"])
    g["perms"] = g["perms"].astype(np.int64)
    return g


def sac_full_case(name):
    """tests/golden/sac_c4full.npz: BASELINE configs[3]'s shape (obs 33, act 8, 256x256, batch 1024, n_step 2) over a 97 000-row
    store regenerated from a seed; parameters regenerated from seeds; the tanh of the rollout's raw actions is what is stored"""
    g = load_npz(f"sac_{name}.npz")
    cfg = json.loads(str(g["cfg_json"]))
    arr, _ = full_rollout_arrays(g, cfg["obs_dim"], cfg["act_dim"])
    arr["st_act"] = arr["buf_act"] = np.tanh(arr["st_act"])
    del arr["indices"]                                   # the fixture's own `indices` are the recorded buffer.sample draws
    g.update(arr)
    for key in ("actor", "critics"):
        shapes = [tuple(s) for s in json.loads(str(g[f"theta_{key}_shapes_json"]))]
        g[f"theta_{key}0"] = synth_theta(int(g[f"theta_{key}_seed"]), shapes)
        assert np.array_equal(theta_checksum(g[f"theta_{key}0"]), g[f"theta_{key}0_checksum"])
    return g
