#!/usr/bin/env python
"""Golden vectors G5 (CPO) and G6 (TRPO-Lagrangian) from the UNMODIFIED reference
(fsrl/policy/cpo.py, fsrl/policy/trpo_lag.py), build container only.  Same rules as
gen_golden.py: inputs and outputs are recorded, no reference source is copied.

    python tests/golden/gen_golden_trust.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from fsrl.policy import CPO, TRPOLagrangian  # noqa: E402
from fsrl.utils.net.common import ActorCritic  # noqa: E402
from torch import nn  # noqa: E402
from torch.distributions import Independent, Normal  # noqa: E402

from gen_golden import CaptureLogger, fill_buffer, flat_params, seed_all  # noqa: E402
from ref_shim import ActorProb, Critic, Net, _Box  # noqa: E402


def build_nets(obs_dim, act_dim, hidden, seed, unbounded=False):
    seed_all(seed)
    actor = ActorProb(Net((obs_dim, ), hidden_sizes=hidden), (act_dim, ), max_action=1.0, unbounded=unbounded)
    critic = [Critic(Net((obs_dim, ), hidden_sizes=hidden)) for _ in range(2)]
    torch.nn.init.constant_(actor.sigma_param, -0.5)
    ac = ActorCritic(actor, critic)
    for m in ac.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    return actor, critic, ac


def dist(*logits):
    return Independent(Normal(*logits), 1)


class PermRecorder:
    """Records the np.random.permutation draws of Batch.split (the full batch is shuffled)."""

    def __enter__(self):
        self.perms, self._orig = [], np.random.permutation

        def rec(n):
            p = self._orig(n)
            self.perms.append(np.asarray(p).copy())
            return p

        np.random.permutation = rec
        return self

    def __exit__(self, *a):
        np.random.permutation = self._orig


def preset_rms(out, policy, ret_rms0):
    """reward_normalization cases: start from the running return statistics an earlier update would have left"""
    if ret_rms0 is not None:
        for rms, (m, v, n) in zip(policy.ret_rms, ret_rms0):
            rms.mean, rms.var, rms.count = float(m), float(v), float(n)
        out["ret_rms0"] = np.array([[r.mean, r.var, r.count] for r in policy.ret_rms], np.float64)


def record_batch(out, buf):
    batch, indices = buf.sample(0)
    out["indices"] = indices
    for k in ("obs", "act", "rew", "terminated", "truncated", "obs_next"):
        out["buf_" + k] = getattr(batch, k)
    out["buf_cost"] = batch.info.cost
    out["unfinished_index"] = buf.unfinished_index()
    out["env_rows"] = np.array([len(b) for b in buf.buffers])


def gen_cpo(name, obs_dim, act_dim, hidden, env_num, ep_lens, repeat, seed, cost_stat, cost_limit,
            lr=1e-3, perturb_actor=0.0, zero_cost_signal=False, unbounded=False, ret_rms0=None, batch_size=99999, **kw):
    """cost_stat / cost_limit steer the optim_case; perturb_actor moves theta away from the
    theta that produced mean_old (exercises the exact-Hessian path on the very first call)."""
    actor, critic, ac = build_nets(obs_dim, act_dim, hidden, seed, unbounded)
    if zero_cost_signal:
        # optim_case 4 (cpo.py:261-268: grad_b . grad_b <= 1e-8 and c < 0): no cost in the data and a cost critic that
        # answers 0 everywhere (zero last layer) => cost advantages are exactly 0 => the cost-surrogate gradient vanishes
        with torch.no_grad():
            for p in critic[1].last.parameters():
                p.zero_()
    optim = torch.optim.Adam(nn.ModuleList(critic).parameters(), lr=lr)
    logger = CaptureLogger()
    policy = CPO(actor, critic, optim, dist, logger=logger, cost_limit=cost_limit,
                 observation_space=_Box(-np.inf, np.inf, (obs_dim, )),
                 action_space=_Box(-1, 1, (act_dim, )), **kw)
    policy.train()
    buf = fill_buffer(np.random.default_rng(seed + 1000), env_num, ep_lens, obs_dim, act_dim,
                      cost_prob=0.0 if zero_cost_signal else 0.1)
    out = {"theta0": flat_params(ac)}
    record_batch(out, buf)
    policy.pre_update_fn(stats_train={"cost": cost_stat})
    # capture internals of every policy_loss call by wrapping the policy's own methods
    cap = {"cg": []}
    orig_cg = policy._conjugate_gradients

    def cg(g, fkg, *a, **k):
        x = orig_cg(g, fkg, *a, **k)
        cap["cg"].append(x.detach().numpy().copy())
        return x

    policy._conjugate_gradients = cg
    preset_rms(out, policy, ret_rms0)
    batch, indices = buf.sample(0)
    import copy
    rms_keep = copy.deepcopy(policy.ret_rms)      # the recording call below must not advance the running statistics
    pbatch = policy.process_fn(batch, buf, indices)
    policy.ret_rms = rms_keep
    out["advs_norm"] = pbatch.advs.numpy().copy()
    out["rets"] = pbatch.rets.numpy().copy()
    out["logp_old"] = pbatch.logp_old.numpy().copy()
    out["mean_old"] = pbatch.mean_old.numpy().copy()
    out["std_old"] = pbatch.std_old.numpy().copy()
    if perturb_actor:
        g = torch.Generator().manual_seed(seed + 5)
        with torch.no_grad():
            for p in actor.parameters():
                p.add_(perturb_actor * torch.randn(p.shape, generator=g))
        out["theta0_perturbed"] = flat_params(ac)
        # one stand-alone policy_loss on the processed batch (theta != theta_old)
        for _ in range(policy._optim_critic_iters):
            policy.critics_loss(pbatch)
        _, st = policy.policy_loss(pbatch)
        out["pl_stats_keys"] = np.array(list(st.keys()))
        out["pl_stats"] = np.array([float(st[k]) for k in st], np.float64)
        out["pl_H_inv_g"] = cap["cg"][0]
        if len(cap["cg"]) > 1:
            out["pl_H_inv_b"] = cap["cg"][1]
        out["theta_after_pl"] = flat_params(ac)
    else:
        with PermRecorder() as pr:
            policy.update(0, buf, batch_size=batch_size, repeat=repeat)
        out["perms"] = np.stack(pr.perms)
        rows = [r for r in logger.rows if "update/gradient_steps" not in r]
        n_rows = len(rows) // 2                       # one (actor, critic) row pair per minibatch of every repeat (cpo.py:357-366)
        assert len(rows) == 2 * n_rows and n_rows % repeat == 0 and (batch_size < len(indices) or n_rows == repeat)
        keys_a = [k for k in rows[0].keys()]
        keys_c = [k for k in rows[1].keys()]
        out["stats_actor_keys"] = np.array(keys_a)
        out["stats_critic_keys"] = np.array(keys_c)
        out["stats_actor"] = np.array([[rows[2 * i][k] for k in keys_a] for i in range(n_rows)])
        out["stats_critic"] = np.array([[rows[2 * i + 1][k] for k in keys_c] for i in range(n_rows)])
        if batch_size < len(indices):
            out["gradient_steps"] = np.array(policy.gradient_steps)
        out["H_inv_g_first"] = cap["cg"][0]
        out["theta_final"] = flat_params(ac)
        if ret_rms0 is not None:
            out["ret_rms_final"] = np.array([[r.mean, r.var, r.count] for r in policy.ret_rms], np.float64)
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, repeat=repeat,
               seed=seed, cost_stat=cost_stat, cost_limit=cost_limit, lr=lr, max_action=1.0,
               perturb_actor=perturb_actor)
    if unbounded:
        cfg["unbounded"] = True
    if batch_size != 99999:
        cfg["batch_size"] = batch_size
    defaults = dict(target_kl=0.01, backtrack_coeff=0.8, damping_coeff=0.1, max_backtracks=10,
                    optim_critic_iters=20, l2_reg=0.001, gae_lambda=0.95, advantage_normalization=True,
                    gamma=0.99)
    cfg.update(kw)
    for k, v in defaults.items():
        cfg.setdefault(k, v)
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"cpo_{name}.npz"), **out)
    if perturb_actor:
        print(f"G5 cpo_{name}.npz  N={len(indices)} policy_loss case={int(st['loss/optim_case'])} "
              f"step={st['loss/step_size']:.4f} kl={st['loss/kl']:.3e}")
    else:
        ic = keys_a.index("loss/optim_case"); isz = keys_a.index("loss/step_size")
        print(f"G5 cpo_{name}.npz  N={len(indices)} cases={out['stats_actor'][:, ic]} "
              f"steps={np.round(out['stats_actor'][:, isz], 4)}")


def set_flat(module, theta):
    """module.parameters() <- the flat vector, in order"""
    o = 0
    with torch.no_grad():
        for p in module.parameters():
            n = p.numel()
            p.copy_(torch.from_numpy(theta[o:o + n]).reshape(p.shape))
            o += n
    assert o == theta.size


def gen_cpo_full(name="c3full", obs_dim=60, act_dim=2, hidden=(256, 256), env_num=20, ep_lens=(1000, ), seed=61,
                 cost_stat=25.0, cost_limit=10.0, lr=1e-3, optim_critic_iters=10, max_backtracks=10):
    """BASELINE configs[2] at full size: CPO (cpo.py:234-351) on obs 60 / act 2 / 256x256, N = 20 000 rows as ONE full batch
    (cpo_cfg.py:84-90), CG 10, 10 critic steps, ONE repeat of the UNMODIFIED CPO.update.  Neither the rollout nor theta0 is
    stored (tests/helpers.synth_rollout / synth_theta regenerate them; checksums here).  Kept: process_fn's advs / logp_old /
    mean_old, the permutation Batch.split drew, both logger rows, H^-1 g and H^-1 b, theta after the update."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import rollout_checksum, synth_rollout, synth_theta, theta_checksum
    from ref_shim import VectorReplayBuffer
    actor, critic, ac = build_nets(obs_dim, act_dim, hidden, seed)
    shapes = [tuple(p.shape) for p in ac.parameters()]
    theta0 = synth_theta(seed + 500, shapes)
    set_flat(ac, theta0)
    optim = torch.optim.Adam(nn.ModuleList(critic).parameters(), lr=lr)
    logger = CaptureLogger()
    kw = dict(optim_critic_iters=optim_critic_iters, max_backtracks=max_backtracks)
    policy = CPO(actor, critic, optim, dist, logger=logger, cost_limit=cost_limit,
                 observation_space=_Box(-np.inf, np.inf, (obs_dim, )), action_space=_Box(-1, 1, (act_dim, )), **kw)
    policy.train()
    steps = synth_rollout(seed + 1000, env_num, [list(ep_lens)] * env_num, obs_dim, act_dim)
    buf = VectorReplayBuffer(100000, env_num)
    for ids, obs, act, rew, cost, term, trunc, nxt in steps:
        buf.add({"obs": obs, "act": act, "rew": rew, "terminated": term, "truncated": trunc, "done": term | trunc,
                 "obs_next": nxt, "info.cost": cost}, ids)
    out = {"rollout_seed": np.array(seed + 1000), "rollout_checksum": rollout_checksum(steps), "ep_lens": np.array(ep_lens),
           "env_num": np.array(env_num), "sub_size": np.array(buf.buffers[0].maxsize), "theta_seed": np.array(seed + 500),
           "theta_shapes_json": np.array(json.dumps([list(s) for s in shapes])), "theta0_checksum": theta_checksum(theta0)}
    policy.pre_update_fn(stats_train={"cost": cost_stat})
    cap = {"cg": []}
    orig_cg = policy._conjugate_gradients

    def cg(g, fkg, *a, **k):
        x = orig_cg(g, fkg, *a, **k)
        cap["cg"].append(x.detach().numpy().copy())
        return x
    policy._conjugate_gradients = cg
    batch, indices = buf.sample(0)
    import copy
    rms_keep = copy.deepcopy(policy.ret_rms)
    pbatch = policy.process_fn(batch, buf, indices)
    policy.ret_rms = rms_keep
    out["advs_norm"] = pbatch.advs.numpy().copy()
    out["logp_old"] = pbatch.logp_old.numpy().copy()
    out["mean_old"] = pbatch.mean_old.numpy().copy()
    with PermRecorder() as pr:
        seed_all(seed + 7)
        policy.update(0, buf, batch_size=99999, repeat=1)
    assert len(pr.perms) == 1 and pr.perms[0].max() < 65536
    out["perms"] = np.stack(pr.perms).astype(np.uint16)
    rows = [r for r in logger.rows if "update/gradient_steps" not in r]
    assert len(rows) == 2 and len(cap["cg"]) == 2
    keys_a, keys_c = list(rows[0].keys()), list(rows[1].keys())
    out["stats_actor_keys"], out["stats_critic_keys"] = np.array(keys_a), np.array(keys_c)
    out["stats_actor"] = np.array([[rows[0][k] for k in keys_a]])
    out["stats_critic"] = np.array([[rows[1][k] for k in keys_c]])
    out["H_inv_g_first"], out["H_inv_b_first"] = cap["cg"][0], cap["cg"][1]
    out["theta_final"] = flat_params(ac)
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, repeat=1, seed=seed, cost_stat=cost_stat,
               cost_limit=cost_limit, lr=lr, max_action=1.0, perturb_actor=0.0, target_kl=0.01, backtrack_coeff=0.8,
               damping_coeff=0.1, l2_reg=0.001, gae_lambda=0.95, advantage_normalization=True, gamma=0.99, **kw)
    out["cfg_json"] = np.array(json.dumps(cfg))
    f = os.path.join(HERE, f"cpo_{name}.npz")
    np.savez_compressed(f, **out)
    ic, isz = keys_a.index("loss/optim_case"), keys_a.index("loss/step_size")
    print(f"G5 cpo_{name}.npz N={len(indices)} case={out['stats_actor'][0, ic]} step={out['stats_actor'][0, isz]:.4f} "
          f"size={os.path.getsize(f) / 1e6:.2f} MB")


def gen_trpo(name, obs_dim, act_dim, hidden, env_num, ep_lens, repeat, seed, cost_stat, cost_limit,
             lr=5e-4, unbounded=False, ret_rms0=None, batch_size=99999, **kw):
    actor, critic, ac = build_nets(obs_dim, act_dim, hidden, seed, unbounded)
    optim = torch.optim.Adam(ac.parameters(), lr=lr)
    logger = CaptureLogger()
    policy = TRPOLagrangian(actor, critic, optim, dist, logger=logger, cost_limit=cost_limit,
                            observation_space=_Box(-np.inf, np.inf, (obs_dim, )),
                            action_space=_Box(-1, 1, (act_dim, )), **kw)
    policy.train()
    buf = fill_buffer(np.random.default_rng(seed + 1000), env_num, ep_lens, obs_dim, act_dim)
    out = {"theta0": flat_params(ac)}
    record_batch(out, buf)
    policy.pre_update_fn(stats_train={"cost": cost_stat})
    out["lagrangian"] = np.array([o.get_lag() for o in policy.lag_optims], np.float64)
    cap = {"cg": []}
    orig_cg = policy._conjugate_gradients

    def cg(g, fkg, *a, **k):
        x = orig_cg(g, fkg, *a, **k)
        cap["cg"].append(x.detach().numpy().copy())
        return x

    policy._conjugate_gradients = cg
    preset_rms(out, policy, ret_rms0)
    with PermRecorder() as pr:
        policy.update(0, buf, batch_size=batch_size, repeat=repeat)
    out["perms"] = np.stack(pr.perms)
    if ret_rms0 is not None:
        out["ret_rms_final"] = np.array([[r.mean, r.var, r.count] for r in policy.ret_rms], np.float64)
    rows = [r for r in logger.rows if "update/gradient_steps" not in r]
    n_rows = len(rows) // 3                           # three logger rows per minibatch of every repeat (trpo_lag.py:178-249)
    assert len(rows) == 3 * n_rows and n_rows % repeat == 0 and (batch_size < len(out["indices"]) or n_rows == repeat)
    keys = []
    for r in rows[:3]:
        keys += list(r.keys())
    out["stats_keys"] = np.array(keys)
    out["stats"] = np.array([[{**rows[3 * i], **rows[3 * i + 1], **rows[3 * i + 2]}[k] for k in keys]
                             for i in range(n_rows)])
    out["cg_first"] = cap["cg"][0]
    out["theta_final"] = flat_params(ac)
    out["gradient_steps"] = np.array(policy.gradient_steps)
    out["msgs"] = np.array(len(logger.msgs))
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, repeat=repeat,
               seed=seed, cost_stat=cost_stat, cost_limit=cost_limit, lr=lr, max_action=1.0)
    if unbounded:
        cfg["unbounded"] = True
    if batch_size != 99999:
        cfg["batch_size"] = batch_size
    defaults = dict(target_kl=0.001, backtrack_coeff=0.8, max_backtracks=10, optim_critic_iters=5,
                    gae_lambda=0.95, advantage_normalization=True, gamma=0.99,
                    lagrangian_pid=(0.05, 0.0005, 0.1), rescaling=True, use_lagrangian=True)
    cfg.update(kw)
    for k, v in defaults.items():
        cfg.setdefault(k, v)
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"trpo_{name}.npz"), **out)
    ik = keys.index("loss/step_size")
    print(f"G6 trpo_{name}.npz N={len(out['indices'])} step_sizes={np.round(out['stats'][:, ik], 5)} "
          f"msgs={len(logger.msgs)}")


if __name__ == "__main__":
    torch.set_num_threads(4)
    eps = [[100, 100], [100, -60], [120, 80]]
    if sys.argv[1:] == ["full"]:
        gen_cpo_full()
        sys.exit(0)
    if sys.argv[1:] == ["case4"]:
        # the vanishing-cost-gradient branch: zero cost signal, no advantage normalisation (0 / 0 otherwise), c < 0
        gen_cpo("case4", 8, 2, (64, 64), 3, eps, repeat=2, seed=16, cost_stat=0.0, cost_limit=10.0,
                optim_critic_iters=3, max_backtracks=12, advantage_normalization=False, zero_cost_signal=True)
        sys.exit(0)
    if sys.argv[1:] == ["options"]:
        # unbounded actor head + reward_normalization from a non-trivial running state, through the trust-region updates
        rms0 = [(2.9, 6.5, 1500.0), (0.35, 0.8, 1500.0)]
        gen_cpo("options", 8, 2, (64, 64), 3, eps, repeat=2, seed=17, cost_stat=25.0, cost_limit=10.0,
                optim_critic_iters=5, max_backtracks=10, unbounded=True, reward_normalization=True, ret_rms0=rms0)
        gen_trpo("options", 8, 2, (64, 64), 3, eps, repeat=2, seed=22, cost_stat=25.0, cost_limit=10.0,
                 optim_critic_iters=5, unbounded=True, reward_normalization=True, ret_rms0=rms0)
        sys.exit(0)
    if sys.argv[1:] == ["widths"]:
        # two hidden layers of different widths that are not 64 / 128 / 256 (zero-padded on the device)
        gen_cpo("widths", 8, 2, (100, 50), 3, eps, repeat=2, seed=33, cost_stat=25.0, cost_limit=10.0,
                optim_critic_iters=3, max_backtracks=10)
        gen_trpo("widths", 8, 2, (48, 80), 3, eps, repeat=2, seed=34, cost_stat=25.0, cost_limit=10.0,
                 optim_critic_iters=3)
        sys.exit(0)
    if sys.argv[1:] == ["depths"]:
        # hidden_sizes the fused kernels do not hold (cpo_agent.py / trpol_agent.py: any tuple): layered contexts on the HIP side
        gen_cpo("deep3", 8, 2, (48, 64, 40), 3, eps, repeat=2, seed=41, cost_stat=25.0, cost_limit=10.0,
                optim_critic_iters=3, max_backtracks=10)
        gen_trpo("deep3", 8, 2, (40, 56, 32), 3, eps, repeat=2, seed=38, cost_stat=25.0, cost_limit=10.0,
                 optim_critic_iters=3)
        gen_cpo("wide1", 8, 2, (272, ), 3, eps, repeat=2, seed=39, cost_stat=3.0, cost_limit=10.0,
                optim_critic_iters=3, max_backtracks=10)
        sys.exit(0)
    if sys.argv[1:] == ["wideobs"]:
        # 100 observation columns: the first layer's weight gradient takes three passes of the weight-side kernel's aux blocks
        gen_cpo("wideobs", 100, 4, (64, 64), 3, eps, repeat=2, seed=36, cost_stat=25.0, cost_limit=10.0,
                optim_critic_iters=3, max_backtracks=10)
        sys.exit(0)
    if sys.argv[1:] == ["minibatch"]:
        # batch_size below the buffer: Batch.split(batch_size, merge_last=True) inside learn (cpo.py:357-358,
        # trpo_lag.py:178).  N = 560 rows, batch 150 -> minibatches of 150 / 150 / 260 (the remainder merged into the last)
        gen_cpo("minibatch", 8, 2, (64, 64), 3, eps, repeat=2, seed=31, cost_stat=25.0, cost_limit=10.0,
                optim_critic_iters=3, max_backtracks=10, batch_size=150)
        gen_trpo("minibatch", 8, 2, (64, 64), 3, eps, repeat=2, seed=32, cost_stat=25.0, cost_limit=10.0,
                 optim_critic_iters=3, batch_size=150)
        sys.exit(0)
    # cost far above the limit (c > 0): infeasible / recovery branches
    gen_cpo("infeasible", 8, 2, (64, 64), 3, eps, repeat=2, seed=10, cost_stat=25.0, cost_limit=10.0,
            optim_critic_iters=5, max_backtracks=10)
    # cost below the limit: feasible branches (cases 2-4)
    gen_cpo("feasible", 8, 2, (64, 64), 3, eps, repeat=2, seed=11, cost_stat=3.0, cost_limit=10.0,
            optim_critic_iters=5, max_backtracks=10)
    gen_cpo("edge", 8, 2, (64, 64), 3, eps, repeat=3, seed=12, cost_stat=9.9, cost_limit=10.0,
            optim_critic_iters=3, max_backtracks=25)
    # |c| tiny => the safety boundary intersects the trust region (B >= 0): cases 1 and 2
    gen_cpo("case1", 8, 2, (64, 64), 3, eps, repeat=2, seed=14, cost_stat=10.002, cost_limit=10.0,
            optim_critic_iters=3, max_backtracks=12)
    gen_cpo("case2", 8, 2, (64, 64), 3, eps, repeat=2, seed=15, cost_stat=9.998, cost_limit=10.0,
            optim_critic_iters=3, max_backtracks=12)
    # 128x128, obs 12 / act 3, actor perturbed away from theta_old: exact Hessian != Fisher
    gen_cpo("perturbed", 12, 3, (128, 128), 2, [[150, 150], [150, -90]], repeat=1, seed=13,
            cost_stat=14.0, cost_limit=10.0, optim_critic_iters=4, max_backtracks=15, perturb_actor=0.02)
    gen_trpo("small", 8, 2, (64, 64), 3, eps, repeat=2, seed=20, cost_stat=25.0, cost_limit=10.0,
             optim_critic_iters=5)
    gen_trpo("c1", 8, 2, (128, 128), 4, [[300], [300], [300], [-150]], repeat=2, seed=21, cost_stat=4.0,
             cost_limit=10.0, optim_critic_iters=5, target_kl=0.01)
