#!/usr/bin/env python
"""Golden vectors G7: SACLagrangian.update (fsrl/policy/sac_lag.py) from the UNMODIFIED reference,
build container only.  Records inputs (replay store, parameters, the sampled indices and the
standard-normal draws of rsample) and outputs (per-update stats, parameters after K updates).

    python tests/golden/gen_golden_sac.py
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
from fsrl.policy import SACLagrangian  # noqa: E402
from fsrl.utils.net.common import ActorCritic  # noqa: E402
from fsrl.utils.net.continuous import DoubleCritic  # noqa: E402
from torch import nn  # noqa: E402

from gen_golden import CaptureLogger, fill_buffer, seed_all  # noqa: E402
from ref_shim import ActorProb, Net, _Box  # noqa: E402


def flat(mods):
    return torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()]).numpy().copy()


def gen(name, obs_dim, act_dim, hidden, env_num, ep_lens, batch_size, n_updates, seed, n_step=2,
        cost_stat=25.0, cost_limit=10.0, auto_alpha=True, alpha=0.005, tau=0.05, actor_lr=5e-4,
        critic_lr=1e-3, alpha_lr=3e-4, gamma=0.99, buffer_size=None, full=False):
    """full: a BASELINE-size case -- the rollout (tests/helpers.synth_rollout) and the initial parameters (synth_theta) are
    regenerated from seeds by the tests, the fixture keeps their checksums, the sampled indices, both noise blocks of every
    update, the logged rows and the parameters after the updates (the target critics as every 8th element)."""
    seed_all(seed)
    actor = ActorProb(Net((obs_dim, ), hidden_sizes=hidden), (act_dim, ), max_action=1.0,
                      conditioned_sigma=True, unbounded=True)
    actor_optim = torch.optim.Adam(actor.parameters(), lr=actor_lr)
    critics = []
    for _ in range(2):
        n1 = Net((obs_dim, ), (act_dim, ), hidden_sizes=hidden, concat=True)
        n2 = Net((obs_dim, ), (act_dim, ), hidden_sizes=hidden, concat=True)
        critics.append(DoubleCritic(n1, n2))
    critic_optim = torch.optim.Adam(nn.ModuleList(critics).parameters(), lr=critic_lr)
    ac = ActorCritic(actor, critics)
    for m in ac.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    # non-zero biases / a spread of sigma so that clamp and squash paths are exercised
    g = torch.Generator().manual_seed(seed + 3)
    with torch.no_grad():
        for p in ac.parameters():
            if p.ndim == 1:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    out = {}
    if full:
        sys.path.insert(0, os.path.dirname(HERE))
        from helpers import rollout_checksum, synth_rollout, synth_theta, theta_checksum
        for key, mods, sd in (("actor", [actor], seed + 500), ("critics", critics, seed + 501)):
            shapes = [tuple(p.shape) for m in mods for p in m.parameters()]
            th = synth_theta(sd, shapes)
            o = 0
            with torch.no_grad():
                for m in mods:
                    for p in m.parameters():
                        p.copy_(torch.from_numpy(th[o:o + p.numel()]).reshape(p.shape)); o += p.numel()
            assert o == th.size
            out[f"theta_{key}_seed"] = np.array(sd)
            out[f"theta_{key}_shapes_json"] = np.array(json.dumps([list(s_) for s_ in shapes]))
            out[f"theta_{key}0_checksum"] = theta_checksum(th)
    alpha_arg = alpha
    if auto_alpha:
        target_entropy = -float(act_dim)
        log_alpha = torch.zeros(1, requires_grad=True)
        alpha_optim = torch.optim.Adam([log_alpha], lr=alpha_lr)
        alpha_arg = (target_entropy, log_alpha, alpha_optim)
    logger = CaptureLogger()
    policy = SACLagrangian(actor=actor, critics=critics, actor_optim=actor_optim, critic_optim=critic_optim,
                           logger=logger, alpha=alpha_arg, tau=tau, gamma=gamma, n_step=n_step,
                           cost_limit=cost_limit, observation_space=_Box(-np.inf, np.inf, (obs_dim, )),
                           action_space=_Box(-1, 1, (act_dim, )))
    policy.train()
    rng = np.random.default_rng(seed + 1000)
    if full:
        from ref_shim import VectorReplayBuffer
        steps = synth_rollout(seed + 1000, env_num, [list(ep_lens)] * env_num, obs_dim, act_dim)
        buf = VectorReplayBuffer(buffer_size or 100000, env_num)
        for ids, obs, act, rew, cost, term, trunc, nxt in steps:
            buf.add({"obs": obs, "act": act, "rew": rew, "terminated": term, "truncated": trunc, "done": term | trunc,
                     "obs_next": nxt, "info.cost": cost}, ids)
        out.update(rollout_seed=np.array(seed + 1000), rollout_checksum=rollout_checksum(steps), ep_lens=np.array(ep_lens),
                   env_num=np.array(env_num))
    else:
        buf = fill_buffer(rng, env_num, ep_lens, obs_dim, act_dim, buffer_size=buffer_size or 100000)
    # stored actions of a tanh policy live in (-1, 1)
    buf._meta["act"][:] = np.tanh(buf._meta["act"])
    meta = buf._meta
    used = np.concatenate([np.arange(o, o + len(b)) for o, b in zip(buf._offset, buf.buffers)])
    if not full:
        out.update(theta_actor0=flat([actor]), theta_critics0=flat(critics))
        out["slots"] = used
        for k in ("obs", "act", "rew", "terminated", "truncated", "obs_next"):
            out["st_" + k] = meta[k][used]
        out["st_cost"] = meta["info.cost"][used]
        out["env_rows"] = np.array([len(b) for b in buf.buffers])
    out["sub_size"] = np.array(buf.buffers[0].maxsize)
    policy.pre_update_fn(stats_train={"cost": cost_stat})
    out["lagrangian"] = np.array([o.get_lag() for o in policy.lag_optims], np.float64)
    # ---- record indices and standard-normal draws
    idx_log, eps_log = [], []
    orig_sample = buf.sample
    import torch.distributions.normal as tdn
    orig_sn = tdn._standard_normal

    def rec_sample(bs):
        b, idx = orig_sample(bs)
        idx_log.append(np.asarray(idx).copy())
        return b, idx

    def rec_sn(shape, dtype, device):
        e = orig_sn(shape, dtype, device)
        eps_log.append(e.numpy().copy())
        return e

    buf.sample = rec_sample
    tdn._standard_normal = rec_sn
    try:
        seed_all(seed + 7)
        for _ in range(n_updates):
            policy.update(batch_size, buf)
    finally:
        tdn._standard_normal = orig_sn
    assert len(idx_log) == n_updates and len(eps_log) == 2 * n_updates
    out["indices"] = np.stack(idx_log)                 # [K][B]
    out["eps_target"] = np.stack(eps_log[0::2])        # [K][B][Da]  (target action at s_{t+n})
    out["eps_pi"] = np.stack(eps_log[1::2])            # [K][B][Da]  (policy action at s_t)
    rows = logger.rows
    assert len(rows) == 2 * n_updates
    ka, kc = list(rows[0].keys()), list(rows[1].keys())
    out["stats_actor_keys"], out["stats_critic_keys"] = np.array(ka), np.array(kc)
    out["stats_actor"] = np.array([[rows[2 * i][k] for k in ka] for i in range(n_updates)], np.float64)
    out["stats_critic"] = np.array([[rows[2 * i + 1][k] for k in kc] for i in range(n_updates)], np.float64)
    out["theta_actor_final"] = flat([actor])
    out["theta_critics_final"] = flat(critics)
    out["theta_critics_old_final"] = flat(list(policy.critics_old))
    if full:
        out["theta_critics_old_final"] = out["theta_critics_old_final"][::8].copy()
    out["alpha_final"] = np.array(float(policy._alpha))
    cfg = dict(obs_dim=obs_dim, act_dim=act_dim, hidden=list(hidden), env_num=env_num, batch_size=batch_size,
               n_updates=n_updates, seed=seed, n_step=n_step, cost_stat=cost_stat, cost_limit=cost_limit,
               auto_alpha=auto_alpha, alpha=alpha, tau=tau, actor_lr=actor_lr, critic_lr=critic_lr,
               alpha_lr=alpha_lr, gamma=gamma, max_action=1.0, lagrangian_pid=[0.05, 0.0005, 0.1],
               buffer_size=buffer_size or 100000)
    out["cfg_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, f"sac_{name}.npz"), **out)
    print(f"G7 sac_{name}.npz rows={len(used)} updates={n_updates} alpha_final={float(policy._alpha):.6f} "
          f"q_total last={out['stats_critic'][-1][kc.index('loss/q_total')]:.4f}")


if __name__ == "__main__":
    torch.set_num_threads(4)
    eps = [[60, 50, -17], [70, 55], [40, 40, 40, -9]]
    if sys.argv[1:] == ["full"]:
        # BASELINE configs[3]'s shape (SafetyAntRun: obs 33, act 8; 256x256; batch 1024; n_step 2, sacl_cfg.py:21) over a
        # 97 000-row store (10 envs x nine 1000-step episodes + an unfinished tail of 700)
        gen("c4full", 33, 8, (256, 256), 10, [1000] * 9 + [-700], batch_size=1024, n_updates=3, seed=62, n_step=2, full=True)
        sys.exit(0)
    if sys.argv[1:] == ["widths"]:
        # two hidden layers of different widths that are not 64 / 128 / 256 (zero-padded on the device)
        gen("widths", 6, 3, (80, 48), 3, eps, batch_size=64, n_updates=5, seed=35, n_step=2)
        sys.exit(0)
    if sys.argv[1:] == ["depths"]:
        # hidden_sizes the fused kernels do not hold (sac_lag_agent.py: any tuple): layered contexts on the HIP side
        gen("deep3", 6, 3, (48, 64, 40), 3, eps, batch_size=64, n_updates=5, seed=36, n_step=2)
        gen("wide1", 8, 2, (272, ), 3, eps, batch_size=100, n_updates=4, seed=37, n_step=3, auto_alpha=False, alpha=0.2)
        sys.exit(0)
    gen("small", 6, 3, (64, 64), 3, eps, batch_size=64, n_updates=6, seed=30, n_step=2)
    gen("nstep3", 8, 2, (64, 64), 3, eps, batch_size=128, n_updates=4, seed=31, n_step=3, auto_alpha=False,
        alpha=0.2)
    gen("c4", 33, 8, (128, 128), 4, [[200, 150], [200, -120], [250, 100], [300]], batch_size=256,
        n_updates=5, seed=32, n_step=2)
