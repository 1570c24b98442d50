"""Seeded random sweep of PPO-Lagrangian option combinations (the flags the fixtures cover one at a time: dual clip, no
advantage normalisation, no Lagrangian, grad-norm clip on / off, unbounded head, reward_normalization, value_clip,
recompute_advantage) x shapes (widths 64 / 128 / 256, two unrelated widths, and for the last seeds layered networks of 1 - 4 ragged layers; ragged sub-buffers, merged last minibatch, 4-row and 16-row tiles)
against the CPU oracle on the same inputs and permutations.  Tolerances as in test_gpu_shapes.py."""
import numpy as np
import pytest
import torch

from test_gpu_shapes import _synthetic

pytestmark = pytest.mark.gpu


def _case(seed):
    r = np.random.default_rng(1000 + seed)
    H = int(r.choice([64, 128, 256]))
    Do, Da = int(r.integers(2, 40)), int(r.integers(1, 7))
    n_env = int(r.integers(1, 5))
    rows = [int(r.integers(20, 400)) for _ in range(n_env)]
    B = int(r.choice([32, 64, 100, 256]))
    rew_norm = bool(r.random() < 0.5)
    return dict(H=H, Do=Do, Da=Da, rows=rows, ep=int(r.integers(15, 120)), B=B, repeat=int(r.integers(1, 4)),
                dual_clip=(float(r.uniform(1.5, 4.0)) if r.random() < 0.4 else None), norm_adv=bool(r.random() < 0.7),
                use_lagrangian=bool(r.random() < 0.8), max_grad_norm=(0.5 if r.random() < 0.6 else None),
                unbounded=bool(r.random() < 0.4), rew_norm=rew_norm, value_clip=bool(rew_norm and r.random() < 0.6),
                recompute=bool(r.random() < 0.4), max_action=float(r.choice([1.0, 1.5])), eps_clip=float(r.choice([0.2, 0.05])),
                vf_coef=float(r.choice([0.25, 1.0])), lr=float(r.choice([5e-4, 2e-3])))


@pytest.mark.parametrize("seed", range(27))
def test_random_option_combination_vs_oracle(seed):
    from fsrl_amd.engine import Engine, EngineConfig
    from oracle.ppo_lag import OnPolicyData, PPOLagConfig, PPOLagOracle
    c = _case(seed)
    hidden = (c["H"], c["H"])
    if seed >= 14:            # two hidden layers of unrelated widths (zero-padded to 64 / 128 / 256 on the device)
        rw = np.random.default_rng(5000 + seed)
        hidden = (int(rw.integers(5, 257)), int(rw.integers(5, 257))) if seed > 14 else (40, 24)      # seed 14: pads to 64
        c["hidden"] = hidden
    if seed >= 19:            # layered contexts (host_layered.inc): 1 - 4 hidden layers of ragged widths, a few above 256
        rw = np.random.default_rng(7000 + seed)
        depth = int(rw.choice([1, 3, 4])) if seed < 25 else 2
        hidden = tuple(int(rw.integers(3, 200)) for _ in range(depth))
        if seed >= 25:        # two layers, one wider than the fused kernels hold; seed 26: every width a multiple of 4 (float4 loads)
            hidden = (int(rw.integers(257, 420)), int(rw.integers(8, 300))) if seed == 25 else (288, 96)
        c["hidden"] = hidden
        if seed == 23:
            # (146, 137, 180, 166) at batch 32 sits on a discontinuity of the PPO objective (a clip boundary crossed at step 4):
            # the fp32 ORACLE restarted from theta0 * (1 + 1e-7 * noise) ends 5.5e-3 away from its own unperturbed run (55 430
            # entries > 1e-5, logged rows 3.7e-4 apart) -- and so does the device.  Batch 31 / 33 / 48 and every neighbouring
            # shape track to 1e-6; the gradients of the batch-32 steps match autograd to 1e-7.  Not a property of either
            # implementation, so the case runs at batch 48.
            c["B"] = 48
    rng = np.random.default_rng(seed)
    cols = _synthetic(rng, c["rows"], c["Do"], c["Da"], c["ep"])
    eng = Engine(EngineConfig(obs_dim=c["Do"], act_dim=c["Da"], hidden_sizes=hidden, env_num=len(c["rows"]), buffer_size=len(c["rows"]) * 512,
                              max_grad_norm=c["max_grad_norm"], target_kl=None, max_action=c["max_action"], dual_clip=c["dual_clip"],
                              norm_adv=c["norm_adv"], use_lagrangian=c["use_lagrangian"], unbounded=c["unbounded"],
                              rew_norm=c["rew_norm"], value_clip=c["value_clip"], recompute_adv=c["recompute"], eps_clip=c["eps_clip"],
                              vf_coef=c["vf_coef"], lr=c["lr"]))
    o = PPOLagOracle(PPOLagConfig(obs_dim=c["Do"], act_dim=c["Da"], hidden=hidden, max_grad_norm=c["max_grad_norm"],
                                  target_kl=1e9, max_action=c["max_action"], dual_clip=c["dual_clip"],
                                  advantage_normalization=c["norm_adv"], use_lagrangian=c["use_lagrangian"], unbounded=c["unbounded"],
                                  reward_normalization=c["rew_norm"], value_clip=c["value_clip"], recompute_advantage=c["recompute"],
                                  eps_clip=c["eps_clip"], vf_coef=c["vf_coef"], lr=c["lr"]))
    torch.manual_seed(seed)
    # an unbounded head with weights this large puts |mu| ~ 10 sigma away from the stored actions: the probability ratios of
    # the second step then reach 1e3 and amplify fp32 rounding of log pi by the same factor (seen: 1e-3 relative on the actor
    # loss) -- a conditioning artefact of the inputs, so the unbounded cases start closer to the data
    theta = ((0.05 if c["unbounded"] else 0.15) * torch.randn(o.n_params)).numpy()
    o.set_params(theta); eng.set_params(theta)
    if c["rew_norm"]:
        rms0 = np.array([[1.3, 4.0, 700.0], [0.2, 0.5, 700.0]])
        o.ret_rms[:] = rms0; eng.ret_rms_set(rms0)
    rows = c["rows"]
    for t in range(max(rows)):
        ids = [e for e in range(len(rows)) if t < rows[e]]
        eng.push(ids, *[np.stack([cols[k][e][t] for e in ids]) for k in ("obs", "act", "rew", "cost", "term", "trunc", "obs_next")])
    cat = {k: np.concatenate(v) for k, v in cols.items()}
    end = (cat["term"] | cat["trunc"]).copy(); end[np.cumsum(rows) - 1] = True
    data = OnPolicyData(obs=cat["obs"], act=cat["act"], rew=cat["rew"], cost=cat["cost"], terminated=cat["term"],
                        truncated=cat["trunc"], obs_next=cat["obs_next"], end_flag=end)
    N = len(data)
    lag = np.array([0.4]); resc = 1 / 1.4
    perms = [rng.permutation(N) for _ in range(c["repeat"])]
    _, ostats, _ = o.update(data, lag, resc, c["B"], c["repeat"], perms=perms)
    stats, stopped = eng.ppo_update(lag, resc, c["B"], c["repeat"], perms=perms)
    ostats = np.asarray(ostats)
    assert stopped == -1 and stats.shape == ostats.shape, c
    scale = np.maximum(np.abs(ostats), 1.0)
    assert (np.abs(stats - ostats) <= 5e-5 * scale).all(), (c, np.abs(stats - ostats).max(0))
    d = np.abs(eng.get_params() - o.get_params())
    lr_units = c["lr"] / 5e-4
    assert np.quantile(d, 0.999) <= 5e-6 * lr_units and d.max() <= 2e-4 * lr_units, (c, np.quantile(d, 0.999), d.max())
    if c["rew_norm"]:
        np.testing.assert_allclose(eng.ret_rms_get(), o.ret_rms, rtol=1e-5, atol=1e-7)
    eng.close()
