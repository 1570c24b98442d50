"""Shape / edge-case sweep of the HIP paths against the CPU oracle on seeded synthetic inputs, plus
the error behaviour of the C ABI on a GPU.  Covers what the golden fixtures do not: wide inputs with
4-row tiles (layer 1 on MFMA), the maximum dimensions, a batch size above N (one minibatch), the
merged last minibatch taking the 16-row kernel inside the same pass, ragged / single-row sub-buffers,
and a wrapped replay store for SAC.  Tolerances as in test_gpu_ppo.py (stats 2e-5, theta 5e-6: few
optimiser steps, so no chaotic drift yet)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _synthetic(rng, rows_per_env, Do, Da, ep):
    cols = {k: [] for k in ("obs", "act", "rew", "cost", "term", "trunc", "obs_next")}
    for T in rows_per_env:
        obs = rng.standard_normal((T + 1, Do)).astype(np.float32)
        act = (0.3 * rng.standard_normal((T, Da))).astype(np.float32)
        rew = rng.normal(0.5, 0.5, T); cost = (rng.random(T) < 0.2).astype(np.float64)
        trunc = np.zeros(T, bool); trunc[ep - 1::ep] = True
        term = np.zeros(T, bool)
        if T > 3:
            term[T // 2] = True
        for k, v in zip(cols, (obs[:-1], act, rew, cost, term, trunc, obs[1:])):
            cols[k].append(v)
    return cols


SHAPES = [  # obs, act, hidden, rows per env, episode length, batch, repeat
    (33, 6, 128, [300, 257, 143], 50, 64, 2),          # wide input + 4-row tiles, ragged envs
    (128, 16, 64, [200, 200], 40, 128, 2),             # maximum obs / act dims
    (17, 1, 256, [300, 300], 75, 256, 2),              # merged last minibatch (344 rows) -> 16-row kernel
    (3, 2, 64, [90, 1, 35], 30, 1000, 3),              # batch > N: one minibatch; a single-row sub-buffer
    (60, 2, 256, [520], 520, 512, 1),                  # one env, unfinished episode only, two 16-row steps
    (8, 2, 256, [700, 600], 100, 1024, 2),             # one merged minibatch of 1 300 rows: the split-K weight-gradient path
    (12, 3, 128, [900, 900, 500], 150, 1024, 2),       # 1 024 + 1 276 rows per pass at 128 wide
    (1, 1, 64, [100, 60], 25, 64, 2),                  # ONE observation column (r6: the stage's divide-by-multiply needs its d = 1 case)
]


@pytest.mark.parametrize("Do,Da,H,rows,ep,B,repeat", SHAPES)
def test_ppo_update_shape_sweep_vs_oracle(Do, Da, H, rows, ep, B, repeat):
    from fsrl_amd.engine import Engine, EngineConfig
    from oracle.ppo_lag import OnPolicyData, PPOLagConfig, PPOLagOracle
    rng = np.random.default_rng(Do * 1000 + H)
    cols = _synthetic(rng, rows, Do, Da, ep)
    eng = Engine(EngineConfig(obs_dim=Do, act_dim=Da, hidden=H, env_num=len(rows), buffer_size=len(rows) * 1024,
                              max_grad_norm=0.5, target_kl=None, max_action=1.5))
    o = PPOLagOracle(PPOLagConfig(obs_dim=Do, act_dim=Da, hidden=(H, H), max_grad_norm=0.5, target_kl=1e9,
                                  max_action=1.5))
    torch.manual_seed(Do)
    theta = (0.15 * torch.randn(o.n_params)).numpy()
    o.set_params(theta); eng.set_params(theta)
    for t in range(max(rows)):                          # lock-step, envs drop out as they run dry
        ids = [e for e in range(len(rows)) if t < rows[e]]
        eng.push(ids, *[np.stack([cols[k][e][t] for e in ids]) for k in ("obs", "act", "rew", "cost", "term", "trunc",
                                                                         "obs_next")])
    cat = {k: np.concatenate(v) for k, v in cols.items()}
    end = cat["term"] | cat["trunc"]
    off = np.cumsum(rows) - 1
    end = end.copy(); end[off] = True                   # unfinished tails
    data = OnPolicyData(obs=cat["obs"], act=cat["act"], rew=cat["rew"], cost=cat["cost"], terminated=cat["term"],
                        truncated=cat["trunc"], obs_next=cat["obs_next"], end_flag=end)
    N = len(data)
    lag = np.array([0.4]); resc = 1 / 1.4
    perms = [rng.permutation(N) for _ in range(repeat)]
    pb, ostats, _ = o.update(data, lag, resc, B, repeat, perms=perms)
    stats, stopped = eng.ppo_update(lag, resc, B, repeat, perms=perms)
    assert stopped == -1 and stats.shape == np.asarray(ostats).shape
    for k in ("advs", "rets", "values"):
        scale = max(1.0, float(pb[k].abs().max()))
        np.testing.assert_allclose(eng.batch_get(k), pb[k].numpy(), rtol=0, atol=5e-6 * scale, err_msg=k)
    np.testing.assert_allclose(stats, np.asarray(ostats), rtol=3e-5, atol=3e-5)
    d = np.abs(eng.get_params() - o.get_params())      # Adam on noise-level gradients: a handful of entries
    assert np.quantile(d, 0.999) <= 5e-6 and d.max() <= 1e-4, (np.quantile(d, 0.999), d.max())   # may differ by ~lr*1e-2
    eng.close()


def test_c_abi_error_behaviour_on_gpu():
    from fsrl_amd.engine import Engine, EngineConfig
    eng = Engine(EngineConfig(obs_dim=4, act_dim=2, hidden=64, env_num=2, buffer_size=64))
    # empty store: like the reference (sample(0) of an empty buffer -> empty batch -> no minibatches)
    assert eng.ppo_begin([0.1], 1.0, 16) == 0
    assert eng.ppo_pass(None) is False
    assert eng.ppo_end_stats(4).shape == (0, 11)
    with pytest.raises((AssertionError, RuntimeError)):          # pass without begin
        eng.ppo_pass(None)
    z = np.zeros
    eng.push([0, 1], z((2, 4), np.float32), z((2, 2), np.float32), z(2), z(2), z(2, bool), z(2, bool), z((2, 4), np.float32))
    with pytest.raises(AssertionError):                          # buffer id out of range
        eng.push([2], z((1, 4), np.float32), z((1, 2), np.float32), z(1), z(1), z(1, bool), z(1, bool), z((1, 4), np.float32))
    with pytest.raises(AssertionError):                          # non-positive batch size
        eng.ppo_begin([0.1], 1.0, 0)
    n = eng.ppo_begin([0.1], 1.0, 16)
    assert n == 2
    with pytest.raises(AssertionError):                          # not a permutation of the batch
        eng.ppo_pass(np.array([0, 0]))
    with pytest.raises((AssertionError, RuntimeError)):          # SAC entry point on a PPO context
        eng.sac_init()
    with pytest.raises(AssertionError):
        eng.set_params(np.zeros(3, np.float32))
    eng.close()


def test_sac_on_a_wrapped_store_device_and_host_chains_agree():
    """Sub-buffers overwritten 2.5 times: ReplayBuffer.next / unfinished_index across the wrap point.
    The device sampler (chains computed in sac_sample_kernel) and the caller-RNG mode (chains computed
    on the host) must produce bit-identical updates from the same indices."""
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    rng = np.random.default_rng(3)
    Do, Da, E, sub = 5, 3, 3, 40

    def make():
        eng = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=Do, act_dim=Da, hidden=64, n_critics=2, env_num=E,
                                  buffer_size=E * sub, gamma=0.97, target_kl=None))
        eng.sac_init(n_step=3)
        r = np.random.default_rng(0)
        eng.sac_set_params((0.2 * r.standard_normal(eng.n_sac_actor)).astype(np.float32),
                           (0.2 * r.standard_normal(eng.n_sac_critics)).astype(np.float32), -0.3)
        return eng
    dev, twin = make(), make()
    T = 100                                                  # 2.5 x the sub-buffer
    obs = rng.standard_normal((T + 1, E, Do)).astype(np.float32)
    act = np.tanh(rng.standard_normal((T, E, Da))).astype(np.float32)
    rew = rng.normal(0, 1, (T, E)); cost = (rng.random((T, E)) < 0.3).astype(np.float64)
    term = rng.random((T, E)) < 0.05; trunc = np.zeros((T, E), bool); trunc[12::13] = True
    for t in range(T):
        ids = [0, 1, 2] if t % 7 else [0, 2]                 # env 1 lags: different write cursors
        for e_ in (dev, twin):
            e_.push(ids, obs[t, ids], act[t, ids], rew[t, ids], cost[t, ids], term[t, ids], trunc[t, ids], obs[t + 1, ids])
    assert len(dev) == E * sub
    for u in range(6):
        dev.sac_update(64, [0.2], 1 / 1.2, seed=5 if u == 0 else 0, sync=False)
        idx, et, ep = dev.sac_last_sample(64)
        assert (idx >= 0).all() and (idx < E * sub).all()
        st = twin.sac_update(64, [0.2], 1 / 1.2, indices=idx, eps_target=et, eps_pi=ep)
    rows = dev.sac_drain()
    assert np.isfinite(rows).all() and np.array_equal(rows[-1], st)
    for which in (0, 1, 2):
        assert np.array_equal(dev.sac_get_params(which)[0], twin.sac_get_params(which)[0])
    dev.close(); twin.close()


SAC_VARIANTS = [  # Do, Da, H, rows per env, batch, n_step, auto_alpha, use_lagrangian
    (7, 3, 64, [40, 17], 100, 1, True, True),            # batch not a multiple of 16, 1-step targets
    (20, 8, 128, [64, 64, 64], 16, 3, False, True),      # fixed temperature, maximum action width
    (9, 2, 256, [33], 1, 2, True, False),                # batch of one row, no Lagrangian term
    (41, 1, 64, [90, 45], 333, 2, True, True),           # batch larger than the store (sampling with replacement)
]


@pytest.mark.parametrize("Do,Da,H,rows,B,n_step,auto_alpha,use_lag", SAC_VARIANTS)
def test_sac_variants_vs_oracle(Do, Da, H, rows, B, n_step, auto_alpha, use_lag):
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    from oracle.sac_lag import ReplayIndex, SACConfig, SACLagOracle
    rng = np.random.default_rng(Do + 10 * Da)
    E, sub = len(rows), 128
    eng = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=Do, act_dim=Da, hidden=H, n_critics=2, env_num=E,
                              buffer_size=E * sub, gamma=0.98, target_kl=None))
    eng.sac_init(n_step=n_step, auto_alpha=auto_alpha, alpha=0.05, use_lagrangian=use_lag, tau=0.1)
    o = SACLagOracle(SACConfig(obs_dim=Do, act_dim=Da, hidden=(H, H), gamma=0.98, n_step=n_step, tau=0.1, alpha=0.05,
                               auto_alpha=auto_alpha, use_lagrangian=use_lag))
    tha = (0.2 * rng.standard_normal(o.n_actor)).astype(np.float32)
    thc = (0.2 * rng.standard_normal(2 * o.n_critic)).astype(np.float32)
    o.set_params(tha, thc, -0.5); eng.sac_set_params(tha, thc, -0.5)
    store = {k: np.zeros((E * sub, ) + s, d) for k, s, d in (("obs", (Do, ), np.float32), ("obs_next", (Do, ), np.float32),
             ("act", (Da, ), np.float32), ("rew", (), np.float64), ("cost", (), np.float64),
             ("terminated", (), bool), ("truncated", (), bool))}
    for t in range(max(rows)):
        ids = [e for e in range(E) if t < rows[e]]
        k = len(ids)
        row = dict(obs=rng.standard_normal((k, Do)).astype(np.float32), act=np.tanh(rng.standard_normal((k, Da))).astype(np.float32),
                   rew=rng.normal(0, 1, k), cost=(rng.random(k) < 0.3).astype(np.float64), terminated=rng.random(k) < 0.1,
                   truncated=np.full(k, t % 11 == 10), obs_next=rng.standard_normal((k, Do)).astype(np.float32))
        eng.push(ids, row["obs"], row["act"], row["rew"], row["cost"], row["terminated"], row["truncated"], row["obs_next"])
        for e, j in zip(ids, range(k)):
            for key in store:
                store[key][e * sub + t] = row[key][j]
    index = ReplayIndex(rows, sub, store["terminated"] | store["truncated"])
    valid = np.concatenate([e * sub + np.arange(r) for e, r in enumerate(rows)])
    lag = [0.3] if use_lag else []
    for u in range(3):
        idx = rng.choice(valid, B)
        et = rng.standard_normal((B, Da)).astype(np.float32); ep = rng.standard_normal((B, Da)).astype(np.float32)
        sa, sc, _ = o.update(store, index, idx, et, ep, lag if use_lag else [0.0], 1 / 1.3)
        st = eng.sac_update(B, lag, 1 / 1.3, indices=idx, eps_target=et, eps_pi=ep)
        want = {**sa, **sc}
        keys = ["loss/rescaling", "loss/lagrangian", "loss/actor_safety", "loss/alpha_loss", "loss/alpha_value",
                "loss/actor_rew", "loss/actor_total", "loss/q0", "loss/q1", "loss/q_total"]
        for j, kname in enumerate(keys):
            if kname in want:
                w = float(want[kname])
                assert abs(st[j] - w) <= 1e-4 * abs(w) + 1e-5, (u, kname, st[j], w)
    for got, ref in ((eng.sac_get_params(0)[0], o.actor_flat()), (eng.sac_get_params(1)[0], o.critics_flat()),
                     (eng.sac_get_params(2)[0], o.critics_flat(old=True))):
        d = np.abs(got - ref)      # Adam: an entry whose gradient is rounding noise moves by +-lr per step either way
        assert np.quantile(d, 0.99) <= 5e-6 and d.max() <= 3 * 1e-3, (np.quantile(d, 0.99), d.max())
    eng.close()


@pytest.mark.parametrize("Do,Da,H,rows", [(5, 2, 64, [37]), (60, 2, 128, [130, 99, 20]), (12, 4, 256, [5, 3])])
def test_trust_region_pieces_on_odd_sizes_vs_autograd(Do, Da, H, rows):
    """Surrogate / KL gradients and the Hessian-vector product for N that is not a multiple of 16 (and as
    small as 8 rows), wide inputs, several ragged sub-buffers."""
    from fsrl_amd.engine import Engine, EngineConfig
    from oracle.ppo_lag import OnPolicyData
    from oracle.trust_region import CPOConfig, CPOOracle
    from torch.distributions import Independent, Normal, kl_divergence
    rng = np.random.default_rng(H + Do)
    cols = _synthetic(rng, rows, Do, Da, 25)
    eng = Engine(EngineConfig(obs_dim=Do, act_dim=Da, hidden=H, env_num=len(rows), buffer_size=len(rows) * 256,
                              target_kl=None, max_action=1.0))
    o = CPOOracle(CPOConfig(obs_dim=Do, act_dim=Da, hidden=(H, H)))
    torch.manual_seed(H)
    theta = (0.2 * torch.randn(o.n_params)).numpy()
    o.set_params(theta); eng.set_params(theta)
    for t in range(max(rows)):
        ids = [e for e in range(len(rows)) if t < rows[e]]
        eng.push(ids, *[np.stack([cols[k][e][t] for e in ids]) for k in ("obs", "act", "rew", "cost", "term", "trunc",
                                                                         "obs_next")])
    cat = {k: np.concatenate(v) for k, v in cols.items()}
    end = (cat["term"] | cat["trunc"]).copy(); end[np.cumsum(rows) - 1] = True
    data = OnPolicyData(obs=cat["obs"], act=cat["act"], rew=cat["rew"], cost=cat["cost"], terminated=cat["term"],
                        truncated=cat["trunc"], obs_next=cat["obs_next"], end_flag=end)
    pb = o.process(data)
    assert eng.tr_begin(target_kl=0.01, norm_adv=True, cost_limit=10.0) == sum(rows)
    theta2 = theta + (0.02 * torch.randn(o.n_params)).numpy()          # theta != theta_old: exact Hessian
    o.set_params(theta2); eng.set_params(theta2)
    dist = o.actor_dist(pb["obs"])
    ratio = torch.exp(dist.log_prob(pb["act"]) - pb["logp_old"])
    obj = torch.mean(ratio * pb["advs"][..., 0])
    kl = kl_divergence(Independent(Normal(pb["mean_old"], pb["std_old"]), 1), dist).mean()
    og = o.flat_grad(obj, retain_graph=True).numpy()
    klg = o.flat_grad(kl, create_graph=True)

    def close(a, b, rel):
        scale = max(float(np.abs(b).max()), 1e-12)
        assert float(np.abs(a - b).max()) <= rel * scale, (float(np.abs(a - b).max()), scale)
    close(eng.tr_grad(0), og, 3e-5)
    close(eng.tr_grad(2), klg.detach().numpy(), 3e-5)
    v = torch.randn(klg.numel())
    hv = o.flat_grad((klg * v).sum(), retain_graph=True).numpy()
    close(eng.tr_hvp(v.numpy()), hv, 1e-4)
    eng.close()


def test_one_shot_ppo_update_entry_point_matches_the_stepwise_calls():
    """fsrl_ppo_update (begin + passes + end in one C call, caller-provided permutations) == the three-call form."""
    import ctypes as C
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    rng = np.random.default_rng(4)
    rows, Do, Da = [120, 77], 6, 2
    cols = _synthetic(rng, rows, Do, Da, 40)
    outs = []
    for one_shot in (False, True):
        eng = Engine(EngineConfig(obs_dim=Do, act_dim=Da, hidden=64, env_num=2, buffer_size=1024, max_grad_norm=0.5, target_kl=0.5))
        eng.set_params((0.1 * np.random.default_rng(0).standard_normal(eng.n_params)).astype(np.float32))
        for t in range(max(rows)):
            ids = [e for e in range(2) if t < rows[e]]
            eng.push(ids, *[np.stack([cols[k][e][t] for e in ids]) for k in ("obs", "act", "rew", "cost", "term", "trunc", "obs_next")])
        N, B, repeat = sum(rows), 32, 3
        perms = np.stack([np.random.default_rng(9 + k).permutation(N) for k in range(repeat)]).astype(np.int64)
        lag = np.array([0.3])
        if one_shot:
            cap = repeat * (N // B + 1)
            stats = np.empty((cap, _lib.PPO_NSTATS), np.float32); n = C.c_int64(); stopped = C.c_int32()
            P = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
            _lib.check(eng.lib.fsrl_ppo_update(eng._ctx, P(lag, C.c_double), 1 / 1.3, B, repeat, P(perms, C.c_int64), 0,
                                               P(stats, C.c_float), cap, C.byref(n), C.byref(stopped)))
            outs.append((stats[:n.value].copy(), stopped.value, eng.get_params()))
        else:
            stats, stopped = eng.ppo_update(lag, 1 / 1.3, B, repeat, perms=list(perms))
            outs.append((stats.copy(), stopped, eng.get_params()))
        eng.close()
    assert outs[0][0].shape == outs[1][0].shape and np.array_equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1] and np.array_equal(outs[0][2], outs[1][2])


def test_large_minibatch_without_grad_clip_vs_oracle():
    """max_grad_norm off + a minibatch above 512 rows: the fused-Adam instantiation does not apply (the weight gradients of a
    large minibatch come from the split-K kernel), Adam runs as its own launch with a clip coefficient of 1."""
    from fsrl_amd.engine import Engine, EngineConfig
    from oracle.ppo_lag import OnPolicyData, PPOLagConfig, PPOLagOracle
    Do, Da, H, rows, ep, B, repeat = 8, 2, 256, [640, 640], 80, 640, 2
    rng = np.random.default_rng(77)
    cols = _synthetic(rng, rows, Do, Da, ep)
    eng = Engine(EngineConfig(obs_dim=Do, act_dim=Da, hidden=H, env_num=2, buffer_size=4096, max_grad_norm=None, target_kl=None))
    o = PPOLagOracle(PPOLagConfig(obs_dim=Do, act_dim=Da, hidden=(H, H), max_grad_norm=None, target_kl=1e9))
    torch.manual_seed(5)
    theta = (0.15 * torch.randn(o.n_params)).numpy()
    o.set_params(theta); eng.set_params(theta)
    for t in range(max(rows)):
        ids = [e for e in range(2) if t < rows[e]]
        eng.push(ids, *[np.stack([cols[k][e][t] for e in ids]) for k in ("obs", "act", "rew", "cost", "term", "trunc", "obs_next")])
    cat = {k: np.concatenate(v) for k, v in cols.items()}
    end = (cat["term"] | cat["trunc"]).copy(); end[np.cumsum(rows) - 1] = True
    data = OnPolicyData(obs=cat["obs"], act=cat["act"], rew=cat["rew"], cost=cat["cost"], terminated=cat["term"],
                        truncated=cat["trunc"], obs_next=cat["obs_next"], end_flag=end)
    perms = [rng.permutation(len(data)) for _ in range(repeat)]
    lag = np.array([0.4])
    _, ostats, _ = o.update(data, lag, 1 / 1.4, B, repeat, perms=perms)
    stats, stopped = eng.ppo_update(lag, 1 / 1.4, B, repeat, perms=perms)
    assert stopped == -1 and stats.shape == np.asarray(ostats).shape == (4, 11)
    np.testing.assert_allclose(stats, np.asarray(ostats), rtol=3e-5, atol=3e-5)
    d = np.abs(eng.get_params() - o.get_params())
    assert np.quantile(d, 0.999) <= 5e-6 and d.max() <= 1e-4, (np.quantile(d, 0.999), d.max())
    eng.close()


@pytest.mark.parametrize("Do,Da,rows", [
    (33, 8, [1500, 1500, 1111]),      # NU = 3 chunks of one observation group, ragged N (4 111 rows: the last tile is 15 rows)
    (64, 1, [4100]),                  # a full 64-column group, one action column
    (65, 3, [2500, 2500]),            # second observation group with ONE column
    (100, 6, [2100, 2000, 13]),       # two groups, 36 columns in the second (NU = 3)
    (128, 16, [4096]),                # the maximum observation / action widths, the minimum row count of the kernel
    (17, 2, [3000, 3000]),            # two chunks (NU' = 2): two dW1 jobs of two tile rows each
    (4, 5, [5000]),                   # one chunk: all four tile rows of hidden units in ONE dW1 job
])
def test_tile_job_weight_gradients_shape_sweep_vs_autograd(Do, Da, rows):
    """kernels_wgrad3.hpp over the shapes its job table branches on (chunks per observation group 1 / 2 / 3 / 4, one and two groups,
    a ragged last group, 1 .. 16 action columns, N not a multiple of 16) against torch autograd of the oracle's losses
    (cpo.py:177-182, 206-220): the surrogate gradient, the KL gradient away from theta_old and a Hessian-vector product, at
    2e-5 / 5e-5 of the vector's largest entry (1e-4 for the surrogate, see below).  Plans:
    4 = the tile jobs forced, 6 = with half the row splits, 2 = round 5's split-K kernel on the same inputs, 0 = automatic.
    Then one whole CPO update (two critics per weight-gradient launch) under plan 4 against plan 2: same branch of the dual
    solve, statistics within 2e-2 of their scale (downstream of conjugate gradients)."""
    from fsrl_amd.engine import Engine, EngineConfig
    from oracle.ppo_lag import OnPolicyData
    from oracle.trust_region import CPOConfig, CPOOracle
    from torch.distributions import Independent, Normal, kl_divergence
    torch.set_num_threads(4)
    H = 256
    rng = np.random.default_rng(Do * 31 + Da)
    cols = _synthetic(rng, rows, Do, Da, 250)
    eng = Engine(EngineConfig(obs_dim=Do, act_dim=Da, hidden=H, env_num=len(rows), buffer_size=len(rows) * 5120,
                              target_kl=None, max_action=1.0, lr=1e-3))
    o = CPOOracle(CPOConfig(obs_dim=Do, act_dim=Da, hidden=(H, H)))
    torch.manual_seed(Do)
    theta = (0.08 * torch.randn(o.n_params)).numpy()
    o.set_params(theta); eng.set_params(theta); eng.optim_reset()
    for t in range(max(rows)):
        ids = [e for e in range(len(rows)) if t < rows[e]]
        eng.push(ids, *[np.stack([cols[k][e][t] for e in ids]) for k in ("obs", "act", "rew", "cost", "term", "trunc",
                                                                         "obs_next")])
    cat = {k: np.concatenate(v) for k, v in cols.items()}
    end = (cat["term"] | cat["trunc"]).copy(); end[np.cumsum(rows) - 1] = True
    data = OnPolicyData(obs=cat["obs"], act=cat["act"], rew=cat["rew"], cost=cat["cost"], terminated=cat["term"],
                        truncated=cat["trunc"], obs_next=cat["obs_next"], end_flag=end)
    pb = o.process(data)
    assert eng.tr_begin(target_kl=0.01, norm_adv=True, cost_limit=10.0) == sum(rows)
    na = eng.n_actor_params
    moved = theta.copy()
    moved[:na] += (0.01 * rng.standard_normal(na)).astype(np.float32)                # theta != theta_old: exact Hessian
    o.set_params(moved); eng.set_params(moved)
    dist = o.actor_dist(pb["obs"])
    ratio = torch.exp(dist.log_prob(pb["act"]) - pb["logp_old"])
    obj = torch.mean(ratio * pb["advs"][..., 0])
    kl = kl_divergence(Independent(Normal(pb["mean_old"], pb["std_old"]), 1), dist).mean()
    og = o.flat_grad(obj, retain_graph=True).numpy()
    klg = o.flat_grad(kl, create_graph=True)
    v = np.random.default_rng(1).standard_normal(og.size).astype(np.float32)
    hv = o.flat_grad(torch.dot(klg, torch.from_numpy(v)), retain_graph=True).numpy()

    errs = {}

    def close(a, b, rel, what):
        scale = max(float(np.abs(b).max()), 1e-12)
        errs[what] = (float(np.abs(np.asarray(a) - np.asarray(b)).max()) / scale, rel)
    got = {}
    for plan in (4, 6, 2, 0):
        eng.tr_set_plan(0, 0, plan)
        got[plan] = (eng.tr_grad(0), eng.tr_grad(2), eng.tr_hvp(v))
        # the cached-activation product (co-resident kernel at 256 wide: its dz2 phase holds four action dimensions' columns of W3 / V3 in
        # registers and fetches the rest inside the loop -- act_dim 5 .. 16 here)
        close(eng.tr_hvp_cached(v), hv, 5e-5, (plan, "hvp cached"))
        # the surrogate gradient is a sum over mean-zero normalised advantages: it cancels to ~1e-2 of its terms, and the fp32
        # advantage pipelines of the two sides (oracle: numpy, HIP: scan kernels) differ at 1e-7 -- 1e-4 of the result, the
        # same under every plan; the plans among themselves agree to 2e-5 (below)
        close(got[plan][0], og, 1e-4, (plan, "surrogate"))
        close(got[plan][1], klg.detach().numpy(), 2e-5, (plan, "kl"))
        close(got[plan][2], hv, 5e-5, (plan, "hvp"))
        for j, what in enumerate(("surrogate", "kl", "hvp")):
            close(got[plan][j], got[4][j], 2e-5, (plan, what, "vs plan 4"))
    assert all(e <= rel for e, rel in errs.values()), errs
    assert not np.array_equal(got[4][2], got[2][2])                                  # the tile jobs did run
    for a, b in zip(got[0], got[4]):                                                 # automatic = the tile jobs (XCD-aware order: same sums)
        assert np.array_equal(a, b)

    def whole(plan):
        eng.tr_set_plan(0, 0, plan)
        eng.set_params(theta); eng.optim_reset()
        eng.tr_begin(target_kl=0.01, l2_reg=0.001, critic_lr=1e-3, max_backtracks=10, optim_critic_iters=3, cost_limit=10.0)
        return eng.cpo_learn(25.0, 1).copy(), eng.get_params().copy()
    (sa, ta), (sb, tb) = whole(4), whole(2)
    assert np.isfinite(sa).all() and sa[0, 12] == sb[0, 12]                          # loss/optim_case
    sc = np.maximum(np.abs(sb[0]), 1e-3 * np.abs(sb[0]).max())
    bad = np.abs(sa[0] - sb[0]) > 2e-2 * sc
    assert not bad.any(), (np.flatnonzero(bad), sa[0], sb[0])
    d = np.abs(ta - tb)
    assert d.max() <= 5e-3 and d.mean() <= 5e-5, (d.max(), d.mean())
    eng.tr_set_plan(0, 0, 0)
    eng.close()
