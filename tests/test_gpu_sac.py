"""GPU parity of the SAC-Lagrangian update through the C ABI against the golden vectors recorded
from the unmodified reference (sampled indices and rsample noise injected), and against the oracle.
Tolerances (fp32): per-update logged stats 5e-5 rel + 5e-6 abs, parameters after K updates 5e-6 abs."""
import numpy as np
import pytest

from test_oracle_sac import old_final, sac_setup

pytestmark = pytest.mark.gpu

SAC_KEYS = ["loss/rescaling", "loss/lagrangian", "loss/actor_safety", "loss/alpha_loss", "loss/alpha_value",
            "loss/actor_rew", "loss/actor_total", "loss/q0", "loss/q1", "loss/q_total"]


def _engine(cfg, g):
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    eng = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"],
                              hidden_sizes=tuple(cfg["hidden"]), n_critics=2, env_num=cfg["env_num"],
                              buffer_size=cfg["buffer_size"], gamma=cfg["gamma"], target_kl=None))
    eng.sac_init(actor_lr=cfg["actor_lr"], critic_lr=cfg["critic_lr"], alpha_lr=cfg["alpha_lr"], tau=cfg["tau"],
                 alpha=cfg["alpha"], n_step=cfg["n_step"], auto_alpha=cfg["auto_alpha"])
    eng.sac_set_params(g["theta_actor0"], g["theta_critics0"], 0.0)
    rows = g["env_rows"]
    off = np.concatenate([[0], np.cumsum(rows)])
    for t in range(rows.max()):
        ids = [e for e in range(len(rows)) if t < rows[e]]
        sel = np.array([off[e] + t for e in ids])
        ptr, *_ = eng.push(ids, g["st_obs"][sel], g["st_act"][sel], g["st_rew"][sel], g["st_cost"][sel],
                           g["st_terminated"][sel], g["st_truncated"][sel], g["st_obs_next"][sel])
        assert np.array_equal(ptr, g["slots"][sel])
    return eng


@pytest.mark.parametrize("splitk", [0, 1])      # weight gradients: 0 = one workgroup per output tile (batches <= 512 rows), 1 = split-K
# deep3, wide1: layered contexts; c4full: BASELINE configs[3]'s shape at batch 1024 from the unmodified reference (r5)
@pytest.mark.parametrize("name", ["small", "nstep3", "c4", "widths", "deep3", "wide1", "c4full"])
def test_sac_updates_vs_golden(name, splitk):
    g, cfg, ocfg, store, index = sac_setup(name)
    eng = _engine(cfg, g)
    eng.sac_set_plan(splitk)
    a0, _ = eng.sac_get_params(0)
    c0, _ = eng.sac_get_params(1)
    assert np.array_equal(a0, g["theta_actor0"]) and np.array_equal(c0, g["theta_critics0"])
    lag = g["lagrangian"]
    resc = 1.0 / (lag.sum() + 1.0)
    ka = [str(k) for k in g["stats_actor_keys"]]; kc = [str(k) for k in g["stats_critic_keys"]]
    for u in range(cfg["n_updates"]):
        st = eng.sac_update(cfg["batch_size"], lag, resc, indices=g["indices"][u], eps_target=g["eps_target"][u],
                            eps_pi=g["eps_pi"][u])
        want = {**dict(zip(ka, g["stats_actor"][u])), **dict(zip(kc, g["stats_critic"][u]))}
        for j, k in enumerate(SAC_KEYS):
            if k in want:
                assert abs(st[j] - want[k]) <= 5e-5 * abs(want[k]) + 5e-6, (u, k, st[j], want[k])
    th_a, alpha = eng.sac_get_params(0)
    th_c, _ = eng.sac_get_params(1)
    th_t, _ = eng.sac_get_params(2)
    # Adam divides by sqrt(v): entries whose gradient sits at the fp32 rounding-noise level get
    # O(lr) updates of noise-determined sign in ANY fp32 implementation, so a small fraction of
    # entries may differ by up to ~n_updates * lr * 1e-2; the bulk must agree to 5e-6.
    for got, key in ((th_a, "theta_actor_final"), (th_c, "theta_critics_final"), (old_final(g, th_t), "theta_critics_old_final")):
        d = np.abs(got - g[key])
        assert np.quantile(d, 0.99) <= 5e-6 and d.max() <= 5e-4, (key, np.quantile(d, 0.99), d.max())
    assert abs(alpha - float(g["alpha_final"])) < 1e-6
    eng.close()


@pytest.mark.parametrize("batch", [256, 1024])
def test_fused_launch_plan_is_bit_identical_to_the_separate_launches(batch):
    """fsrl_sac_set_plan bits 1 and 2: the sampler and the row gather / the float64 n-step targets as launches of their own (12
    launches per update) against the default (sample + gather in one launch, the critics' tile launch computing its targets itself: 10);
    bit 3: the next update's sample drawn on the side stream while this update runs (rows pushed in between invalidate it);
    bit 4 (r5): sample + gather as ONE launch of their own instead of inside the forward launch of both actors (the default: 9);
    bit 5 (r6): no rider blocks -- by default, at batch 1024, the actor's weight-gradient launch also draws the NEXT update's batch;
    bit 6 (r6): the critics' split-K weight-gradient launch in plain instead of XCD-aware block order.
    The same Philox counters, the same rows, the same float64 operations: statistics and parameters must agree BIT FOR BIT over
    a run of library-RNG updates; with the caller's indices (parity mode) only the n-step fold differs, also bit for bit."""
    g, cfg, ocfg, store, index = sac_setup("c4")
    outs = []
    for plan in (0, 32, 64, 16, 6, 2, 8, 40, 14, 22):        # 0 (r5 default): sample + gather inside the actors' forward launch (9 launches); 16: round 4's 10
        eng = _engine(cfg, g)
        eng.sac_set_plan(plan)
        rows = [eng.sac_update(batch, [0.5], 1 / 1.5, seed=7 if u == 0 else 0).copy() for u in range(12)]
        k = cfg["env_num"]                                  # new rows: the store changed, the prefetched sample must be dropped
        eng.push(np.arange(k), g["st_obs"][:k], g["st_act"][:k], np.ones(k), np.zeros(k), np.zeros(k, bool), np.zeros(k, bool), g["st_obs"][:k])
        rows += [eng.sac_update(batch, [0.5], 1 / 1.5, sync=(u % 2 == 0)) for u in range(6)]
        rows = [r.copy() for r in rows if r is not None]
        lag = g["lagrangian"]
        rows.append(eng.sac_update(cfg["batch_size"], lag, 1.0 / (lag.sum() + 1.0), indices=g["indices"][0],
                                   eps_target=g["eps_target"][0], eps_pi=g["eps_pi"][0]).copy())
        outs.append((np.stack(rows), eng.sac_get_params(0)[0], eng.sac_get_params(1)[0], eng.sac_get_params(2)[0],
                     eng.sac_last_sample(cfg["batch_size"])[0].copy()))
        eng.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)
    assert np.isfinite(outs[0][0]).all()


@pytest.mark.parametrize("name", ["c4", "deep3"])      # deep3: a layered context
def test_sac_perf_mode_runs_and_is_finite(name):
    """indices / noise drawn by the library (perf mode): finite stats, parameters move."""
    g, cfg, ocfg, store, index = sac_setup(name)
    eng = _engine(cfg, g)
    a0, _ = eng.sac_get_params(0)
    for u in range(20):
        st = eng.sac_update(256 if name == "c4" else 64, [0.5], 1 / 1.5, seed=u + 1)
        assert np.isfinite(st).all()
    a1, alpha = eng.sac_get_params(0)
    assert np.abs(a1 - a0).max() > 1e-4 and 0.9 < alpha < 1.0
    mu, sigma = eng.sac_actor_forward(g["st_obs"][:37])
    assert mu.shape == (37, cfg["act_dim"]) and (sigma > 0).all() and np.isfinite(mu).all()
    eng.close()


def test_device_sampling_replays_through_caller_rng_and_is_well_distributed():
    """Library-RNG mode (Philox on the device): (1) the sample it drew, fed back through the
    caller-RNG arguments into a twin context, gives bit-identical statistics and parameters -- the two
    modes share every kernel after sampling, including the n-step chains / end flags computed on the
    device vs the host; (2) rows are uniform over the stored rows; noise is N(0,1)."""
    g, cfg, ocfg, store, index = sac_setup("c4")
    dev, twin = _engine(cfg, g), _engine(cfg, g)
    B = 512
    valid = np.concatenate([e * (cfg["buffer_size"] // cfg["env_num"]) + np.arange(r)
                            for e, r in enumerate(g["env_rows"])])
    all_idx, all_eps = [], []
    for u in range(12):
        dev.sac_update(B, [0.5], 1 / 1.5, seed=11 if u == 0 else 0, sync=False)
        idx, et, ep = dev.sac_last_sample(B)
        assert np.isin(idx, valid).all()
        st_twin = twin.sac_update(B, [0.5], 1 / 1.5, indices=idx, eps_target=et, eps_pi=ep)
        all_idx.append(idx); all_eps += [et, ep]
        if u == 0:
            first = (idx.copy(), et.copy())
    rows = dev.sac_drain()
    assert rows.shape == (12, len(SAC_KEYS)) and np.array_equal(rows[-1], st_twin)
    assert dev.sac_drain().shape[0] == 0
    for which in (0, 1, 2):
        assert np.array_equal(dev.sac_get_params(which)[0], twin.sac_get_params(which)[0])
    idx = np.concatenate(all_idx); eps = np.concatenate([e.ravel() for e in all_eps])
    assert not np.array_equal(all_idx[0], all_idx[1])
    # uniform over stored rows: per-env share within 5 sigma of the binomial expectation
    share = g["env_rows"] / g["env_rows"].sum()
    sub = cfg["buffer_size"] // cfg["env_num"]
    cnt = np.bincount(idx // sub, minlength=cfg["env_num"])
    assert (np.abs(cnt - idx.size * share) <= 5 * np.sqrt(idx.size * share * (1 - share)) + 1).all()
    n = eps.size
    assert abs(eps.mean()) < 5 / np.sqrt(n) and abs(eps.var() - 1) < 5 * np.sqrt(2 / n)
    assert abs((eps**4).mean() - 3) < 0.2 and abs((eps**3).mean()) < 0.1 and np.abs(eps).max() < 7
    # same key + same update counter -> same sample
    again = _engine(cfg, g)
    again.sac_update(B, [0.5], 1 / 1.5, seed=11, sync=False)
    i2, e2, _ = again.sac_last_sample(B)
    assert np.array_equal(i2, first[0]) and np.array_equal(e2, first[1])
    for e in (dev, twin, again):
        e.close()


def test_more_sub_buffers_than_the_samplers_lds_table_holds():
    """600 envs (the samplers keep up to 512 sub-buffers' bookkeeping in LDS and read it from global memory beyond): the launch plans must
    still agree bit for bit at batch 1024 -- rider blocks + folded sample (default) | no riders | sample + gather as one launch of their
    own | sampler and gather as two launches -- and every sampled row must be a stored one.  n_step 3 walks the chains across the
    sub-buffers' write heads (ragged fill: 1 .. 40 rows per env)."""
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    E, sub, Do, Da, B = 600, 64, 11, 3, 1024
    rng = np.random.default_rng(3)
    rows = rng.integers(1, 41, E)
    outs = []
    for plan in (0, 32, 48, 2):
        eng = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=Do, act_dim=Da, hidden_sizes=(64, 64), n_critics=2, env_num=E,
                                  buffer_size=E * sub, gamma=0.99, target_kl=None))
        eng.sac_init(n_step=3)
        r2 = np.random.default_rng(4)
        for t in range(int(rows.max())):
            ids = np.flatnonzero(rows > t)
            k = ids.size
            eng.push(ids, r2.standard_normal((k, Do)).astype(np.float32), np.tanh(r2.standard_normal((k, Da))).astype(np.float32),
                     r2.standard_normal(k), (r2.random(k) < 0.2).astype(np.float64), r2.random(k) < 0.05, r2.random(k) < 0.05,
                     r2.standard_normal((k, Do)).astype(np.float32))
        eng.sac_set_plan(plan)
        st = [eng.sac_update(B, [0.3], 1 / 1.3, seed=5 if u == 0 else 0).copy() for u in range(6)]
        idx = eng.sac_last_sample(B)[0].copy()
        valid = np.concatenate([e * sub + np.arange(rows[e]) for e in range(E)])
        assert np.isin(idx, valid).all()
        outs.append((np.stack(st), idx, eng.sac_get_params(0)[0], eng.sac_get_params(1)[0]))
        eng.close()
    assert np.isfinite(outs[0][0]).all()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)
