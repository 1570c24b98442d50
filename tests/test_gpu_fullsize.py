"""Full-size GPU parity at the sizes BASELINE.json states, through the C ABI, HIP vs the pinned oracle on the same inputs.

  configs[2]  CPO, SafetyPointGoal shape: obs 60 / act 2 / 256x256 / N = 20 000 (20 envs x 1000) / CG 10
              (fsrl/policy/cpo.py:234-351, 353-370)               vs oracle.trust_region.CPOOracle
  configs[1]' TRPO-Lagrangian on the configs[1] shape: obs 8 / act 2 / 256x256 / N = 20 000
              (fsrl/policy/trpo_lag.py:173-251)                   vs oracle.trust_region.TRPOLagOracle
  configs[3]  SAC-Lagrangian, SafetyAntRun shape: obs 33 / act 8 / 256x256 / batch 1024 / n_step 2, replay store
              filled to 1 M rows in HBM (fsrl/policy/sac_lag.py:185-269) vs oracle.sac_lag.SACLagOracle, three updates
              with the caller's indices and rsample noise

The oracles are pinned to fixtures of the unmodified reference at smaller sizes (tests/test_oracle_*.py); here they run
at full size on the host (a few seconds each).  Tolerances are written at each assert.  The CPU side uses 4 torch
threads like the reference's default `thread=4`."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(rng, envs, T, obs_dim, act_dim, ep):
    obs = rng.standard_normal((T + 1, envs, obs_dim)).astype(np.float32)
    act = (0.3 * rng.standard_normal((T, envs, act_dim))).astype(np.float32)
    rew = rng.normal(0.5, 0.5, (T, envs))
    cost = (rng.random((T, envs)) < 0.1).astype(np.float64)
    trunc = np.zeros((T, envs), bool)
    trunc[ep - 1::ep] = True
    return obs, act, rew, cost, np.zeros((T, envs), bool), trunc


def _orth_theta(o, seed):
    """Agent init (ppo_lag_agent.py:147-153 and the CPO / TRPO agents alike): orthogonal W, zero b, sigma_param -0.5."""
    torch.manual_seed(seed)
    parts = []
    for spec in o.specs:
        for name, shape in spec.items():
            if name == "sigma_param":
                parts.append(torch.full(shape, -0.5).reshape(-1))
            elif name.startswith("W"):
                w = torch.empty(shape)
                torch.nn.init.orthogonal_(w)
                parts.append(w.reshape(-1))
            else:
                parts.append(torch.zeros(shape).reshape(-1))
    return torch.cat(parts).numpy()


def _setup_onpolicy(obs_dim, act_dim, hid, ep, lr):
    from fsrl_amd.engine import Engine, EngineConfig
    from oracle.ppo_lag import OnPolicyData
    envs, T = 20, 1000
    rng = np.random.default_rng(7)
    obs, act, rew, cost, term, trunc = _inputs(rng, envs, T, obs_dim, act_dim, ep)
    eng = Engine(EngineConfig(obs_dim=obs_dim, act_dim=act_dim, hidden=hid, env_num=envs, target_kl=None, lr=lr))
    ids = np.arange(envs)
    for t in range(T):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    em = lambda a: np.concatenate([a[:, e] for e in range(envs)])  # noqa: E731   env-major = sample(0) order
    data = OnPolicyData(obs=em(obs[:-1]), act=em(act), rew=em(rew), cost=em(cost), terminated=em(term),
                        truncated=em(trunc), obs_next=em(obs[1:]), end_flag=em(term | trunc))
    return eng, data


def _rel(a, b, floor=1e-3):
    return abs(float(a) - float(b)) / max(abs(float(b)), floor)


def test_cpo_configs2_full_size():
    """BASELINE configs[2]: one CPO repeat (10 critic steps, 2 CG solves of 10 iterations, 22 HVPs, line search) on
    N = 20 000 rows, 256x256, obs 60."""
    from oracle.trust_region import CPOConfig, CPOOracle
    torch.set_num_threads(4)
    eng, data = _setup_onpolicy(60, 2, 256, 1000, 1e-3)
    ocfg = CPOConfig(obs_dim=60, act_dim=2, hidden=(256, 256), optim_critic_iters=10, max_backtracks=10, cost_limit=10.0,
                     l2_reg=0.001, target_kl=0.01)
    o = CPOOracle(ocfg)
    theta = _orth_theta(o, 0)
    o.set_params(theta)
    eng.set_params(theta); eng.optim_reset()
    n = eng.tr_begin(target_kl=0.01, l2_reg=0.001, critic_lr=1e-3, max_backtracks=10, optim_critic_iters=10,
                     cost_limit=10.0)
    assert n == 20000
    st = eng.cpo_learn(25.0, 1)[0]
    pb, rows = o.update(data, 25.0, 1)
    sa, sc = rows[0][:2]
    # process_fn products at N = 20 000 (float64 GAE, full-batch normalisation): 2e-5 abs on normalised advantages
    np.testing.assert_allclose(eng.batch_get("advs"), pb["advs"].numpy(), rtol=0, atol=2e-5)
    keys = ["loss/kl", "loss/entropy", "loss/rew_loss", "loss/cost_loss", "loss/optim_A", "loss/optim_B", "loss/optim_C",
            "loss/optim_Q", "loss/optim_R", "loss/optim_S", "loss/optim_lam", "loss/optim_nu", "loss/optim_case",
            "loss/step_size"]
    got = dict(zip(keys, st[:14]))
    print("cpo full size  hip:", {k: float(v) for k, v in got.items()}, " vf", st[14:].tolist())
    print("cpo full size  oracle:", sa, sc)
    assert int(got["loss/optim_case"]) == int(sa["loss/optim_case"])              # same branch of the dual solve
    np.testing.assert_allclose(got["loss/step_size"], sa["loss/step_size"], rtol=1e-6)   # same backtrack count
    # critic regression after 10 Adam steps: 2e-4 relative
    assert _rel(st[14], sc["loss/vf0"]) <= 2e-4 and _rel(st[15], sc["loss/vf1"]) <= 2e-4, (st[14:], sc)
    # before CG: 2e-5 relative; rew_loss = mean(ratio * A) with normalised advantages at theta_old is a sum of 20 000
    # O(1) terms that cancels to ~1e-7: absolute 2e-6
    for k in ("loss/entropy", "loss/cost_loss", "loss/optim_C"):
        assert _rel(got[k], sa[k]) <= 2e-5 + 1e-6, (k, got[k], sa[k])
    assert abs(got["loss/rew_loss"] - sa["loss/rew_loss"]) <= 2e-6
    # everything downstream of the two CG solves (Q, R, S and what the dual solve derives from them).  fp32 CG amplifies
    # rounding: the reference's own fp32 arithmetic sits 3e-4 .. 6e-4 from the float64 evaluation of the same algorithm
    # (S, nu; measured).  The device accumulates the CG dot products and the split-K partial sums in float64, so it must
    # be (a) within 2e-3 of the fp32 oracle and (b) at least as close to the float64 oracle as the fp32 oracle is
    # (observed: S 8e-5 vs 6e-4, nu 4e-5 vs 3e-4, Q 4.5e-4 vs 4.4e-4).
    o64 = CPOOracle(ocfg, dtype=torch.float64)
    o64.set_params(theta)
    _, rows64 = o64.update(data, 25.0, 1)
    s64 = rows64[0][0]
    print("cpo full size  float64 oracle:", {k: s64[k] for k in ("loss/optim_Q", "loss/optim_R", "loss/optim_S", "loss/optim_nu")})
    # R = g.H^-1 b is a cross term (here |R| << sqrt(Q S): the two gradients are nearly H-orthogonal), so it is measured
    # against sqrt(Q S), the scale its rounding noise has
    qs = float(np.sqrt(sa["loss/optim_Q"] * sa["loss/optim_S"]))
    assert int(s64["loss/optim_case"]) == int(sa["loss/optim_case"])
    for k in ("loss/optim_Q", "loss/optim_R", "loss/optim_S", "loss/optim_A", "loss/optim_B", "loss/optim_lam",
              "loss/optim_nu"):
        floor = qs if k == "loss/optim_R" else 1e-3
        assert _rel(got[k], sa[k], floor) <= 2e-3, (k, got[k], sa[k])
        scale = max(abs(float(s64[k])), floor)
        err_dev, err_ref = abs(float(got[k]) - float(s64[k])) / scale, abs(float(sa[k]) - float(s64[k])) / scale
        assert err_dev <= 1.25 * err_ref + 1e-4, (k, float(got[k]), sa[k], s64[k], err_dev, err_ref)
    th = eng.get_params()
    d = np.abs(th - o.get_params())
    print("cpo theta diff max / mean", d.max(), d.mean())
    assert d.max() <= 2e-3 and d.mean() <= 5e-5
    eng.close()


def test_trpo_full_size():
    """TRPO-Lagrangian at 256x256 / N = 20 000: one repeat = surrogate gradient, CG (10), step size, line search,
    20 critic steps.  Resolves the round-1 observation (profiles/r01_bench_trust.json: vf0 55.9 vs 43.3): with a fresh
    optimiser state on both sides the critic losses agree to 2e-4."""
    from oracle.trust_region import TRPOConfig, TRPOLagOracle
    torch.set_num_threads(4)
    eng, data = _setup_onpolicy(8, 2, 256, 250, 5e-4)
    o = TRPOLagOracle(TRPOConfig(obs_dim=8, act_dim=2, hidden=(256, 256), optim_critic_iters=20))
    theta = _orth_theta(o, 0)
    o.set_params(theta)
    eng.set_params(theta); eng.optim_reset()
    n = eng.tr_begin(target_kl=0.001, critic_lr=5e-4, max_backtracks=10, optim_critic_iters=20)
    assert n == 20000
    st = eng.trpo_learn([0.75], 1 / 1.75, 1)[0]
    _, rows = o.update(data, [0.75], 1 / 1.75, 1)
    so = rows[0][0]
    keys = ["loss/rescaling", "loss/lagrangian", "loss/actor_safety", "loss/actor_rew", "loss/actor_total", "loss/vf0",
            "loss/vf1", "loss/vf_total", "loss/kl", "loss/step_size", "loss/entropy"]
    got = dict(zip(keys, st))
    print("trpo full size hip:", {k: float(v) for k, v in got.items()})
    print("trpo full size oracle:", so)
    for k in keys:
        tol = 2e-3 if k in ("loss/kl", "loss/step_size") else 2e-4
        assert _rel(got[k], so[k]) <= tol + 1e-6, (k, got[k], so[k])
    d = np.abs(eng.get_params() - o.get_params())
    print("trpo theta diff max / mean", d.max(), d.mean())
    assert d.max() <= 2e-3 and d.mean() <= 5e-5
    # the same update again WITHOUT resetting the optimiser: Adam moments of the first run persist, the critic losses
    # move away (this is what tools/bench_trust.py compared with a fresh oracle in round 1)
    eng.set_params(theta)
    eng.tr_begin(target_kl=0.001, critic_lr=5e-4, max_backtracks=10, optim_critic_iters=20)
    st2 = eng.trpo_learn([0.75], 1 / 1.75, 1)[0]
    assert _rel(st2[5], so["loss/vf0"]) > 1e-2
    eng.close()


def test_sac_configs3_full_size():
    """BASELINE configs[3]: 1 M-row store in HBM (10 sub-buffers x 100 000, 1000-step episodes), 256x256, batch 1024,
    n_step 2; three updates with the caller's indices / noise against the oracle on a host copy of the same store."""
    from fsrl_amd import _lib
    from fsrl_amd.engine import Engine, EngineConfig
    from oracle.sac_lag import ReplayIndex, SACConfig, SACLagOracle
    torch.set_num_threads(4)
    Do, Da, H, E, B, ROWS = 33, 8, 256, 10, 1024, 1_000_000
    T = ROWS // E
    rng = np.random.default_rng(3)
    eng = Engine(EngineConfig(algo=_lib.ALGO_SAC_LAG, obs_dim=Do, act_dim=Da, hidden=H, n_critics=2, env_num=E,
                              buffer_size=ROWS, gamma=0.99, target_kl=None))
    eng.sac_init()
    o = SACLagOracle(SACConfig(obs_dim=Do, act_dim=Da, hidden=(H, H)))
    torch.manual_seed(0)

    def orth(spec):
        parts = []
        for name, shape in spec.items():
            if name.startswith("W"):
                w = torch.empty(shape)
                torch.nn.init.orthogonal_(w)
                parts.append(w.reshape(-1))
            else:
                parts.append(torch.zeros(shape).reshape(-1))
        return torch.cat(parts).numpy()
    th_a = orth(o.aspec)
    th_c = np.concatenate([orth(o.cspec), orth(o.cspec)])
    eng.sac_set_params(th_a, th_c, 0.0)
    o.set_params(th_a, th_c, 0.0)
    obs = rng.standard_normal((T + 1, E, Do)).astype(np.float32)
    act = np.tanh(rng.standard_normal((T, E, Da))).astype(np.float32)
    rew = rng.normal(0.5, 0.5, (T, E))
    cost = (rng.random((T, E)) < 0.1).astype(np.float64)
    trunc = np.zeros((T, E), bool)
    trunc[999::1000] = True
    term = rng.random((T, E)) < 0.0005            # a few true terminations: the n-step target must be masked there
    ids = np.arange(E)
    for t in range(T):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    eng.sync()
    assert len(eng) == ROWS
    slot = lambda x: np.ascontiguousarray(np.swapaxes(x, 0, 1)).reshape((E * T, ) + x.shape[2:])  # noqa: E731
    store = {"obs": slot(obs[:-1]), "obs_next": slot(obs[1:]), "act": slot(act), "rew": slot(rew), "cost": slot(cost),
             "terminated": slot(term)}
    index = ReplayIndex([T] * E, T, slot(term | trunc))
    lag, resc = [0.3], 1.0 / 1.3
    keys = ["loss/rescaling", "loss/lagrangian", "loss/actor_safety", "loss/alpha_loss", "loss/alpha_value",
            "loss/actor_rew", "loss/actor_total", "loss/q0", "loss/q1", "loss/q_total"]
    r2 = np.random.default_rng(11)
    for u in range(3):
        idx = r2.integers(0, ROWS, B)
        if u == 0:                                   # make sure episode ends, sub-buffer tails and terminations are hit
            idx[:4] = [998, 999, T - 1, ROWS - 1]
            idx[4:8] = np.flatnonzero(store["terminated"])[:4]
        et = r2.standard_normal((B, Da)).astype(np.float32)
        ep = r2.standard_normal((B, Da)).astype(np.float32)
        st = eng.sac_update(B, lag, resc, indices=idx, eps_target=et, eps_pi=ep)
        sa, sc, _ = o.update(store, index, idx, et, ep, lag, resc)
        want = {**sa, **sc}
        for j, k in enumerate(keys):
            # logged statistics of one update: 5e-5 relative + 5e-6 absolute (the fixture tests' tolerance)
            assert abs(st[j] - want[k]) <= 5e-5 * abs(want[k]) + 5e-6, (u, k, float(st[j]), want[k])
    for got, ref in ((eng.sac_get_params(0)[0], o.actor_flat()), (eng.sac_get_params(1)[0], o.critics_flat()),
                     (eng.sac_get_params(2)[0], o.critics_flat(old=True))):
        d = np.abs(got - ref)
        # Adam divides by sqrt(v): entries whose gradient is rounding noise take O(lr) steps of noise-determined sign in
        # any fp32 implementation -- the bulk agrees to 5e-6, the tail is bounded by 3 updates x lr 1e-3
        assert np.quantile(d, 0.99) <= 5e-6 and d.max() <= 3.1e-3, (np.quantile(d, 0.99), d.max())
    assert abs(eng.sac_get_params(0)[1] - float(o.alpha)) < 1e-6
    eng.close()


def test_hessian_vector_product_properties_at_full_size():
    """Size-independent properties of the exact R-op Hessian-vector product of the mean KL (cpo.py:177-182) at configs[2]'s
    size (N = 20 000 rows, 256x256, obs 60), no oracle needed: linearity H(a v + b w) = a Hv + b Hw, symmetry v.Hw = w.Hv,
    positive semi-definiteness at theta_old (the Hessian of a KL at its minimum is the Fisher matrix), and -- after the actor
    has moved away from theta_old -- still symmetric (the exact Hessian, not the Fisher approximation)."""
    from oracle.trust_region import CPOConfig, CPOOracle
    eng, _ = _setup_onpolicy(60, 2, 256, 1000, 1e-3)
    o = CPOOracle(CPOConfig(obs_dim=60, act_dim=2, hidden=(256, 256)))
    theta = _orth_theta(o, 0)
    eng.set_params(theta); eng.optim_reset()
    n = eng.tr_begin(target_kl=0.01, l2_reg=0.001, critic_lr=1e-3, max_backtracks=10, optim_critic_iters=10, cost_limit=10.0)
    assert n == 20000
    na = eng.n_actor_params
    rng = np.random.default_rng(3)
    v, w = rng.standard_normal(na).astype(np.float32), rng.standard_normal(na).astype(np.float32)

    def check(tag):
        hv, hw = eng.tr_hvp(v).astype(np.float64), eng.tr_hvp(w).astype(np.float64)
        hvw = eng.tr_hvp((0.5 * v - 2.0 * w).astype(np.float32)).astype(np.float64)
        scale = np.abs(hvw).max()
        assert np.abs(hvw - (0.5 * hv - 2.0 * hw)).max() <= 2e-5 * scale, (tag, np.abs(hvw - (0.5 * hv - 2.0 * hw)).max(), scale)
        a, b = float(v.astype(np.float64) @ hw), float(w.astype(np.float64) @ hv)
        assert abs(a - b) <= 2e-5 * np.sqrt(float(v @ hv) * float(w @ hw)), (tag, a, b)
        return float(v @ hv), float(w @ hw)
    qv, qw = check("theta_old")
    assert qv > 0 and qw > 0                                        # Fisher matrix at theta_old: v.Hv = E[(J v)^2 / var] >= 0
    moved = theta.copy()
    moved[:na] += (0.02 * rng.standard_normal(na)).astype(np.float32)           # actor away from the theta of mean_old
    eng.set_params(moved)
    check("moved")
    eng.close()


@pytest.mark.parametrize("obs_dim,hid,T", [(60, 256, 1000), (8, 256, 300), (24, 128, 700)])
def test_full_batch_kernel_plans_are_bit_identical(obs_dim, hid, T):
    """The full-batch path's three tile / HVP kernel plans (fsrl_tr_set_plan) must agree BIT FOR BIT: 16-row tiles + the HVP kernel
    that recomputes everything (round 2's path) | mixed 32- / 16-row tiles without the HVP cache | mixed tiles + the
    theta-only activations of the KL Hessian product computed once per conjugate-gradient solve and read back (default).
    A row's arithmetic does not depend on its tile's height, relu'(z) is read off h > 0, and the per-tile statistics keep
    their 16-row slots.  Checked on the building blocks (three gradients, line-search statistics, one HVP) and on whole
    CPO / TRPO-Lag updates of two repeats (22 + 11 HVPs per repeat through the cached kernel)."""
    from fsrl_amd.engine import Engine, EngineConfig
    envs = 20
    rng = np.random.default_rng(11)
    obs, act, rew, cost, term, trunc = _inputs(rng, envs, T, obs_dim, 2, 250)
    eng = Engine(EngineConfig(obs_dim=obs_dim, act_dim=2, hidden=hid, env_num=envs, target_kl=None, lr=1e-3))
    ids = np.arange(envs)
    for t in range(T):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    theta = (0.1 * np.random.default_rng(3).standard_normal(eng.n_params)).astype(np.float32)
    v = np.random.default_rng(4).standard_normal(eng.n_actor_params).astype(np.float32)

    def run(tile_rows, hvp, wgrad=0):
        out = {}
        eng.tr_set_plan(tile_rows, hvp, wgrad)
        eng.set_params(theta); eng.optim_reset()
        assert eng.tr_begin(target_kl=0.01, l2_reg=0.001, critic_lr=1e-3, max_backtracks=10, optim_critic_iters=3,
                            cost_limit=10.0) == envs * T
        for w in range(3):
            out[f"grad{w}"] = eng.tr_grad(w)
        out["eval"] = eng.tr_eval()
        out["hvp"] = eng.tr_hvp(v)
        out["hvp_cached"] = eng.tr_hvp_cached(v[::-1].copy())        # the cached-activation kernel (r5: co-resident at 256 wide)
        out["cpo"] = eng.cpo_learn(25.0, 2).copy()
        out["theta_cpo"] = eng.get_params().copy()
        eng.set_params(theta); eng.optim_reset()
        eng.tr_begin(target_kl=0.001, critic_lr=1e-3, max_backtracks=10, optim_critic_iters=3)
        out["trpo"] = eng.trpo_learn([0.4], 1 / 1.4, 2).copy()
        out["theta_trpo"] = eng.get_params().copy()
        return out

    # the tile / HVP plans are bit-identical under EVERY weight-gradient kernel (wgrad 4: r6's tile jobs, two workgroups per CU,
    # wherever they apply -- 256-wide layers, >= 4096 rows; wgrad 2: round 5's split-K kernel; wgrad 3: the one-pass streaming
    # kernel); wgrad 0 = automatic: the tile jobs in XCD-aware block order (same sums as the plain order of wgrad 4)
    refs = {}
    for wg in (3, 2, 4):
        ref = refs[wg] = run(16, 1, wg)
        assert np.isfinite(ref["cpo"]).all() and np.isfinite(ref["trpo"]).all() and np.abs(ref["hvp"]).max() > 0
        # (32, 3): round 4's one-workgroup-per-CU kernels; (0, 0): round 5's co-resident pairs where they apply (256 wide)
        # tile_rows + 64: the critics' steps on the compute stream, behind each other, instead of beside the actor's step (r5)
        # tile_rows + 128: read-backs by hipMemcpyAsync + hipStreamSynchronize instead of the polled completion words (r6)
        for plan in ((0, 2, wg), (0, 0, wg), (16, 0, wg), (32, 3, wg), (32, 0, wg), (0, 3, wg), (64, 0, wg), (96, 3, wg), (128, 0, wg)) + (((0, 0, 1), ) if wg == 2 else ()):
            got = run(*plan)
            for k in ref:
                assert np.array_equal(ref[k], got[k]), (plan, k, np.abs(np.asarray(ref[k], np.float64) - got[k]).max())
    # the two weight-gradient kernels add the rows up in different orders (fp32 MFMA chains over 4 vs 16 interleaved row
    # classes, 32-64 vs <= 24 partials in float64): the building blocks agree to 2e-5 of the vector's largest entry (the
    # tolerance of the autograd comparison in test_gpu_trust.py); whole updates -- conjugate gradients amplify summation
    # order, DESIGN "conditioning note" -- at the fixture tolerances of test_gpu_trust.py (2e-2 on what is downstream of CG)
    auto = run(0, 0, 0)
    same_as = refs[4]                # below 256 wide / 4096 rows every plan is round 5's kernel and refs[4] == refs[2] (checked below)
    for k in auto:
        assert np.array_equal(auto[k], same_as[k]), ("automatic weight-gradient plan", k)
    for a, b in ((refs[3], refs[2]), (refs[4], refs[2])):
      if hid == 256 and envs * T >= 4096:
        for k in ("grad0", "grad1", "grad2", "hvp"):
            scale = max(float(np.abs(b[k]).max()), 1e-12)
            assert float(np.abs(a[k] - b[k]).max()) <= 2e-5 * scale, (k, float(np.abs(a[k] - b[k]).max()), scale)
        assert not np.array_equal(a["hvp"], b["hvp"])                    # the new kernel did run
        np.testing.assert_allclose(a["eval"], b["eval"], rtol=0, atol=0)
        # first repeat: same branch of the dual solve, same number of backtracks, everything else within 2e-2 of its scale (the
        # reference's own error band on quantities downstream of conjugate gradients); the second repeat starts from thetas
        # that differ at the 1e-4 level and is only required to stay finite and to take the same branch
        assert a["cpo"][0, 12] == b["cpo"][0, 12] and a["cpo"][1, 12] == b["cpo"][1, 12]          # loss/optim_case
        np.testing.assert_allclose(a["cpo"][0, 13], b["cpo"][0, 13], rtol=1e-6)                   # loss/step_size
        for k in ("cpo", "trpo"):
            sc = np.maximum(np.abs(b[k][0]), 1e-3 * np.abs(b[k][0]).max())
            bad = np.abs(a[k][0] - b[k][0]) > 2e-2 * sc
            assert not bad.any(), (k, np.flatnonzero(bad), a[k][0], b[k][0])
            assert np.isfinite(a[k]).all()
        for k in ("theta_cpo", "theta_trpo"):
            d = np.abs(a[k] - b[k])
            assert d.max() <= 5e-3 and d.mean() <= 5e-5, (k, d.max(), d.mean())
      else:
        for k in a:
            assert np.array_equal(a[k], b[k]), k                         # below the new kernels' range nothing changes
    eng.tr_set_plan(0, 0, 0)
    eng.close()


@pytest.mark.parametrize("at_theta_old", [False, True])
@pytest.mark.parametrize("obs_dim,T", [(60, 1000), (8, 1000), (33, 163)])
def test_co_resident_kernels_match_one_workgroup_per_cu(obs_dim, T, at_theta_old):
    """Round 5's co-resident tile kernels (two 512-thread workgroups per CU, two time-multiplexed LDS slots; kernels_fbco.hpp)
    against round 4's one-1024-thread-workgroup-per-CU kernels on the building blocks, BIT FOR BIT, for the persistent launches
    (tiles drawn from a device counter), for the static grid and for forced tile mixes (all whole 32-row tiles, all 16-row
    tiles, one 32-row tile, different mixes for the two kernels): the three gradients (SUR,
    SUR, KL heads), the line-search statistics (EVAL), the critics' regression step (VF, through one CPO repeat) and the cached
    Hessian-vector product.  N = 20 000 (BASELINE configs[2] / configs[1] shapes) and N = 3 260 (partial tiles, one round).
    at_theta_old: the products are taken at the theta of tr_begin -- the Gauss-Newton form (r5: the co-resident kernel skips the
    dz2 / dout terms, round 4's kernel computes them as exact zeros: the same bits) -- and once more with that form switched off
    (hvp plan + 4), which must agree to rounding: the terms it drops are at most the device's own 1e-7 noise around an exact zero."""
    from fsrl_amd.engine import Engine, EngineConfig
    envs = 20
    rng = np.random.default_rng(21)
    obs, act, rew, cost, term, trunc = _inputs(rng, envs, T, obs_dim, 2, 250)
    eng = Engine(EngineConfig(obs_dim=obs_dim, act_dim=2, hidden=256, env_num=envs, target_kl=None, lr=1e-3))
    ids = np.arange(envs)
    for t in range(T):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    theta = (0.1 * np.random.default_rng(3).standard_normal(eng.n_params)).astype(np.float32)
    moved = theta.copy()
    moved[:eng.n_actor_params] += (0.01 * np.random.default_rng(5).standard_normal(eng.n_actor_params)).astype(np.float32)
    v = np.random.default_rng(4).standard_normal(eng.n_actor_params).astype(np.float32)
    N = envs * T

    def run(tile_rows, hvp, split=(-1, -1)):
        out = {}
        eng.tr_set_plan(tile_rows, hvp, 0)
        eng.tr_set_tile_split(*split)
        eng.set_params(theta); eng.optim_reset()
        assert eng.tr_begin(target_kl=0.01, l2_reg=0.001, critic_lr=1e-3, max_backtracks=10, optim_critic_iters=2, cost_limit=10.0) == N
        if not at_theta_old:
            eng.set_params(moved)                               # theta != theta_old: every term of the R-op is live
        for w in range(3):
            out[f"grad{w}"] = eng.tr_grad(w)
        out["eval"] = eng.tr_eval()
        out["hvp"] = eng.tr_hvp(v)
        out["hvp_cached"] = eng.tr_hvp_cached(v)
        out["hvp_cached2"] = eng.tr_hvp_cached(v[::-1].copy())
        out["cpo"] = eng.cpo_learn(25.0, 1).copy()
        out["theta"] = eng.get_params().copy()
        return out
    ref = run(32, 3)
    assert np.array_equal(ref["hvp"], ref["hvp_cached"]) and np.abs(ref["hvp"]).max() > 0 and np.isfinite(ref["cpo"]).all()
    bad = []
    # (-1, -1): one workgroup per tile (the default), automatic tile mix; (-2, -2): PERSISTENT workgroups drawing tiles from the device counter
    for split in ((-1, -1), (-2, -2), (N // 32, N // 32), (0, 0), (1, 1), (300, 100)):
        got = run(0, 0, split)
        for k in ref:
            if not np.array_equal(ref[k], got[k]):
                d = np.abs(np.asarray(ref[k], np.float64) - got[k])
                bad.append((split, k, float(d.max()), int((d > 0).sum()), int(np.size(d))))
    if at_theta_old:
        full = run(0, 4)                                        # the same co-resident kernels, Gauss-Newton form off
        for k in ("hvp", "hvp_cached", "hvp_cached2"):
            scale = float(np.abs(full[k]).max())
            err = float(np.abs(full[k] - ref[k]).max())
            assert 0 < scale and err <= 2e-5 * scale, (k, err, scale)
        # (on this device the two usually agree to the BIT: process_fn's forward and the product's forward add the same numbers
        # in the same order, so mu - mean_old is already an exact zero; the bench legs show that the switch does switch work)
    eng.tr_set_tile_split(-1, -1)
    eng.tr_set_plan(0, 0, 0)
    eng.close()
    assert not bad, bad


def test_streaming_weight_gradients_vs_autograd_at_full_size():
    """The weight-gradient kernels (256-wide layers, N = 20 000, obs 60: four 16-column chunks of dW1 with a ragged last one) against torch
    autograd of the oracle's losses on the same batch: the two surrogate gradients, the KL gradient away from theta_old, and
    three Hessian-vector products of the mean KL (cpo.py:177-182, 206-220), for both weight-gradient kernels.  2e-5 / 5e-5 of
    the vector's largest entry, the bars of the small-fixture test (tests/test_gpu_trust.py).  The critics' gradients take the
    same kernel with two networks per launch: test_cpo_configs2_full_size / test_trpo_full_size compare their losses after
    the Adam steps with the oracle's."""
    from oracle.trust_region import CPOConfig, CPOOracle
    from torch.distributions import Independent, Normal, kl_divergence
    torch.set_num_threads(4)
    eng, data = _setup_onpolicy(60, 2, 256, 1000, 1e-3)
    o = CPOOracle(CPOConfig(obs_dim=60, act_dim=2, hidden=(256, 256), cost_limit=10.0))
    theta = _orth_theta(o, 0)
    rng = np.random.default_rng(5)
    theta = theta + (0.02 * rng.standard_normal(theta.size)).astype(np.float32)      # sigma, biases and W3 away from their init
    o.set_params(theta); eng.set_params(theta); eng.optim_reset()
    assert eng.tr_begin(target_kl=0.01, norm_adv=True, cost_limit=10.0) == 20000
    pb = o.process(data)
    moved = theta.copy()
    na = eng.n_actor_params
    moved[:na] += (0.01 * rng.standard_normal(na)).astype(np.float32)                # theta != theta_old: exact Hessian
    o.set_params(moved); eng.set_params(moved)
    dist = o.actor_dist(pb["obs"])
    ratio = torch.exp(dist.log_prob(pb["act"]) - pb["logp_old"])
    obj = torch.mean(ratio * pb["advs"][..., 0])
    csur = torch.mean(ratio * pb["advs"][..., 1])
    kl = kl_divergence(Independent(Normal(pb["mean_old"], pb["std_old"]), 1), dist).mean()
    og = o.flat_grad(obj, retain_graph=True).numpy()
    ob = o.flat_grad(-csur, retain_graph=True).numpy()
    klg = o.flat_grad(kl, create_graph=True)

    def close(a, b, rel):
        scale = max(float(np.abs(b).max()), 1e-12)
        err = float(np.abs(np.asarray(a) - np.asarray(b)).max())
        assert err <= rel * scale, (err, scale)
    for plan in (3, 2, 5, 0):          # streaming | round 5's split-K | r6's tile jobs with half the row splits | r6 default
        eng.tr_set_plan(0, 0, plan)
        close(eng.tr_grad(0), og, 2e-5)
        close(eng.tr_grad(1), ob, 2e-5)
        close(eng.tr_grad(2), klg.detach().numpy(), 2e-5)
        for k in range(3):
            v = np.random.default_rng(k).standard_normal(og.size).astype(np.float32)
            hv = o.flat_grad(torch.dot(klg, torch.from_numpy(v)), retain_graph=True).numpy()
            close(eng.tr_hvp(v), hv, 5e-5)
    eng.tr_set_plan(0, 0, 0)
    eng.close()


def test_cpo_minibatched_at_full_size():
    """configs[2] with batch_size below the buffer: N = 20 000 rows, Batch.split(6000, merge_last=True) -> minibatches of
    6 000 / 6 000 / 8 000 rows of one permutation (cpo.py:357-358).  Every minibatch runs the mixed 32 / 16-row tile plan on
    a row VIEW of the permuted copy (offsets that are not multiples of 32 rows x obs_dim floats).  First minibatch against
    the oracle at the full-size bars; the later ones must agree on the dual-solve branch and stay finite."""
    from oracle.trust_region import CPOConfig, CPOOracle
    torch.set_num_threads(4)
    eng, data = _setup_onpolicy(60, 2, 256, 1000, 1e-3)
    ocfg = CPOConfig(obs_dim=60, act_dim=2, hidden=(256, 256), optim_critic_iters=4, max_backtracks=10, cost_limit=10.0,
                     l2_reg=0.001, target_kl=0.01)
    o = CPOOracle(ocfg)
    theta = _orth_theta(o, 0)
    o.set_params(theta)
    eng.set_params(theta); eng.optim_reset()
    assert eng.tr_begin(target_kl=0.01, l2_reg=0.001, critic_lr=1e-3, max_backtracks=10, optim_critic_iters=4,
                        cost_limit=10.0) == 20000
    perm = np.random.default_rng(5).permutation(20000)
    st = eng.cpo_learn(25.0, 1, batch_size=6000, perms=[perm])
    _, rows = o.update(data, 25.0, 1, perms=[perm], batch_size=6000)
    assert st.shape == (3, 17) and len(rows) == 3 and np.isfinite(st).all()
    keys = ["loss/kl", "loss/entropy", "loss/rew_loss", "loss/cost_loss", "loss/optim_A", "loss/optim_B", "loss/optim_C",
            "loss/optim_Q", "loss/optim_R", "loss/optim_S", "loss/optim_lam", "loss/optim_nu", "loss/optim_case",
            "loss/step_size"]
    for r in range(3):
        assert int(st[r, 12]) == int(rows[r][0]["loss/optim_case"]), (r, st[r], rows[r][0])
    sa, sc = rows[0][:2]
    got = dict(zip(keys, st[0, :14]))
    np.testing.assert_allclose(got["loss/step_size"], sa["loss/step_size"], rtol=1e-6)
    assert _rel(st[0, 14], sc["loss/vf0"]) <= 2e-4 and _rel(st[0, 15], sc["loss/vf1"]) <= 2e-4, (st[0, 14:], sc)
    qs = float(np.sqrt(abs(sa["loss/optim_Q"] * sa["loss/optim_S"])))
    for k in ("loss/optim_Q", "loss/optim_R", "loss/optim_S", "loss/optim_nu", "loss/cost_loss", "loss/optim_C"):
        floor = qs if k == "loss/optim_R" else 1e-3
        assert _rel(got[k], sa[k], floor) <= 3e-3, (k, got[k], sa[k])
    eng.close()
