"""Grouped PPO-Lagrangian updates (fsrl_group_*): k agents of one shape stepped in lock step, every launch of the minibatch
step carrying all members.  Per member the update must be the single-agent update: bit-identical when the tile shape of
the fused kernel agrees (the group picks 4-row tiles only while ALL members fit the chip in one round), within the fp32
tolerances of the golden tests otherwise."""
import numpy as np
import pytest

from helpers import ppo_case
from test_gpu_ppo import _engine, _push_golden, _rescale

pytestmark = pytest.mark.gpu


def _members(cfg, g, k, **over):
    """k engines on the golden case's store; member i starts from theta0 perturbed by seed i and uses its own multipliers"""
    engs, thetas, lags = [], [], []
    for i in range(k):
        eng = _engine(cfg, **over)
        th = g["theta0"] + (0.01 * np.random.default_rng(100 + i).standard_normal(g["theta0"].size)).astype(np.float32) * (i > 0)
        eng.set_params(th)
        if eng.cfg.rew_norm:
            eng.ret_rms_set(g["ret_rms0"])
        _push_golden(eng, g)
        engs.append(eng); thetas.append(th); lags.append(g["lagrangian"] * (1.0 + 0.5 * i))
    return engs, thetas, lags


@pytest.mark.parametrize("name,k,over", [("c1", 3, {}), ("c2", 2, {}), ("tiny", 4, {}), ("earlystop", 3, {}),
                                         ("c1", 3, {"max_grad_norm": None}),
                                         ("rewnorm_recompute", 2, {}), ("unbounded", 2, {})])      # the on-policy options, grouped
def test_grouped_update_equals_member_by_member(name, k, over):
    from fsrl_amd.engine import EngineGroup
    cfg, g = ppo_case(name)
    R, B = cfg["repeat"], cfg["batch_size"]
    n = len(g["indices"])
    rng = np.random.default_rng(5)
    perms = [[rng.permutation(n) for _ in range(R)] for _ in range(k)]
    for j, pj in enumerate(g["perms"][:R]):             # member 0 replays the golden case (all the passes it recorded)
        perms[0][j] = pj
    # ---- one by one
    solo, thetas, lags = _members(cfg, g, k, **over)
    want = []
    for i, eng in enumerate(solo):
        st, stop = eng.ppo_update(lags[i], _rescale(lags[i]), B, R, perms=perms[i])
        st2, _ = eng.ppo_update(lags[i], _rescale(lags[i]), B, R, perms=perms[i])       # Adam state carried over
        want.append((st, stop, st2, eng.get_params()))
        eng.close()
    # ---- grouped
    engs, _, _ = _members(cfg, g, k, **over)
    grp = EngineGroup(engs)
    resc = [_rescale(l) for l in lags]
    st_a, stop_a = grp.ppo_update(np.stack(lags), resc, B, R, perms=perms)
    st_b, _ = grp.ppo_update(np.stack(lags), resc, B, R, perms=perms)
    same_shape = k == 1          # a lone member follows the single-agent tile rule step by step; k >= 2 runs 16-row tiles
    for i in range(k):
        st, stop, st2, th = want[i]
        assert stop_a[i] == stop, (i, stop_a, stop)
        assert st_a[i].shape == st.shape and st_b[i].shape == st2.shape
        if same_shape:
            assert np.array_equal(st_a[i], st) and np.array_equal(st_b[i], st2) and np.array_equal(engs[i].get_params(), th)
        else:
            np.testing.assert_allclose(st_a[i], st, rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(st_b[i], st2, rtol=2e-4, atol=2e-4)
            # 16-row vs 4-row tiles round differently; two updates of clipped / sign-sensitive Adam steps amplify that on
            # the few entries whose gradient is rounding noise (an Adam step there is +-lr whatever the magnitude): 99 % of
            # the entries agree to 1 % of one learning-rate step, the tail stays below a fifth of a step
            # (without the gradient-norm clip the early Adam steps are full +-lr steps on every entry: 5 % / one step)
            bulk, tail = (0.05, 1.0) if "max_grad_norm" in over else (0.01, 0.2)
            d = np.abs(engs[i].get_params() - th)
            assert np.quantile(d, 0.99) <= bulk * cfg["lr"] and d.max() <= tail * cfg["lr"], (np.quantile(d, 0.99), d.max())
    if not over:                                        # member 0 == the reference's golden update
        np.testing.assert_allclose(st_a[0], g["stats"], rtol=2e-5, atol=2e-5)
    grp.close()
    for e in engs:
        e.close()


def test_group_of_one_is_the_single_agent_update_bit_for_bit():
    from fsrl_amd.engine import EngineGroup
    cfg, g = ppo_case("c2")
    lag = g["lagrangian"]
    a = _engine(cfg); a.set_params(g["theta0"]); _push_golden(a, g)
    b = _engine(cfg); b.set_params(g["theta0"]); _push_golden(b, g)
    sa, _ = a.ppo_update(lag, _rescale(lag), cfg["batch_size"], cfg["repeat"], perms=g["perms"])
    grp = EngineGroup([b])
    sb, _ = grp.ppo_update([lag], [_rescale(lag)], cfg["batch_size"], cfg["repeat"], perms=[g["perms"]])
    assert np.array_equal(sa, sb[0]) and np.array_equal(a.get_params(), b.get_params())
    # a member's own update still works while grouped (it runs on the group's stream) ...
    sc, _ = b.ppo_update(lag, _rescale(lag), cfg["batch_size"], cfg["repeat"], perms=g["perms"])
    sd, _ = a.ppo_update(lag, _rescale(lag), cfg["batch_size"], cfg["repeat"], perms=g["perms"])
    assert np.array_equal(sc, sd)
    grp.close()
    # ... and after the group is gone
    se, _ = b.ppo_update(lag, _rescale(lag), cfg["batch_size"], cfg["repeat"], perms=g["perms"])
    sf, _ = a.ppo_update(lag, _rescale(lag), cfg["batch_size"], cfg["repeat"], perms=g["perms"])
    assert np.array_equal(se, sf)
    a.close(); b.close()


def test_members_with_different_batch_lengths_and_shape_checks():
    """N_i may differ (episodes end at different times): members with fewer minibatches sit out the tail steps."""
    from fsrl_amd.engine import Engine, EngineConfig, EngineGroup
    rng = np.random.default_rng(0)
    Do, Da, H = 8, 2, 64
    lens = [300, 212, 431]
    engs, solo = [], []
    for which in (engs, solo):
        for i, T in enumerate(lens):
            e = Engine(EngineConfig(obs_dim=Do, act_dim=Da, hidden=H, env_num=2, max_grad_norm=0.5, target_kl=None))
            r = np.random.default_rng(10 + i)
            e.set_params((0.1 * r.standard_normal(e.n_params)).astype(np.float32))
            obs = r.standard_normal((T + 1, 2, Do)).astype(np.float32)
            for t in range(T):
                e.push([0, 1], obs[t], 0.3 * r.standard_normal((2, Da)).astype(np.float32), r.normal(0.5, 0.5, 2),
                       (r.random(2) < 0.1).astype(np.float64), [False, False], [t == T - 1] * 2, obs[t + 1])
            which.append(e)
    grp = EngineGroup(engs)
    perms = [[rng.permutation(2 * T) for _ in range(2)] for T in lens]
    lags = np.array([[0.2], [0.5], [0.9]])
    resc = [1 / 1.2, 1 / 1.5, 1 / 1.9]
    st, stop = grp.ppo_update(lags, resc, 128, 2, perms=perms)
    for i, e in enumerate(solo):
        s1, _ = e.ppo_update(lags[i], resc[i], 128, 2, perms=perms[i])
        assert st[i].shape == s1.shape
        np.testing.assert_allclose(st[i], s1, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(engs[i].get_params(), e.get_params(), rtol=0, atol=5e-6)
    other = Engine(EngineConfig(obs_dim=Do, act_dim=Da, hidden=128, env_num=2))
    with pytest.raises(AssertionError):
        EngineGroup([solo[0], other])                   # one network shape per group
    with pytest.raises(AssertionError):
        EngineGroup([engs[0]])                          # already a member of a group
    grp.close()
    for e in engs + solo + [other]:
        e.close()


def test_policy_group_trains_k_seeds_in_lock_step(tmp_path):
    """Facade level: PolicyGroup.update == PPOLagrangian.update per member (same engine calls through the group), and the
    example's grouped loop runs end to end."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "train_multi_seed.py"), "--algo", "ppol", "--seeds", "3",
                          "--epoch", "1", "--envs", "4", "--grouped"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "3 seeds x 1 epochs grouped" in out.stdout and out.stdout.count("epoch 1 seed") == 3


def _filled(T, seed, target_kl, lr=5e-4, Do=8, Da=2, H=64):
    from fsrl_amd.engine import Engine, EngineConfig
    e = Engine(EngineConfig(obs_dim=Do, act_dim=Da, hidden=H, env_num=2, max_grad_norm=0.5, target_kl=target_kl, lr=lr))
    r = np.random.default_rng(seed)
    e.set_params((0.1 * r.standard_normal(e.n_params)).astype(np.float32))
    obs = r.standard_normal((T + 1, 2, Do)).astype(np.float32)
    for t in range(T):
        e.push([0, 1], obs[t], 0.3 * r.standard_normal((2, Da)).astype(np.float32), r.normal(0.5, 0.5, 2),
               (r.random(2) < 0.1).astype(np.float64), [False, False], [t == T - 1] * 2, obs[t + 1])
    return e


def test_longest_member_stops_on_kl_first_and_an_empty_member_sits_out():
    """minibatch indices that no ACTIVE member has must not launch an empty grid.  Member 0 has the most
    minibatches and a learning rate that trips the KL stop in pass 0; member 1 (fewer minibatches) runs all passes; member 2
    holds no data at all but carries a longer stale plan from an earlier update."""
    from fsrl_amd.engine import EngineGroup
    lens = [700, 200, 500]
    engs = [_filled(lens[0], 1, 0.01, lr=2e-2), _filled(lens[1], 2, 0.01, lr=1e-5), _filled(lens[2], 3, 0.01, lr=1e-5)]
    solo = [_filled(lens[0], 1, 0.01, lr=2e-2), _filled(lens[1], 2, 0.01, lr=1e-5)]
    engs[2].ppo_update([0.3], 1 / 1.3, 64, 1, seed=3)                 # leaves a 15-minibatch plan behind ...
    engs[2].reset_store()                                               # ... and then no rows
    assert len(engs[2]) == 0
    grp = EngineGroup(engs)
    rng = np.random.default_rng(0)
    perms = [[rng.permutation(2 * T) for _ in range(4)] for T in lens[:2]] + [[np.zeros(0, np.int64) for _ in range(4)]]
    lags, resc = np.array([[0.2], [0.5], [0.9]]), [1 / 1.2, 1 / 1.5, 1 / 1.9]
    st, stop = grp.ppo_update(lags, resc, 128, 4, perms=perms)
    assert stop[0] == 0 and stop[1] == -1 and st[2].shape[0] == 0, (stop, [s.shape for s in st])
    assert st[0].shape[0] == 10 and st[1].shape[0] == 4 * 3            # one pass of 10 minibatches; four passes of 3
    for i, e in enumerate(solo):
        s1, sp = e.ppo_update(lags[i], resc[i], 128, 4, perms=perms[i])
        assert sp == stop[i] and s1.shape == st[i].shape
        np.testing.assert_allclose(st[i], s1, rtol=2e-4, atol=2e-4)
    grp.close()
    for e in engs + solo:
        e.close()


def test_destroying_member_zero_first_leaves_the_other_members_usable():
    """the shared stream is the group's own; when a member dies first the survivors get their own streams back
    and keep working, and destroying the broken group afterwards is harmless."""
    from fsrl_amd.engine import EngineGroup
    engs = [_filled(150, 5, None), _filled(150, 6, None)]
    ref = _filled(150, 6, None)
    grp = EngineGroup(engs)
    grp.ppo_update(np.array([[0.2], [0.4]]), [1 / 1.2, 1 / 1.4], 64, 1, seed=9)
    ref.ppo_update([0.4], 1 / 1.4, 64, 1, seed=9)                      # not the same shuffle stream: only shapes compared
    engs[0].close()                                                     # member 0 goes first
    s_a, _ = engs[1].ppo_update([0.4], 1 / 1.4, 64, 1, seed=11)          # the survivor's own calls still work
    assert np.isfinite(s_a).all() and np.isfinite(engs[1].get_params()).all()
    with pytest.raises(Exception):
        grp.ppo_update(np.array([[0.2], [0.4]]), [1 / 1.2, 1 / 1.4], 64, 1, seed=9)    # the group itself is over
    grp.close()
    s_b, _ = engs[1].ppo_update([0.4], 1 / 1.4, 64, 1, seed=12)
    assert np.isfinite(s_b).all()
    engs[1].close(); ref.close()


@pytest.mark.parametrize("H,k", [(128, 6), (256, 8)])
def test_large_groups_match_member_by_member(H, k):
    """More members than fill the chip in one round (k x 16 tiles x 3 networks > 256 workgroups).  The members' own updates
    use 4-row tiles at this minibatch size (v_mfma_f32_4x4x1: another K order), so the logged rows agree to fp32 rounding
    and the parameters to a few 1e-6 after four Adam steps."""
    from fsrl_amd.engine import EngineGroup
    T = 300                                                     # 600 rows per member: minibatches of 256 and 344 (merged)
    engs = [_filled(T, 20 + i, None, H=H) for i in range(k)]
    solo = [_filled(T, 20 + i, None, H=H) for i in range(k)]
    grp = EngineGroup(engs)
    rng = np.random.default_rng(1)
    perms = [[rng.permutation(2 * T) for _ in range(2)] for _ in range(k)]
    lags = np.linspace(0.1, 0.9, k).reshape(k, 1)
    resc = [1.0 / (1.0 + float(l)) for l in lags[:, 0]]
    st, stop = grp.ppo_update(lags, resc, 256, 2, perms=perms)
    for i, e in enumerate(solo):
        s1, _ = e.ppo_update(lags[i], resc[i], 256, 2, perms=perms[i])
        assert st[i].shape == s1.shape == (4, 11)
        np.testing.assert_allclose(st[i], s1, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(engs[i].get_params(), e.get_params(), rtol=0, atol=2e-5)
    grp.close()
    for e in engs + solo:
        e.close()


@pytest.mark.parametrize("H,k,Do", [(256, 8, 8), (128, 7, 40), (256, 9, 60)])
def test_tall_tile_plans_are_bit_identical(H, k, Do):
    """fsrl_group_set_plan: the forward / backward launch with 32-row tiles for the first n tiles of every (member, network) -- automatic
    (all of them once the 16-row tiles exceed the CU count), none (16-row tiles only), one (a mix), all -- gives the same logged rows and the same parameters to the bit: a
    row's arithmetic and the 16-row statistic slots do not depend on the tile height.  Members have different lengths (ragged last
    minibatches: 256 + merged 280 / 296 / ..., and one member with a single short minibatch whose tail tile is half empty)."""
    from fsrl_amd.engine import EngineGroup
    Ts = [268 + 8 * i for i in range(k - 1)] + [77]            # rows per member = 2 T: 536, 552, ... and 154 (< one minibatch of 256)
    rng = np.random.default_rng(3)
    perms = [[rng.permutation(2 * T) for _ in range(2)] for T in Ts]
    lags = np.linspace(0.1, 0.9, k).reshape(k, 1)
    resc = [1.0 / (1.0 + float(l)) for l in lags[:, 0]]
    got = {}
    for plan in (0, -1, 1, 64):
        engs = [_filled(T, 40 + i, None, H=H, Do=Do) for i, T in enumerate(Ts)]
        grp = EngineGroup(engs)
        grp.set_plan(plan)
        st, _ = grp.ppo_update(lags, resc, 256, 2, perms=perms)
        st2, _ = grp.ppo_update(lags, resc, 256, 2, perms=perms)
        got[plan] = (st, st2, [e.get_params() for e in engs])
        grp.close()
        for e in engs:
            e.close()
    base = got[0]
    assert all(np.isfinite(s).all() for s in base[0])
    for plan in (-1, 1, 64):
        for i in range(k):
            assert np.array_equal(got[plan][0][i], base[0][i]), (plan, i)
            assert np.array_equal(got[plan][1][i], base[1][i]), (plan, i)
            assert np.array_equal(got[plan][2][i], base[2][i]), (plan, i)


@pytest.mark.parametrize("H,Do,B", [(256, 8, 2048), (128, 40, 1536)])
def test_single_agent_tall_tile_plans_are_bit_identical(H, Do, B):
    """fsrl_ppo_set_plan: minibatches whose 16-row tiles exceed the CU count (2 048 rows x 3 networks = 384 tiles) run 32-row tiles by
    default; none / automatic / a mix of 3 tall tiles give the same logged rows and parameters to the bit (merged last minibatch ragged)."""
    T = 2300                                                    # 4 600 rows: minibatches of B and the merged rest
    rng = np.random.default_rng(4)
    perms = [rng.permutation(2 * T) for _ in range(2)]
    got = {}
    for plan in (0, -1, 3):
        e = _filled(T, 77, None, H=H, Do=Do)
        e.ppo_set_plan(plan)
        s1, _ = e.ppo_update([0.4], 1 / 1.4, B, 2, perms=perms)
        s2, _ = e.ppo_update([0.4], 1 / 1.4, B, 2, perms=perms)
        got[plan] = (s1, s2, e.get_params())
        e.close()
    assert np.isfinite(got[0][0]).all() and got[0][0].shape[0] == 2 * len(range(0, 2 * T - B + 1, B))
    for plan in (-1, 3):
        for a, b in zip(got[plan], got[0]):
            assert np.array_equal(a, b), plan
