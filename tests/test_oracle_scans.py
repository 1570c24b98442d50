"""Oracle (numpy + C) vs golden vectors of the reference's gae_return / nstep_return / PID.
Bit-exact: these are float64 recurrences with a fixed evaluation order."""
import numpy as np
import pytest

from helpers import load_npz
from oracle import scans
from oracle.pid import PIDLagrangian


@pytest.mark.parametrize("n", [1, 7, 300, 2048, 20000])
@pytest.mark.parametrize("gl", [(0.99, 0.95), (1.0, 1.0), (0.9, 0.0)])
def test_gae_bit_exact(n, gl):
    g = load_npz("gae_cases.npz")
    gamma, lam = gl
    want = g[f"n{n}_g{gamma}_l{lam}_adv"]
    args = (g[f"n{n}_v"], g[f"n{n}_vn"], g[f"n{n}_rew"], g[f"n{n}_end"], gamma, lam)
    got_c = scans.gae_return_c(*args)
    assert got_c.dtype == np.float64 and np.array_equal(got_c, want)
    if n <= 2048:
        assert np.array_equal(scans.gae_return_np(*args), want)


def test_gae_empty():
    z32, z64 = np.zeros(0, np.float32), np.zeros(0)
    assert scans.gae_return_c(z32, z32, z64, np.zeros(0, bool), 0.99, 0.95).shape == (0, )


@pytest.mark.parametrize("n_step", [1, 2, 3, 5])
@pytest.mark.parametrize("gamma", [0.99, 0.9])
def test_nstep_bit_exact(n_step, gamma):
    g = load_npz("nstep_cases.npz")
    want = g[f"n{n_step}_g{gamma}_ret"]
    args = (g["metric"], g["end_flag"], g[f"n{n_step}_target_q"], g[f"n{n_step}_indices"], gamma,
            n_step)
    assert np.array_equal(scans.nstep_return_np(*args), want)
    assert np.array_equal(scans.nstep_return_c(*args), want)


@pytest.mark.parametrize("name", ["default", "sgd"])
def test_pid_trace_bit_exact(name):
    g = load_npz("pid_trace.npz")
    pid, limit = g[name + "_pid"], float(g[name + "_limit"])
    opt = PIDLagrangian(pid)
    st = np.zeros(3)
    for i, c in enumerate(g["costs"]):
        lag = opt.step(c, limit)
        lag_c = scans.pid_step_c(st, pid, float(c), limit)
        assert lag == g[name + "_lag"][i] == lag_c
        assert opt.error_integral == g[name + "_integral"][i] == st[1]
        assert opt.error_old == g[name + "_error_old"][i] == st[0]
