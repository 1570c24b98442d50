"""The boundary is a C ABI: a plain-C host (tests/c_host/abi_host.c, gcc, no Python / torch in the process) links
libfsrl_hip.so through include/fsrl_hip.h, fills the store, runs one PPO-Lagrangian update and prints checksums; the same
call sequence from Python (ctypes) must give the same bytes."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_host", "abi_host.c")


def _build(tmp_path):
    exe = str(tmp_path / "abi_host")
    lib_dir = os.path.join(ROOT, "fsrl_amd")
    cmd = ["gcc", "-O2", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-L" + lib_dir, "-lfsrl_hip",
           "-Wl,-rpath," + lib_dir, "-lm", "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def test_header_is_plain_c_and_the_library_links_from_c(tmp_path):
    """no GPU needed: include/fsrl_hip.h compiles as C99 (-Wall -Werror) and every symbol the host uses resolves"""
    assert os.path.exists(_build(tmp_path))


def _fnv1a(b: bytes) -> int:
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.gpu
def test_c_host_and_python_host_produce_the_same_bytes(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("params ")][-1].split()
    got = dict(zip(line[0::2], line[1::2]))
    # ---- the same sequence through ctypes
    from fsrl_amd.engine import Engine, EngineConfig
    M = (1 << 64) - 1
    state = [88172645463325252]

    def u():
        s = state[0]
        s ^= s >> 12; s ^= (s << 25) & M; s ^= s >> 27
        state[0] = s
        return ((s * 2685821657736338717 & M) >> 11) / 9007199254740992.0
    E, T, Do, Da, H, B, R = 4, 150, 8, 2, 64, 64, 3
    eng = Engine(EngineConfig(obs_dim=Do, act_dim=Da, hidden=H, n_critics=2, env_num=E, buffer_size=4000, max_grad_norm=0.5,
                              target_kl=None))
    assert eng.n_params == int(got["params"])
    eng.set_params(np.array([np.float32((u() - 0.5) * 0.2) for _ in range(eng.n_params)], np.float32))
    obs = np.array([np.float32(u() * 2.0 - 1.0) for _ in range(E * Do)], np.float32).reshape(E, Do)
    ids = np.arange(E)
    for t in range(T):
        nxt = np.array([np.float32(u() * 2.0 - 1.0) for _ in range(E * Do)], np.float32).reshape(E, Do)
        act = np.array([np.float32(u() * 0.6 - 0.3) for _ in range(E * Da)], np.float32).reshape(E, Da)
        rew, cost, term, trunc = np.zeros(E), np.zeros(E), np.zeros(E, bool), np.zeros(E, bool)
        for e in range(E):
            rew[e] = u()
            cost[e] = 1.0 if u() < 0.1 else 0.0
            term[e] = (e == 3 and t == 120)
            trunc[e] = ((t + 1) % 50 == 0 and not term[e])
        eng.push(ids, obs, act, rew, cost, term, trunc, nxt)
        obs = nxt
    stats, stopped = eng.ppo_update(np.array([0.75]), 1.0 / 1.75, B, R, perms=None, seed=12345)
    theta = eng.get_params()
    assert stats.shape[0] == int(got["steps"]) and stopped == int(got["stopped"]) == -1
    assert "%016x" % _fnv1a(theta.astype(np.float32).tobytes()) == got["theta"]
    assert "%016x" % _fnv1a(np.ascontiguousarray(stats, np.float32).tobytes()) == got["stats"]
    assert (int(got["rank"]), int(got["world"])) == (0, 1)
    eng.close()
