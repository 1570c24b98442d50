"""GAE / n-step scans of the oracle: a numpy restatement and ctypes bindings of the C one.

Reference: fsrl/policy/base_policy.py:524-540 (gae_return), :543-567 (nstep_return).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "fsrl_oracle.c")
_OUT = os.path.join(_HERE, "_build", "libfsrl_oracle.so")
_lib = None


def build_c(force=False):
    """gcc-compile oracle/csrc/fsrl_oracle.c -> oracle/_build/libfsrl_oracle.so."""
    if force or not os.path.exists(_OUT) or os.path.getmtime(_OUT) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", _OUT,
                               _SRC, "-lm"])
    return _OUT


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build_c())
        _lib.fsrl_oracle_pid_step.restype = ctypes.c_double
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def gae_return_np(v, v_next, rew, end_flag, gamma, gae_lambda):
    """Sequential float64 reverse scan (numba typing: f32 inputs are promoted to f64
    before the multiply by gamma)."""
    v = np.asarray(v).astype(np.float64)  # (float64 inputs are accepted: yardstick mode)
    v_next = np.asarray(v_next).astype(np.float64)
    rew = np.asarray(rew, np.float64)
    delta = rew + v_next * np.float64(gamma) - v
    disc = (1.0 - np.asarray(end_flag).astype(np.float64)) * (np.float64(gamma) * np.float64(gae_lambda))
    out = np.zeros(rew.shape)
    g = 0.0
    for i in range(len(rew) - 1, -1, -1):
        g = delta[i] + disc[i] * g
        out[i] = g
    return out


def gae_return_c(v, v_next, rew, end_flag, gamma, gae_lambda):
    lib = _load()
    v = np.ascontiguousarray(v, np.float32)
    v_next = np.ascontiguousarray(v_next, np.float32)
    rew = np.ascontiguousarray(rew, np.float64)
    end = np.ascontiguousarray(end_flag).astype(np.uint8)
    out = np.empty(len(rew), np.float64)
    lib.fsrl_oracle_gae_return(_p(v, ctypes.c_float), _p(v_next, ctypes.c_float),
                               _p(rew, ctypes.c_double), _p(end, ctypes.c_uint8),
                               ctypes.c_int64(len(rew)), ctypes.c_double(gamma),
                               ctypes.c_double(gae_lambda), _p(out, ctypes.c_double))
    return out


def nstep_return_np(metric, end_flag, target_q, indices, gamma, n_step):
    """target_q: [bsz, q] (value-masked); indices: [n_step, bsz]."""
    gamma = np.float64(gamma)
    gamma_buffer = np.ones(n_step + 1)
    for i in range(1, n_step + 1):
        gamma_buffer[i] = gamma_buffer[i - 1] * gamma
    shape = target_q.shape
    bsz = shape[0]
    tq = np.asarray(target_q).astype(np.float64).reshape(bsz, -1)
    ret = np.zeros(tq.shape)
    gammas = np.full(bsz, n_step)
    for n in range(n_step - 1, -1, -1):
        now = indices[n]
        hit = np.asarray(end_flag)[now] > 0
        gammas[hit] = n + 1
        ret[hit] = 0.0
        ret = np.asarray(metric, np.float64)[now].reshape(bsz, 1) + gamma * ret
    return (tq * gamma_buffer[gammas].reshape(bsz, 1) + ret).reshape(shape)


def nstep_return_c(metric, end_flag, target_q, indices, gamma, n_step):
    lib = _load()
    shape = target_q.shape
    bsz = shape[0]
    tq = np.ascontiguousarray(np.asarray(target_q).astype(np.float64).reshape(bsz, -1))
    metric = np.ascontiguousarray(metric, np.float64)
    end = np.ascontiguousarray(end_flag).astype(np.uint8)
    idx = np.ascontiguousarray(indices, np.int64)
    out = np.empty_like(tq)
    lib.fsrl_oracle_nstep_return(_p(metric, ctypes.c_double), _p(end, ctypes.c_uint8),
                                 _p(tq, ctypes.c_double), _p(idx, ctypes.c_int64),
                                 ctypes.c_int64(bsz), ctypes.c_int64(tq.shape[1]),
                                 ctypes.c_double(gamma), ctypes.c_int64(n_step),
                                 _p(out, ctypes.c_double))
    return out.reshape(shape)


def pid_step_c(state, pid, value, threshold):
    """state: float64[3] = (error_old, error_integral, lagrangian), updated in place."""
    lib = _load()
    pid = np.ascontiguousarray(pid, np.float64)
    return lib.fsrl_oracle_pid_step(_p(state, ctypes.c_double), _p(pid, ctypes.c_double),
                                    ctypes.c_double(value), ctypes.c_double(threshold))
