#!/usr/bin/env python
"""In-kernel timeline of fb_hvp_mixed_kernel<256, true> at configs[2] size (probe build only):

    bash fsrl_amd/csrc/build.sh --probes
    FSRL_HIP_LIB=$PWD/fsrl_amd/libfsrl_hip_probe.so FSRL_TSTAMP=1 python tools/tstamp_hvp.py

Every workgroup of the LAST Hessian-vector product of a CPO repeat stamps the shader clock at its phase boundaries; medians
over the 32-row and over the 16-row workgroups, cycles since the workgroup's own entry (2400 cycles = 1 us nominal)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_trust import inputs  # noqa: E402
from fsrl_amd.engine import Engine, EngineConfig  # noqa: E402

LABELS = ["entry", "loads issued (fragments, cached h1 / h2, x)", "barrier", "layer-1 tangent + barrier", "R{z2}: W2 R{h1} GEMM",
          "V2 fragment ingest + V2 h1 GEMM", "R{h2} written + barrier", "R{h1} spill, head pre-activations + barrier",
          "KL head, R{h2} spill + barrier", "dz2 / R{dz2} + barrier", "R{dz2} W2 GEMM (incl. its fragment ingest)",
          "dz2 V2 GEMM (incl. its fragment ingest)", "R{dz1} store, R{dz2} / R{dout} spill (end)"]


def main():
    envs, T = 20, 1000
    obs, act, rew, cost, term, trunc = inputs(np.random.default_rng(0), envs, T, 60, 2, 1000)
    eng = Engine(EngineConfig(obs_dim=60, act_dim=2, hidden=256, env_num=envs, target_kl=None, lr=1e-3))
    eng.set_params((0.05 * np.random.default_rng(1).standard_normal(eng.n_params)).astype(np.float32))
    ids = np.arange(envs)
    for t in range(T):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    eng.tr_begin(target_kl=0.01, l2_reg=0.001, critic_lr=1e-3, max_backtracks=10, optim_critic_iters=2, cost_limit=10.0)
    eng.cpo_learn(25.0, 1)
    fn = eng.lib.fsrl_probe_tstamps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
    ts = np.zeros((1024, 16), np.uint64)
    assert fn(eng._ctx, ts.ctypes.data_as(C.POINTER(C.c_uint64)), ts.size) == 0, "build with --probes and set FSRL_TSTAMP=1"
    ts = ts.astype(np.int64)
    used = np.flatnonzero(ts[:, 0] > 0)
    print("workgroups stamped:", len(used), "(512 x 32 rows, then 226 x 16 rows at N = 20 000)")
    t0 = ts[used, 0].min()
    for name, sel in (("32-row tiles", used[used < 512]), ("16-row tiles", used[used >= 512])):
        if not len(sel):
            continue
        x = ts[sel]
        print(f"--- {name}: {len(sel)} workgroups; entry since first: median {np.median(x[:, 0] - t0):.0f} max {(x[:, 0] - t0).max():.0f}; "
              f"total median {np.median(x[:, 12] - x[:, 0]):.0f} cycles")
        for k in range(1, 13):
            print(f"{k:2d} {LABELS[k]:58s} +{np.median(x[:, k] - x[:, k - 1]):7.0f}   (since entry {np.median(x[:, k] - x[:, 0]):7.0f})")
    # the launch's schedule.  The shader clock is per XCD (eight unrelated bases): workgroups are clustered by base, and each
    # cluster (= one XCD, 32 CUs) gets its own timeline: when its workgroups enter and leave relative to the cluster's first entry
    order = np.argsort(ts[used, 0])
    u = used[order]
    cuts = np.flatnonzero(np.diff(ts[u, 0]) > 10**9)
    clusters = np.split(u, cuts + 1)
    print(f"--- schedule: {len(clusters)} clock domains (XCDs)")
    for ci, cl in enumerate(clusters):
        base = ts[cl, 0].min()
        ent, end = ts[cl, 0] - base, ts[cl, 12] - base
        big = cl < 512
        line = f"XCD {ci}: {len(cl)} workgroups ({int(big.sum())} x 32 rows), span {end.max()} cycles = {end.max() / 2400:.1f} us; busy sum / 32 CUs = {(end - ent).sum() / 32:.0f}"
        print(line)
        if ci == 0:
            o = np.argsort(ent)
            print("   entry -> exit (rows) of every 4th workgroup in entry order:")
            print("   " + "  ".join(f"{int(ent[k])}->{int(end[k])}({32 if big[k] else 16})" for k in o[::4]))
    eng.close()


if __name__ == "__main__":
    main()
