#!/usr/bin/env python
"""In-kernel timeline of ppo_fwd_bwd_kernel (probe build only):

    bash fsrl_amd/csrc/build.sh --probes
    FSRL_HIP_LIB=$PWD/fsrl_amd/libfsrl_hip_probe.so FSRL_TSTAMP=1 python tools/tstamp_probe.py

Every workgroup stamps the shader clock (s_memtime) at 14 phase boundaries of the LAST fused-kernel launch of an update;
this prints, per phase, the median / max over workgroups of the time since the earliest workgroup's entry stamp, in
shader cycles (2.4 GHz nominal: 2400 cycles = 1 us)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import BATCH, ENVS, HID, NROWS, OBS, ACT, REPEAT, make_inputs, orthogonal_theta  # noqa: E402
from fsrl_amd.engine import Engine, EngineConfig  # noqa: E402

NAMES = ["entry", "loads issued", "small loads landed, LDS staged", "barrier", "layer 1 + barrier", "layer-2 MFMA loop done",
         "h2 in LDS + barrier", "head done (tile_forward end)", "A1/A2 stores + W2 column loads issued", "loss head done",
         "barrier", "dz2 + barrier", "D2/DO stores issued", "dz1 MFMA done", "D1 stored (end)"]
ORDER = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13]


def main():
    eng = Engine(EngineConfig(obs_dim=OBS, act_dim=ACT, hidden=HID, env_num=ENVS, buffer_size=100000, max_grad_norm=0.5,
                              target_kl=None))
    eng.set_params(orthogonal_theta(0, eng.n_params))
    obs, act, rew, cost, term, trunc = make_inputs(0)
    ids = np.arange(ENVS)
    for t in range(NROWS // ENVS):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    for k in range(3):
        eng.ppo_update(np.array([0.75]), 1 / 1.75, BATCH, REPEAT, seed=k + 1)
    fn = eng.lib.fsrl_probe_tstamps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
    ts = np.zeros((1024, 16), np.uint64)
    assert fn(eng._ctx, ts.ctypes.data_as(C.POINTER(C.c_uint64)), ts.size) == 0, "build with --probes and set FSRL_TSTAMP=1"
    used = ts[:, 0] > 0
    ts = ts[used].astype(np.int64)
    print("workgroups stamped:", len(ts))
    t0 = ts[:, 0].min()
    # stamp k of FSRL_TS: 0 entry,1 issued,2 staged+barrier,3 layer1+barrier,4 mfma done,5 h2+barrier,6 forward end,
    # 7 stores + wb issued, 8 loss head done, 9 barrier, 10 dz2 + barrier, 11 D2/DO stores, 12 dz1 mfma, 13 end
    labels = ["entry", "loads issued", "staged + barrier", "layer 1 + barrier", "layer-2 MFMA done", "h2 LDS + barrier",
              "head + barrier (forward end)", "A1/A2 stores, wb loads issued", "loss head done", "barrier", "dz2 + barrier",
              "D2/DO stores issued", "dz1 MFMA done", "D1 stored"]
    prev = None
    for k, lab in enumerate(labels):
        rel = ts[:, k] - t0
        own = ts[:, k] - ts[:, 0]
        step = "" if prev is None else f"  step median {np.median(ts[:, k] - ts[:, prev]):7.0f}"
        print(f"{k:2d} {lab:34s} since first entry: median {np.median(rel):7.0f} max {rel.max():7.0f} | since own entry median"
              f" {np.median(own):7.0f}{step}")
        prev = k
    print("entry skew over workgroups (cycles): max", (ts[:, 0] - t0).max(), " last exit", (ts[:, 13] - t0).max())
    eng.close()


if __name__ == "__main__":
    main()
