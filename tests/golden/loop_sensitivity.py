#!/usr/bin/env python
"""Diagnosis (build container only): how far does the UNMODIFIED reference move from its own closed-loop fixtures when only the
summation order of its BLAS calls changes?  Re-runs the FOCOPS / TRPO-Lag / CPO loops of gen_golden_loop.py with another torch
thread count (the fixtures were recorded with 4) into a scratch directory and prints the per-cycle |reward| / |cost| differences.
Measured (torch 2.10 CPU, 1 thread vs 4):

    focops  reward 0 in every cycle, theta max diff 2.3e-7           (128-row minibatches: same GEMM path)
    trpo    reward 0, 0.20, 0.46, 0.50, 0.74, 1.63, 2.29, 2.81 ; cost up to 1.0 ; theta max diff 0.040
    cpo     reward 0, 0.74, 2.10, 3.06, 8.82, 26.1, 1.10, 17.0 ; cost up to 4.6 ; theta max diff 0.126
    ppo / sac / ddpg   max |reward diff| 7e-5 / 7e-5 / 1.5e-5, costs identical        (`all` on the command line)
    cvpo               max |reward diff| 0.018, costs identical

These are the bands tests/test_gpu_loop.py::test_closed_trust_region_loop... works with: the trust-region loops amplify a rounding
difference (fp32 conjugate gradients, line-search accept / reject) to these sizes within eight cycles in the reference itself.
    python tests/golden/loop_sensitivity.py [threads]"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_loop as G  # noqa: E402

if __name__ == "__main__":
    nt = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
    names = ["focops", "trpo", "cpo"]
    G.HERE = tempfile.mkdtemp()
    torch.set_num_threads(nt)
    G.gen_focops("focops", 8, 2, (64, 64), env_num=8, ep_len=60, cycles=10, batch_size=128, repeat=4, seed=72, cost_limit=8.0)
    G.gen_trust("trpo", "trpo", 8, 2, (64, 64), env_num=8, ep_len=60, cycles=8, repeat=2, seed=75, cost_limit=20.0)
    G.gen_trust("cpo", "cpo", 8, 2, (64, 64), env_num=8, ep_len=60, cycles=8, repeat=2, seed=74, cost_limit=20.0)
    if "all" in sys.argv[1:]:
        G.gen("ppo", 8, 2, (64, 64), env_num=8, ep_len=60, cycles=12, batch_size=128, repeat=4, seed=70, cost_limit=8.0,
              target_kl=0.5, max_grad_norm=0.5)
        G.gen_sac("sac", 8, 2, (64, 64), env_num=6, ep_len=50, cycles=10, batch_size=64, updates_per_cycle=30, seed=71, cost_limit=5.0)
        G.gen_ddpg("ddpg", 8, 2, (64, 64), env_num=6, ep_len=50, cycles=10, batch_size=64, updates_per_cycle=30, seed=73, cost_limit=5.0)
        G.gen_cvpo("cvpo", 8, 2, (64, 64), env_num=6, ep_len=50, cycles=10, batch_size=64, updates_per_cycle=30, seed=76, cost_limit=2.0)
        names += ["ppo", "sac", "ddpg", "cvpo"]
    for name in names:
        a = np.load(os.path.join(G.HERE, f"loop_{name}.npz")); b = np.load(os.path.join(HERE, f"loop_{name}.npz"))
        d = np.abs(a["curve"][:, :2] - b["curve"][:, :2])
        th = float(np.abs(a["theta_final"] - b["theta_final"]).max()) if "theta_final" in a.files else float("nan")
        print(name, "threads", nt, "|reward diff| per cycle", np.round(d[:, 0], 3).tolist(), "|cost diff|", np.round(d[:, 1], 3).tolist(),
              "theta max diff", th)
