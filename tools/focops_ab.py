#!/usr/bin/env python
"""A/B of two library builds on the FOCOPS golden cases: parameters after the update, bit for bit.
    FSRL_HIP_LIB=<other build> python tools/focops_ab.py dump a.npz ; python tools/focops_ab.py dump b.npz ; ... cmp"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import json  # noqa: E402
from helpers import load_npz  # noqa: E402
from test_gpu_focops import _engine  # noqa: E402


def _run_case(name):
    g = load_npz(f"focops_{name}.npz")
    cfg = json.loads(str(g["cfg_json"]))
    eng = _engine(cfg, g)
    nu = float(g["stats_nu"][0][1]); nu_loss = float(g["stats_nu"][0][0])
    perms = list(g["perms"]) + [np.arange(len(g["indices"]))] * (cfg["repeat"] - len(g["perms"]))
    stats, stopped = eng.focops_update(nu, nu_loss, cfg["batch_size"], cfg["repeat"], perms=perms)
    theta = eng.get_params()
    d = np.abs(theta - g["theta_final"])
    print(name, "vs golden: q999", np.quantile(d, 0.999), "max", d.max(), "mean", d.mean())
    eng.close()
    return theta, stats


if sys.argv[1] == "dump":
    out = {}
    for name in ("small", "c1", "earlystop"):
        theta, stats = _run_case(name)
        out[name + "_theta"] = theta; out[name + "_stats"] = stats
    np.savez(sys.argv[2], **out)
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print(k, "equal" if np.array_equal(a[k], b[k]) else f"max diff {d.max():.3e} mean {d.mean():.3e}")
        if k.endswith("theta"):
            n = len(d); na = n - 2 * ((n - 0) // 3)      # rough thirds: actor | critic | critic
            for nm, sl in (("first third", slice(0, n // 3)), ("second", slice(n // 3, 2 * n // 3)), ("third", slice(2 * n // 3, n))):
                print("   ", nm, "differing entries", int((d[sl] > 0).sum()), "of", d[sl].size)
