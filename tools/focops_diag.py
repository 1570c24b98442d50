import json, sys
import numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from helpers import load_npz
from test_gpu_focops import _engine
for name in ("small", "earlystop"):
    g = load_npz(f"focops_{name}.npz"); cfg = json.loads(str(g["cfg_json"]))
    eng = _engine(cfg, g)
    nu = float(g["stats_nu"][0][1]); nu_loss = float(g["stats_nu"][0][0])
    perms = list(g["perms"]) + [np.arange(len(g["indices"]))] * (cfg["repeat"] - len(g["perms"]))
    stats, stopped = eng.focops_update(nu, nu_loss, cfg["batch_size"], cfg["repeat"], perms=perms)
    want = np.concatenate([g["stats_nu"], g["stats_actor"], g["stats_critic"]], 1)
    np.set_printoptions(linewidth=200, precision=3, suppress=False)
    print(name, "rel diff per step/col:")
    print(np.abs(stats - want) / (np.abs(want) + 1e-9))
    print("entropy got ", stats[:6, 4]); print("entropy want", want[:6, 4])
    print("theta diff", np.abs(eng.get_params() - g["theta_final"]).max())
