#!/usr/bin/env python
"""One-shot GPU diagnostic: every stage of the HIP PPO-Lag path vs the oracle on the golden
cases, printing max errors (no asserts) so a single gpurun call tells what is wrong."""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_npz, oracle_cfg_and_data, ppo_case, end_flag_of  # noqa: E402
from fsrl_amd.engine import Engine, EngineConfig  # noqa: E402
from oracle.pid import rescaling_factor  # noqa: E402
from oracle.ppo_lag import PPOLagOracle, split_chunks  # noqa: E402


def engine_for(cfg, g, **over):
    ec = EngineConfig(obs_dim=cfg["obs_dim"], act_dim=cfg["act_dim"], hidden=cfg["hidden"][0],
                      n_critics=2, env_num=cfg["env_num"], buffer_size=100000,
                      max_action=cfg["max_action"], gamma=cfg["gamma"], gae_lambda=cfg["gae_lambda"],
                      eps_clip=cfg["eps_clip"], dual_clip=cfg["dual_clip"], vf_coef=cfg["vf_coef"],
                      max_grad_norm=cfg["max_grad_norm"], target_kl=cfg["target_kl"],
                      norm_adv=cfg["advantage_normalization"], use_lagrangian=cfg["use_lagrangian"],
                      lr=cfg["lr"])
    for k, v in over.items():
        setattr(ec, k, v)
    return Engine(ec)


def push_golden(eng, g):
    """Replay the golden buffer into the device store in lock-step like the collector."""
    rows = g["env_rows"]
    off = np.concatenate([[0], np.cumsum(rows)])
    T = rows.max()
    for t in range(T):
        ids = [e for e in range(len(rows)) if t < rows[e]]
        sel = np.array([off[e] + t for e in ids])
        eng.push(ids, g["buf_obs"][sel], g["buf_act"][sel], g["buf_rew"][sel], g["buf_cost"][sel],
                 g["buf_terminated"][sel], g["buf_truncated"][sel], g["buf_obs_next"][sel])


def err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max()) if a.size else 0.0


def main():
    print("torch", torch.__version__, "cuda", torch.cuda.is_available())
    # ---- GAE standalone
    g = load_npz("gae_cases.npz")
    cfg, gg = ppo_case("tiny")
    eng = engine_for(cfg, gg)
    for n in (1, 7, 300, 2048, 20000):
        for gamma, lam in ((0.99, 0.95), (1.0, 1.0), (0.9, 0.0)):
            got = eng.gae_return(g[f"n{n}_v"], g[f"n{n}_vn"], g[f"n{n}_rew"], g[f"n{n}_end"], gamma, lam)
            want = g[f"n{n}_g{gamma}_l{lam}_adv"]
            print(f"gae n={n} g={gamma} l={lam}: bit_exact={np.array_equal(got, want)} maxerr={err(got, want):.3e}")
    eng.close()

    for name in ("tiny", "dualclip", "earlystop", "c1", "c2"):
        try:
            cfg, g = ppo_case(name)
            print(f"==== case {name}: N={len(g['indices'])} H={cfg['hidden']} B={cfg['batch_size']}")
            eng = engine_for(cfg, g)
            eng.set_params(g["theta0"])
            print(" params roundtrip", err(eng.get_params(), g["theta0"]))
            push_golden(eng, g)
            idx = eng.sample0()
            print(" sample0 order equal:", np.array_equal(idx, g["indices"]), "len", len(eng))
            lag = g["lagrangian"]
            n = eng.ppo_begin(lag, rescaling_factor(lag), cfg["batch_size"])
            for k in ("values", "rets", "advs", "logp_old"):
                print(f" process {k}: maxerr {err(eng.batch_get(k), g[k]):.3e}  (ref scale {np.abs(g[k]).max():.3e})")
            # --- gradient of the first minibatch (lr=0 engine so params stay put)
            eng.ppo_end()
            eng0 = engine_for(cfg, g, lr=0.0, target_kl=None)
            eng0.set_params(g["theta0"])
            push_golden(eng0, g)
            eng0.ppo_begin(lag, rescaling_factor(lag), cfg["batch_size"])
            eng0.ppo_pass(g["perms"][0])
            st0 = eng0.ppo_end_stats(1000)
            grads = eng0.get_grads()
            ocfg, data = oracle_cfg_and_data(cfg, g)
            o = PPOLagOracle(ocfg)
            o.set_params(g["theta0"])
            pb = o.process(data)
            chunks = split_chunks(len(data), cfg["batch_size"], g["perms"][0])
            loss, dist, st = o._minibatch_losses(pb, chunks[-1], lag, rescaling_factor(lag))
            o.optim.zero_grad(); loss.backward()
            og = torch.cat([t.grad.reshape(-1) for t in o._leaves]).numpy()
            print(f" last-minibatch grad: maxerr {err(grads, og):.3e} (scale {np.abs(og).max():.3e}) "
                  f"norm hip {np.linalg.norm(grads):.6f} oracle {np.linalg.norm(og):.6f}")
            off = 0
            for si, spec in enumerate(o.specs):
                for pname, shape in spec.items():
                    sz = int(np.prod(shape))
                    e = err(grads[off:off + sz], og[off:off + sz])
                    print(f"   net{si}.{pname:12s} err {e:.3e} scale {np.abs(og[off:off+sz]).max():.3e}")
                    off += sz
            print(" stats row0 hip   ", np.array2string(st0[0], precision=5))
            loss0, _, s0 = o._minibatch_losses(pb, chunks[0], lag, rescaling_factor(lag))
            print(" stats row0 oracle", {k: round(v, 5) for k, v in s0.items()})
            eng0.close()
            # --- full update
            eng.optim_reset()
            t0 = time.time()
            stats, stopped = eng.ppo_update(lag, rescaling_factor(lag), cfg["batch_size"], cfg["repeat"],
                                            perms=g["perms"])
            dt = time.time() - t0
            print(f" full update: steps {len(stats)} (golden {len(g['stats'])}) stopped_pass {stopped} "
                  f"(golden early={int(g['early_stop_msgs'])}) wall {dt*1e3:.2f} ms")
            m = min(len(stats), len(g["stats"]))
            print(f" stats maxerr {err(stats[:m], g['stats'][:m]):.3e}")
            for j, key in enumerate(g["stats_keys"]):
                print(f"   {str(key):20s} err {err(stats[:m, j], g['stats'][:m, j]):.3e} scale {np.abs(g['stats'][:, j]).max():.3e}")
            print(f" theta_final maxerr {err(eng.get_params(), g['theta_final']):.3e}")
            print(" timing", eng.last_timing())
            eng.close()
        except Exception:
            traceback.print_exc()


if __name__ == "__main__":
    main()
