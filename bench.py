#!/usr/bin/env python
"""bench.py -- policy-updates/s of the PPO-Lagrangian update path on MI355X.

A "step" = ONE full `policy.update()` (BasePolicy.update, fsrl/policy/base_policy.py:332-355):
buffer.sample(0) -> process_fn (V(obs), V(obs_next), float64 GAE x2, logp_old) -> learn
(repeat=4 passes x 78 minibatches of 256 = 312 fwd/bwd/clip/Adam steps) over a HBM-resident
20 000-row on-policy buffer, SafetyCarCircle-v0 shape (obs 8, act 2), 256x256 MLPs
(BASELINE.json configs[1]).  Synthetic data, random-init (orthogonal)
weights.  KL early stop is disabled for throughput (target_kl = inf), like BASELINE.md says.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one independent agent (seed = rank) per GPU, no data-path collective ("weak" scaling);
RCCL is used only for the barrier and the max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

OBS, ACT, HID, NROWS, ENVS, EPLEN, BATCH, REPEAT = 8, 2, 256, 20000, 20, 250, 256, 4
F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: fp32 matrix = vector peak


def csrc_sha16():
    """hash of the kernel / host sources the library was built from: the committed PMC files carry the hash they were captured
    at (tools/collect_profiles.py), so a traffic figure read from a stale capture says so in the bench line"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fsrl_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".inc")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def make_inputs(seed):
    """SURVEY 8(d) synthetic store: 20 envs x 1000 rows, end_flag every 250 (N = 20 000)."""
    rng = np.random.default_rng(seed)
    T = NROWS // ENVS
    obs = rng.standard_normal((T + 1, ENVS, OBS)).astype(np.float32)
    act = (0.3 * rng.standard_normal((T, ENVS, ACT))).astype(np.float32)
    rew = rng.normal(0.5, 0.5, (T, ENVS))
    cost = (rng.random((T, ENVS)) < 0.1).astype(np.float64)
    trunc = np.zeros((T, ENVS), bool)
    trunc[EPLEN - 1::EPLEN] = True
    term = np.zeros((T, ENVS), bool)
    return obs, act, rew, cost, term, trunc


def orthogonal_theta(seed, n_params_check=None, hid=None):
    """PPOLagAgent init (ppo_lag_agent.py:147-153): orthogonal W, zero b, sigma_param = -0.5."""
    import torch
    HID = hid or globals()["HID"]
    torch.manual_seed(seed)
    parts = []

    def lin(o, i):
        w = torch.empty(o, i)
        torch.nn.init.orthogonal_(w)
        return [w.reshape(-1), torch.zeros(o)]

    parts.append(torch.full((ACT, ), -0.5))
    parts += lin(HID, OBS) + lin(HID, HID) + lin(ACT, HID)
    for _ in range(2):
        parts += lin(HID, OBS) + lin(HID, HID) + lin(1, HID)
    theta = torch.cat(parts).numpy()
    assert n_params_check is None or theta.size == n_params_check
    return theta


def flops_fwdbwd_launch(rows):
    """Algorithmic FLOPs of ONE ppo_fwd_bwd_kernel launch (DESIGN.md 'Roofline'):
    forward of the 3 nets (2*(Do*H + H*H + H*out) per row and net) + activation backward
    (dz2 = dout W3, dz1 = dz2 W2: 2*(H*out + H*H)); weight gradients are the other kernel."""
    per_row = 0
    for out in (ACT, 1, 1):
        per_row += 2 * (OBS * HID + HID * HID + HID * out) + 2 * (HID * out + HID * HID)
    return per_row * rows


def flops_wgrad_launch(rows):
    """Algorithmic FLOPs of ONE ppo_wgrad_kernel launch: dW_l = (upstream)^T (input) for the three layers of the 3 nets."""
    per_row = 0
    for out in (ACT, 1, 1):
        per_row += 2 * (OBS * HID + HID * HID + HID * out)
    return per_row * rows


def _usable_cpus():
    from fsrl_amd.parallel import usable_cpus
    return usable_cpus()


def cpu_baseline(theta, inputs, seconds=12.0, threads=4, hid=None):
    """The oracle (torch CPU fp32 port of the reference update) timed on the host cores,
    on a bounded sample: whole updates of the SAME workload until `seconds` elapsed.
    threads = 4: the reference default `thread=4` (fsrl/config/ppol_cfg.py:11)."""
    import torch
    from oracle.ppo_lag import OnPolicyData, PPOLagConfig, PPOLagOracle
    HID = hid or globals()["HID"]
    torch.set_num_threads(threads)
    obs, act, rew, cost, term, trunc = inputs
    em = lambda a: np.concatenate([a[:, e] for e in range(ENVS)])
    data = OnPolicyData(obs=em(obs[:-1]), act=em(act), rew=em(rew), cost=em(cost), terminated=em(term),
                        truncated=em(trunc), obs_next=em(obs[1:]), end_flag=em(term | trunc))
    o = PPOLagOracle(PPOLagConfig(obs_dim=OBS, act_dim=ACT, hidden=(HID, HID), max_grad_norm=0.5,
                                  target_kl=1e9))
    o.set_params(theta)
    rng = np.random.default_rng(0)
    lag = np.array([0.75])
    n, t0 = 0, time.perf_counter()
    while True:
        perms = [rng.permutation(NROWS) for _ in range(REPEAT)]
        o.update(data, lag, 1.0 / 1.75, BATCH, REPEAT, perms=perms)
        n += 1
        if time.perf_counter() - t0 > seconds or n >= 12:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "updates/s", "cores": threads, "kind": "port",
            "sample": f"{n} full updates (312 grad steps each) of the same 20k-row workload, {HID}x{HID} MLPs, "
                      f"torch CPU fp32, {threads} threads; host: {os.cpu_count()} cpus visible, "
                      f"{_usable_cpus():g} usable (affinity / cgroup quota)"}


def torch_rocm_baseline(inputs, updates=2, seconds=20.0):
    """Context, never a target and never on the product path: the reference's ONLY GPU path is stock PyTorch with
    `device="cuda"` (fsrl/agent/ppo_lag_agent.py:136-145 puts the Tianshou nets on the device; ppo_lag.py:214-257 then runs
    312 autograd steps with ~12 `.item()` read-backs each).  This leg runs that update -- same workload, same structure: per-
    critic value passes + a host GAE scan (base_policy.py:427-446), then repeat x Batch.split minibatches of forward, losses,
    backward, clip_grad_norm_, one Adam over all nets, the logged scalars read back per step -- with torch.nn / torch.optim
    on THIS GPU through ROCm PyTorch, so the hand-written path is shown against the same chip, not only against 4 CPU threads.
    Written here from the reference's call structure with stock modules (it is a comparator, not the oracle, and checks
    nothing)."""
    import torch
    from scipy.signal import lfilter
    from torch import nn
    from torch.distributions import Independent, Normal
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(0)

    def mlp(out):
        m = nn.Sequential(nn.Linear(OBS, HID), nn.ReLU(), nn.Linear(HID, HID), nn.ReLU(), nn.Linear(HID, out))
        for l in m:
            if isinstance(l, nn.Linear):
                nn.init.orthogonal_(l.weight); nn.init.zeros_(l.bias)
        return m.to(dev)
    actor, critics = mlp(ACT), [mlp(1), mlp(1)]
    sigma = nn.Parameter(torch.full((ACT, 1), -0.5, device=dev))
    params = [sigma] + list(actor.parameters()) + [p for c in critics for p in c.parameters()]
    optim = torch.optim.Adam(params, lr=5e-4)
    obs, act, rew, cost, term, trunc = inputs
    em = lambda a: np.concatenate([a[:, e] for e in range(ENVS)])
    h_obs, h_next, h_act = em(obs[:-1]), em(obs[1:]), em(act)
    metrics = [em(rew).astype(np.float64), em(cost).astype(np.float64)]
    end = em(term | trunc); mask = ~em(term)
    gamma, lam_gae, eps_clip, lag, resc = 0.99, 0.95, 0.2, 0.75, 1.0 / 1.75
    bounds = np.flatnonzero(end) + 1
    starts = np.concatenate([[0], bounds[:-1]]) if len(bounds) else np.array([0])
    if len(bounds) == 0 or bounds[-1] != NROWS:
        starts, bounds = np.append(starts, bounds[-1] if len(bounds) else 0), np.append(bounds, NROWS)

    def dist_of(o):
        mu = torch.tanh(actor(o))
        return Independent(Normal(mu, (sigma.view(1, -1) + torch.zeros_like(mu)).exp()), 1)

    def one_update(rng):
        # process_fn: values on the device, the scan on the host (the reference's numba gae_return), back to the device
        t_obs = torch.as_tensor(h_obs, device=dev); t_next = torch.as_tensor(h_next, device=dev); t_act = torch.as_tensor(h_act, device=dev)
        advs, rets = [], []
        with torch.no_grad():
            for i, c in enumerate(critics):
                v = c(t_obs).flatten().cpu().numpy(); vn = c(t_next).flatten().cpu().numpy() * mask
                delta = metrics[i] + gamma * vn - v
                adv = np.empty(NROWS)
                for a, b in zip(starts, bounds):          # discounted reverse cumulative sum per episode segment
                    adv[a:b] = lfilter([1.0], [1.0, -gamma * lam_gae], delta[a:b][::-1])[::-1]
                advs.append(torch.as_tensor(adv, dtype=torch.float32, device=dev))
                rets.append(torch.as_tensor(adv + v, dtype=torch.float32, device=dev))
            logp_old = dist_of(t_obs).log_prob(t_act)
        advs, rets = torch.stack(advs, -1), torch.stack(rets, -1)
        steps = 0
        for _ in range(REPEAT):
            perm = rng.permutation(NROWS)
            chunks = [perm[i:i + BATCH] for i in range(0, NROWS, BATCH)]
            if len(chunks) > 1 and len(chunks[-1]) < BATCH:        # merge_last
                chunks[-2] = np.concatenate([chunks[-2], chunks[-1]]); chunks.pop()
            for idx in chunks:
                ix = torch.as_tensor(idx, device=dev)
                o, a, adv, ret, lo = t_obs[ix], t_act[ix], advs[ix].clone(), rets[ix], logp_old[ix]
                d = dist_of(o)
                logp = d.log_prob(a)
                ratio = (logp - lo).exp().reshape(1, -1)
                for i in range(2):
                    adv[..., i] = (adv[..., i] - adv[..., i].mean()) / adv[..., i].std()
                s1, s2 = ratio * adv[..., 0], ratio.clamp(1 - eps_clip, 1 + eps_clip) * adv[..., 0]
                l_rew = -torch.min(s1, s2).mean()
                l_safe = torch.mean(ratio * adv[..., 1] * lag)
                l_actor = resc * (l_rew + l_safe)
                vfs = [(ret[..., i] - critics[i](o).flatten()).pow(2).mean() for i in range(2)]
                loss = l_actor + 0.25 * (vfs[0] + vfs[1])
                optim.zero_grad()
                loss.backward()
                nn.utils.clip_grad_norm_(params, max_norm=0.5)
                optim.step()
                # the reference's logger.store(...) read-backs (ppo_lag.py:196-211, 166-170, 242-247)
                _ = (l_safe.item(), l_rew.item(), l_actor.item(), (lo - logp).mean().item(), vfs[0].item(), vfs[1].item(),
                     (vfs[0] + vfs[1]).item(), loss.item(), d.entropy().mean().item())
                steps += 1
        return steps
    rng = np.random.default_rng(0)
    one_update(rng)                                   # warm-up (allocator, kernel selection)
    torch.cuda.synchronize()
    n, steps, t0 = 0, 0, time.perf_counter()
    while n < updates and time.perf_counter() - t0 < seconds:
        steps = one_update(rng); n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "updates/s", "kind": "port on ROCm torch", "device": torch.cuda.get_device_name(dev),
            "grad_steps_per_update": steps, "us_per_grad_step": dt / max(n * steps, 1) * 1e6,
            "sample": f"{n} full updates of the same 20k-row workload with stock torch.nn / torch.optim on the GPU (torch {torch.__version__}), "
                      "one warm-up update before; host GAE scan and per-step .item() read-backs as in the reference"}


def gpu_config0(inputs, seed, steps=8):
    """BASELINE configs[0]'s shape (128x128 MLPs, otherwise the headline workload) through the HIP path, beside
    `cpu_baseline_c0`: the reference's CPU-runnable configuration on both sides of this box."""
    from fsrl_amd.engine import Engine, EngineConfig
    eng = Engine(EngineConfig(obs_dim=OBS, act_dim=ACT, hidden=128, env_num=ENVS, buffer_size=100000, max_grad_norm=0.5,
                              target_kl=None))
    theta = orthogonal_theta(seed, eng.n_params, hid=128)
    obs, act, rew, cost, term, trunc = inputs
    ids = np.arange(ENVS)
    for t in range(NROWS // ENVS):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    eng.sync()
    lag, resc = np.array([0.75]), 1.0 / 1.75

    eng.set_params(theta); eng.optim_reset(); eng.state_snapshot()

    def one(k):
        eng.state_restore()
        return eng.ppo_update(lag, resc, BATCH, REPEAT, perms=None, seed=k + 1)[0]
    one(0)
    eng.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        st = one(k + 1)
    eng.sync()
    dt = (time.perf_counter() - t0) / steps
    eng.close()
    return {"value": 1.0 / dt, "unit": "updates/s", "ms_per_update": dt * 1e3, "us_per_step": dt * 1e6 / st.shape[0],
            "workload": "BASELINE configs[0] shape: 128x128 MLPs, N=20000, batch 256, repeat 4, grad-clip 0.5"}


def env_bound(envs, workers, busy_us, cpus):
    """Upper bound on env-steps/s of a vector env whose every env step burns `busy_us` of host time: a vector step takes
    ceil(envs / lanes) serial env steps, lanes = the worker processes that can run at once = min(workers, usable CPUs)
    (4 workers x 8 envs x 100 us => 40 k/s however many CPUs there are; 32 workers on 16 CPUs => 2 rounds => 160 k/s)."""
    if busy_us <= 0:
        return None
    lanes = max(1, min(int(workers) if workers > 0 else 1, int(cpus)))
    return envs / (-(-envs // lanes) * busy_us * 1e-6)


def eng_cfg_obs(agent):
    return int(agent.policy.engine.cfg.obs_dim)


def end_to_end(local_rank, seed, seconds=6.0, device_actor=False, workers=0, busy_us=0.0, envs=ENVS, cap_workers=False):
    """Secondary figure (outside the timed region): the whole training loop of
    OnpolicyAgent.learn -- host collector over a SYNTHETIC SafetyCarCircle-shaped vector env
    (300-step episodes, one episode per env and collect) feeding the HIP-resident store, one
    device update per collect -- reported like the reference's `train_speed` (env-steps/s).
    workers == 0: the in-process vector env (a zero-cost env: the collector's own ceiling).  workers > 0: the
    shared-memory multi-process env (fsrl_amd/env/shmem.py: `workers` processes over `envs` envs, `busy_us` of host time
    burnt per env step in the worker -- SURVEY 8(d): 0 and ~100 us to mimic PyBullet, 4 / 32 worker processes)."""
    from fsrl_amd.agent import PPOLagAgent
    from fsrl_amd.data import FastCollector, HipVectorReplayBuffer
    from fsrl_amd.env import ShmemVectorEnv, SyntheticSafetyVectorEnv
    from fsrl_amd.trainer import OnpolicyTrainer
    if workers > 0:
        env = ShmemVectorEnv(env_num=envs, workers=workers, obs_dim=OBS, act_dim=ACT, episode_len=300, seed=seed, busy_us=busy_us,
                             cap_workers=cap_workers)
    else:
        env = SyntheticSafetyVectorEnv(env_num=envs, obs_dim=OBS, act_dim=ACT, episode_len=300, seed=seed, busy_us=busy_us)
    agent = PPOLagAgent(env, cost_limit=10, device=f"cuda:{local_rank}", seed=seed, hidden_sizes=(HID, HID),
                        max_grad_norm=0.5, training_num=envs)
    agent.policy.train()
    buf = HipVectorReplayBuffer(agent.policy.engine, 100000, envs)
    col = FastCollector(agent.policy, env, buf, exploration_noise=True, device_actor=device_actor, split_phase="auto")
    tr = OnpolicyTrainer(agent.policy, col, None, max_epoch=10**6, batch_size=BATCH, cost_limit=10,
                         step_per_epoch=6000, repeat_per_collect=REPEAT, episode_per_collect=envs,
                         verbose=False)
    tr.reset()
    t0 = time.perf_counter()
    collects, update_s = 0, 0.0
    while time.perf_counter() - t0 < seconds:
        st = tr.train_step()
        t1 = time.perf_counter()
        tr.policy_update_fn(st)
        update_s += time.perf_counter() - t1
        collects += 1
    dt = time.perf_counter() - t0
    from fsrl_amd.parallel import usable_cpus
    kind = (f"shared-memory multi-process vector env: {workers} worker processes, {busy_us:g} us of host time per env step"
            if workers > 0 else "in-process vector env, zero-cost step")
    procs = int(getattr(env, "workers", workers)) if workers > 0 else 0      # worker PROCESSES (capped at the usable CPUs)
    out = {"env": "synthetic SafetyCarCircle-shaped dynamics (not PyBullet); " + kind, "envs": envs, "workers": workers,
           "worker_processes": procs, "worker_mode": getattr(env, "worker_mode", None), "busy_us": busy_us, "host_cpus_usable": usable_cpus(),
           "handshake": (("futex generation word + completion counter per lane (libfsrl_env.so), spin "
                          f"{getattr(env, 'spin_us', 0):g} us before sleeping") if workers > 0 else None),
           "split_phase": bool(col.split_phase),
           "collector_loop": ((("native split-phase (fsrl_collect_episodes_split: one library call per collect, the two lanes "
                                 "alternate)" if col.split_phase else "native (fsrl_collect_episodes: one library call per collect)")
                               if col.native_loop is True
                               else "native (fsrl_collect_run: Python sees episode boundaries only)")
                              if (device_actor and col.native_loop and hasattr(env, "native_desc")
                                  and (col.native_loop is True or not col.split_phase))
                              else "interpreted, one library call per vector step" if device_actor else "interpreted, host actor"),
           "env_bound_env_steps_per_s": env_bound(envs, procs, busy_us, usable_cpus()),
           "actor": "device (fsrl_collect_step: one call per vector step, library RNG; served by a resident workgroup through a doorbell in pinned memory: one kernel launch per collect)" if device_actor else "host mirror (torch CPU, torch RNG)",
           "collects": collects, "env_steps_per_s": col.collect_step / dt,
           "frac_of_env_bound": (col.collect_step / dt / env_bound(envs, procs, busy_us, usable_cpus())
                                 if busy_us > 0 else None),
           "collector_only_env_steps_per_s": col.collect_step / col.collect_time,
           "update_ms_per_collect": update_s / collects * 1e3,
           "policy_updates_per_s": collects / dt}
    if device_actor:        # r6: the actor as a resident workgroup (fsrl_actor_set_resident): launches vs calls served through the doorbell
        st = agent.policy.engine.actor_resident_stats()
        out["actor_resident"] = {"kernel_launches": st["launches"], "calls_served": st["requests"],
                                 "per_collect": round(st["launches"] / max(collects, 1), 2)}
        if workers == 0:    # ... and what one actor call of this vector env costs, resident against one launch per call (2 x 2000 calls)
            eng_, obs_ = agent.policy.engine, np.zeros((envs, eng_cfg_obs(agent)), np.float32)
            us = {}
            for name, on in (("resident", True), ("launched", False)):
                eng_.actor_set_resident(on)
                for _ in range(200):
                    eng_.actor_sample(obs_)
                t0 = time.perf_counter()
                for _ in range(2000):
                    eng_.actor_sample(obs_)
                us[name] = round((time.perf_counter() - t0) / 2000 * 1e6, 2)
            eng_.actor_set_resident(True)
            out["actor_call_us"] = us
    agent.policy.engine.close()
    if hasattr(env, "close"):
        env.close()
    return out


def multi_seed(seeds=3, steps=6):
    """Secondary figure: `seeds` independent agents on ONE GPU (one host thread + one library context each, own HIP
    streams).  One agent's dependent kernels leave latency gaps another agent fills: aggregate updates/s."""
    import threading
    from fsrl_amd.engine import Engine, EngineConfig
    agents = []
    for s_ in range(seeds):
        eng = Engine(EngineConfig(obs_dim=OBS, act_dim=ACT, hidden=HID, env_num=ENVS, buffer_size=100000, max_grad_norm=0.5,
                                  target_kl=None))
        theta = orthogonal_theta(s_, eng.n_params)
        obs, act, rew, cost, term, trunc = make_inputs(s_)
        ids = np.arange(ENVS)
        for t in range(NROWS // ENVS):
            eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
        eng.sync()
        agents.append((eng, theta))
    lag, resc = np.array([0.75]), 1.0 / 1.75

    def work(i, n):
        eng, theta = agents[i]
        for k in range(n):
            eng.set_params(theta); eng.optim_reset()
            eng.ppo_update(lag, resc, BATCH, REPEAT, perms=None, seed=1000 * i + k + 1)

    for i in range(seeds):
        work(i, 1)
    th = [threading.Thread(target=work, args=(i, steps)) for i in range(seeds)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    for eng, _ in agents:
        eng.close()
    return {"seeds_per_gpu": seeds, "value": seeds * steps / dt, "unit": "updates/s (aggregate)",
            "note": "same configs[1] workload per agent; not the headline value (that is one agent per GPU)"}


def grouped(k=4, updates=5):
    """Secondary figure: k independent agents of the SAME configs[1] workload on ONE GPU, stepped in lock step by the
    grouped launches (fsrl_group_ppo_update: grid.y = agent): aggregate updates/s.  Not the headline (one agent per GPU)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_group import run
    r = run(k, updates, 0.5)
    r["unit"] = "updates/s (aggregate over the agents on this GPU)"
    r["fwdbwd_tflops_all_agents"] = None
    return r


def no_clip_variant(theta, inputs, steps=6):
    """Secondary figure: the same update with max_grad_norm off -- PPOLagAgent's default (ppo_lag_agent.py:97; the 0.5 of
    the headline comes from the reference's config file, ppol_cfg.py:21).  Without a global gradient norm nothing stands
    between a reduced gradient element and its Adam update: the weight-gradient kernel applies it, 2 launches per step."""
    from fsrl_amd.engine import Engine, EngineConfig
    eng = Engine(EngineConfig(obs_dim=OBS, act_dim=ACT, hidden=HID, env_num=ENVS, buffer_size=100000, max_grad_norm=None,
                              target_kl=None))
    obs, act, rew, cost, term, trunc = inputs
    ids = np.arange(ENVS)
    for t in range(NROWS // ENVS):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    eng.sync()
    lag, resc = np.array([0.75]), 1.0 / 1.75

    eng.set_params(theta); eng.optim_reset(); eng.state_snapshot()

    def one(k):
        eng.state_restore()
        return eng.ppo_update(lag, resc, BATCH, REPEAT, perms=None, seed=k + 1)[0]
    one(0)
    t0 = time.perf_counter()
    for k in range(steps):
        st = one(k + 1)
    eng.sync()
    dt = (time.perf_counter() - t0) / steps
    eng.close()
    return {"value": 1.0 / dt, "unit": "updates/s", "ms_per_update": dt * 1e3, "launches_per_step": 2,
            "us_per_step": dt * 1e6 / st.shape[0],
            "note": "max_grad_norm=None (agent default): Adam fused into the weight-gradient kernel; not the headline config"}


def layered_variant(inputs, steps=3):
    """The headline workload (N 20 000, batch 256, 4 passes, clip 0.5) on networks the fused kernels do not hold: the layered
    contexts of include/fsrl_hip.h fsrl_config.n_hidden (one MFMA GEMM launch per Linear, 2 L + 5 launches per step).  The
    (256, 256) row runs the headline network itself through those kernels (force_layered) -- what the fusion is worth."""
    import torch
    from fsrl_amd.engine import Engine, EngineConfig
    obs, act, rew, cost, term, trunc = inputs
    ids = np.arange(ENVS)
    lag, resc = np.array([0.75]), 1.0 / 1.75
    out = {}
    for hid in ((256, 256), (256, 256, 256), (512, 512), (1024, 1024)):
        eng = Engine(EngineConfig(obs_dim=OBS, act_dim=ACT, hidden_sizes=hid, force_layered=True, env_num=ENVS, buffer_size=100000,
                                  max_grad_norm=0.5, target_kl=None))
        for t in range(NROWS // ENVS):
            eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
        torch.manual_seed(1)
        eng.set_params((0.05 * torch.randn(eng.n_params)).numpy()); eng.optim_reset(); eng.state_snapshot()

        def one(k):
            eng.state_restore()
            return eng.ppo_update(lag, resc, BATCH, REPEAT, perms=None, seed=k + 1)[0]
        one(0)
        eng.sync()
        t0 = time.perf_counter()
        for k in range(steps):
            st = one(k + 1)
        eng.sync()
        dt = (time.perf_counter() - t0) / steps
        eng.close()
        L = len(hid)
        out["x".join(map(str, hid))] = {"value": 1.0 / dt, "unit": "updates/s", "ms_per_update": dt * 1e3,
                                        "us_per_step": dt * 1e6 / st.shape[0], "launches_per_step": 2 * L + 5,
                                        "n_params": int(eng.n_params)}
    out["note"] = ("hidden_sizes outside the fused kernels' reach (layered contexts: PPO-Lag here; FOCOPS, CPO, TRPO-Lag run them too): "
                   "layer-by-layer GEMM launches, activations in HBM; 256x256 = the headline network through the same kernels")
    return out


def pmc_traffic():
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE, separate runs; tools/capture_profiles.sh + tools/collect_profiles.py apply
    the guide's KiB unit and gfx950 x2 FETCH correction).  Counters cannot be read inside this process,
    so the figure is the one measured when the profile was captured; None when there is none."""
    here = os.path.dirname(os.path.abspath(__file__))
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        f = os.path.join(here, "profiles", f"{tag}_pmc_traffic.json")
        if os.path.exists(f):
            j = json.load(open(f))
            meta = j.get("_meta", {})
            for name, v in j.items():
                if name.startswith("void ppo_fwd_bwd_kernel<256"):
                    return (float(v["hbm_bytes_per_launch"]), f"profiles/{tag}_pmc_traffic.json (rocprofv3 --pmc, separate pass)",
                            meta.get("head"), meta.get("csrc_sha16"))
    return None


def kl_on_variant(theta, inputs, seed, steps=10):
    """"disable [the KL early stop] for throughput runs; report both".  The same workload with the reference's
    default target_kl = 0.02 (ppo_lag_agent.py:95): the pass-level check (ppo_lag.py:251-255) costs one 24-byte read-back per
    pass, and an update whose pass-mean KL exceeds 1.5 x target_kl ends early (fewer gradient steps: both are reported)."""
    from fsrl_amd.engine import Engine, EngineConfig
    eng = Engine(EngineConfig(obs_dim=OBS, act_dim=ACT, hidden=HID, env_num=ENVS, buffer_size=100000, max_grad_norm=0.5,
                              target_kl=0.02))
    obs, act, rew, cost, term, trunc = inputs
    ids = np.arange(ENVS)
    for t in range(NROWS // ENVS):
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    eng.sync()
    lag, resc = np.array([0.75]), 1.0 / 1.75
    eng.set_params(theta); eng.optim_reset(); eng.state_snapshot()

    def one(k):
        eng.state_restore()
        return eng.ppo_update(lag, resc, BATCH, REPEAT, perms=None, seed=1000 * seed + k + 1)
    one(0)
    eng.sync()
    n_steps, stopped = [], 0
    t0 = time.perf_counter()
    for k in range(steps):
        st, sp = one(k + 1)
        n_steps.append(int(st.shape[0])); stopped += int(sp >= 0)
    eng.sync()
    dt = (time.perf_counter() - t0) / steps
    eng.close()
    return {"value": 1.0 / dt, "unit": "updates/s", "ms_per_update": dt * 1e3, "target_kl": 0.02,
            "grad_steps_per_update_mean": float(np.mean(n_steps)), "updates_stopped_early": stopped, "updates": steps,
            "us_per_step": dt * 1e6 / max(float(np.mean(n_steps)), 1.0),
            "note": "the headline workload with the KL early stop ON (reference default 0.02); the headline value has it off"}


def cpo_c2_leg():
    """BASELINE configs[2]: CPO on the SafetyPointGoal shape (obs 60, act 2, 256x256, N = 20 000 full batch, CG 10, 4 repeats)
    -- the body of tools/bench_trust.py; roofline of the whole update + the oracle on 1 of its 4 repeats as cpu_baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_trust
    return bench_trust.run("cpo", 60, 2, 256, timed=5, emit=False, no_cpu=False)


def trpo_c1_leg():
    """TRPO-Lagrangian on the configs[1] shape (obs 8, 256x256, N = 20 000 full batch): the FVP + line-search path's other user."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_trust
    return bench_trust.run("trpo", 8, 2, 256, ep=250, timed=5, emit=False, no_cpu=False)


def sac_c3_leg():
    """BASELINE configs[3]: SAC-Lagrangian, SafetyAntRun shape, 1 M-row replay store resident in HBM, batch 1024, n_step 2:
    1 000 consecutive updates -- the body of tools/bench_sac.py."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_sac
    return bench_sac.main(["--updates", "1000", "--cpu-updates", "10"], emit=False)


def configs_summary(out):
    """Every BASELINE configuration's figures in one compact object (<= 600 characters), written as the LAST key of the
    line so that a record which keeps only the tail of stdout still carries them: time, whole-update fp32-MFMA fraction,
    memory-side traffic (committed PMC pass) and the CPU port's rate on this box."""
    def g(d, *ks):
        for k in ks:
            d = d.get(k) if isinstance(d, dict) else None
        return d

    def r(x, n=3):
        return None if not isinstance(x, (int, float)) else round(float(x), n)
    c = {}
    roof = out.get("roofline") if isinstance(out.get("roofline"), dict) else {}
    c["c1_ppo"] = {"ms": r(out.get("ms_per_step")), "step_us": r(roof.get("step_us"), 2), "frac": r(roof.get("frac"), 4),
                   "cpu_updates_s": r(g(out, "cpu_baseline", "value"))}
    c0 = out.get("gpu_c0")
    if isinstance(c0, dict) and "ms_per_update" in c0:
        c["c0_ppo128"] = {"ms": r(c0["ms_per_update"]), "cpu_updates_s": r(g(out, "cpu_baseline_c0", "value"))}
    for key, name in (("cpo_c2", "c2_cpo"), ("trpo_c1", "c1_trpo")):
        d = out.get(key)
        if isinstance(d, dict):
            tb = g(d, "roofline", "traffic")
            c[name] = ({"err": 1} if "error" in d else
                       {"ms": r(d.get("hip_ms_per_update"), 2), "frac": r(g(d, "roofline", "frac")),
                        "traffic_gb": r(tb / 1e9 if tb else None, 1), "cpu_updates_s": r(g(d, "cpu_baseline", "value"), 4)})
    d = out.get("sac_c3")
    if isinstance(d, dict):
        tb = g(d, "roofline", "traffic")
        c["c3_sac"] = ({"err": 1} if "error" in d else
                       {"us": r(d["ms_per_update"] * 1e3 if "ms_per_update" in d else None, 1), "frac": r(g(d, "roofline", "frac")),
                        "traffic_mb": r(tb / 1e6 if tb else None, 1), "launches": d.get("launches_per_update"),
                        "cpu_updates_s": r(g(d, "cpu_baseline", "value"), 2)})
    d = out.get("kl_on")
    if isinstance(d, dict) and "ms_per_update" in d:
        c["kl_on"] = {"ms": r(d["ms_per_update"]), "steps": r(d.get("grad_steps_per_update_mean"), 1),
                      "stopped": d.get("updates_stopped_early"), "of": d.get("updates")}
    # r6: BOTH halves of the headline metric (BASELINE.json: "env-steps/sec + policy-updates/sec") where the record keeps them:
    # the training loop's env-steps/s (in-process zero-cost env, device actor), the update rate, and how close the worker-process
    # env at 32 workers x 100 us per step comes to its own bound; at N > 1 the whole job's sum over the ranks
    d = out.get("end_to_end")
    if isinstance(d, dict) and "env_steps_per_s" in d:
        e = {"env_steps_s": r(d["env_steps_per_s"], 0), "updates_s": r(out.get("value"), 1)}
        for w in out.get("end_to_end_shmem") or []:
            if isinstance(w, dict) and w.get("workers") == 32 and w.get("busy_us") == 100.0 and "frac_of_env_bound" in w:
                e["w32_b100_frac_of_bound"] = r(w["frac_of_env_bound"], 2)
        c["e2e"] = e
    d = out.get("end_to_end_job")
    if isinstance(d, dict) and "env_steps_per_s" in d:
        c["job"] = {"env_steps_s": r(d["env_steps_per_s"], 0), "updates_s": r(out.get("value"), 1), "ranks_ok": d.get("ranks_ok")}
    d = out.get("torch_rocm_baseline")
    if isinstance(d, dict) and "value" in d:
        c["c1_ppo"]["torch_gpu_updates_s"] = r(d["value"], 2)
    return c


class Legs:
    """Secondary legs of the bench line.  The headline dict is complete BEFORE any of them runs; every leg runs inside
    try/except (its key becomes {"error": ...} on failure) under a watchdog thread with a per-leg deadline: when a leg
    hangs (a collective another rank never joins, a worker process that never answers), rank 0 prints the line as it
    stands -- exactly one JSON line in every case -- and every rank leaves with os._exit(0), so one stuck leg cannot
    forfeit the measured headline or stall the other ranks.  Test hooks: FSRL_BENCH_FAIL_LEG=<key> raises inside that leg,
    FSRL_BENCH_HANG_LEG=<key> sleeps forever inside it."""

    def __init__(self, out, rank, total_budget_s):
        import threading
        self.out, self.rank = out, rank
        self.lock = threading.Lock()
        self.printed = False
        self.leg, self.leg_deadline = None, None
        self.t_end = time.monotonic() + total_budget_s
        self.fail = os.environ.get("FSRL_BENCH_FAIL_LEG", "").split(",")
        self.hang = os.environ.get("FSRL_BENCH_HANG_LEG", "").split(",")
        th = threading.Thread(target=self._watch, daemon=True)
        th.start()

    def _watch(self):
        while True:
            time.sleep(0.25)
            with self.lock:
                dl, leg = self.leg_deadline, self.leg
            if dl is not None and time.monotonic() > dl:
                if self.out is not None and leg is not None:
                    self.out[leg] = {"error": "watchdog: leg did not finish within its deadline; later legs skipped"}
                self.emit()
                sys.stdout.flush(); sys.stderr.flush()
                os._exit(0)

    def emit(self):
        with self.lock:
            if self.printed:
                return
            self.printed = True
        if self.rank == 0 and self.out is not None:
            self.out.pop("configs", None)
            self.out["configs"] = configs_summary(self.out)       # LAST key: the stored tail of the line always holds it
            print(json.dumps(self.out), flush=True)

    def remaining(self):
        return self.t_end - time.monotonic()

    def run(self, key, fn, timeout_s, store=True):
        """fn() under the leg's deadline; returns its value or None.  A leg is skipped when the total budget is spent."""
        if self.remaining() <= 1.0:
            if store and self.out is not None:
                self.out[key] = {"error": "skipped: secondary-leg time budget spent"}
            return None
        with self.lock:
            self.leg, self.leg_deadline = key, time.monotonic() + min(timeout_s, max(self.remaining(), 1.0))
        try:
            if key in self.fail:
                raise RuntimeError(f"FSRL_BENCH_FAIL_LEG={key} (test hook)")
            if key in self.hang:
                while True:
                    time.sleep(1.0)
            val = fn()
            if store and self.out is not None:
                self.out[key] = val
            return val
        except BaseException as e:                                   # noqa: BLE001 -- a leg must never take the headline down
            if isinstance(e, KeyboardInterrupt):
                raise
            if store and self.out is not None:
                self.out[key] = {"error": f"{type(e).__name__}: {e}"[:400]}
            return None
        finally:
            with self.lock:
                self.leg, self.leg_deadline = None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip every secondary leg (profiling runs)")
    # test hooks (tests/test_gpu_facade.py): run the N > 1 code path on a one-GPU box -- every rank on device 0, gloo
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--share-gpu", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    comm_fallback = None
    nccl_pg = None                   # the RCCL group of the timing exchange (N > 1, when it started on every rank)
    if world > 1:
        import torch.distributed as dist
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        # The control plane is ALWAYS a gloo group (barrier, rank agreement, the id broadcast of the library's communicator): it is
        # created once, first, with an explicit timeout -- no second init_process_group on the same MASTER_PORT whatever RCCL does.
        from datetime import timedelta
        dist.init_process_group("gloo", timeout=timedelta(seconds=180))
        if args.backend == "nccl":
            # RCCL (a SECOND group) carries the timing exchange when it starts on every rank.  It has never run with more than one
            # rank where this was built (no multi-GPU lease), so it is proven with one collective before anything depends on it, and
            # the ranks AGREE on the outcome over gloo: one rank raising while another succeeds cannot leave them in different
            # groups.  A rank that hangs inside RCCL instead of raising is ended by the group's own timeout (90 s).  The ranks' work
            # has no data-path collective, so the figure is the same either way; the line says which exchange ran.
            ok, why = 1, None
            try:
                # no device_id: the communicator is created lazily by the first collective on the device torch.cuda.set_device chose
                # (eager initialisation of a SECOND group next to a gloo default group is the less travelled path)
                nccl_pg = dist.new_group(backend="nccl", timeout=timedelta(seconds=90))
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe, group=nccl_pg)
                torch.cuda.synchronize()
                assert float(probe.item()) == float(world)
            except Exception as e:                                 # noqa: BLE001
                ok = 0
                why = f"{type(e).__name__}: {str(e).strip().splitlines()[-1][:160] if str(e).strip() else ''}"
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)            # over gloo: every rank takes the same branch
            if int(flag.item()) != 1:
                comm_fallback = why or "another rank could not start RCCL"
                nccl_pg = None
                args.backend = "gloo"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from fsrl_amd.engine import Engine, EngineConfig
    seed = rank
    eng = Engine(EngineConfig(obs_dim=OBS, act_dim=ACT, hidden=HID, env_num=ENVS, buffer_size=100000,
                              max_grad_norm=0.5, target_kl=None), device=local_rank)
    theta = orthogonal_theta(seed, eng.n_params)
    eng.set_params(theta)
    inputs = make_inputs(seed)
    obs, act, rew, cost, term, trunc = inputs
    ids = np.arange(ENVS)
    for t in range(NROWS // ENVS):   # fill the HBM-resident store (not timed)
        eng.push(ids, obs[t], act[t], rew[t], cost[t], term[t], trunc[t], obs[t + 1])
    eng.sync()
    assert len(eng) == NROWS
    lag, resc = np.array([0.75]), 1.0 / 1.75

    def barrier():
        if dist is not None:
            dist.barrier()
        eng.sync()
        torch.cuda.synchronize()

    eng.optim_reset()
    eng.state_snapshot()             # orthogonal-init weights + fresh Adam state, checkpointed in HBM

    def one_update(k):
        # every timed update is the same workload: orthogonal-init weights, fresh Adam state
        # (re-training 20x on ONE synthetic buffer with the KL stop off would diverge to inf;
        # the restore -- three device-to-device copies, 2.4 MB, on the compute stream -- stays inside the timed region)
        eng.state_restore()
        stats, _ = eng.ppo_update(lag, resc, BATCH, REPEAT, perms=None, seed=1000 * seed + k + 1)
        return stats

    for k in range(args.warmup):
        one_update(k)
    # ---- timed region: exactly K updates, un-instrumented
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        stats = one_update(args.warmup + k)
    barrier()
    dt = time.perf_counter() - t0
    grad_steps = stats.shape[0]
    assert np.isfinite(stats).all()
    per_rank, dt_local = None, dt
    if dist is not None:
        # the contract's max-over-ranks time, and the SURVEY 8(e) exchange: every rank's figures gathered once
        from fsrl_amd import parallel
        tt = torch.tensor([dt], device="cuda" if nccl_pg is not None else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=nccl_pg)     # group None = the gloo control plane
        dt = float(tt.item())
        per_rank = parallel.allgather_metrics({"rank": float(rank), "seed": float(seed), "updates": float(args.steps),
                                               "grad_steps": float(grad_steps * args.steps), "seconds": dt_local,
                                               "updates_per_s": args.steps / dt_local})

    # ---- the headline dict: complete here, before the roofline leg and every secondary leg
    out = None
    if rank == 0:
        ups = args.steps / dt
        out = {
            "metric": "policy-updates/sec (PPO-Lagrangian update(), SafetyCarCircle-v0 shape, 256x256 MLP, 20k-step buffer)",
            "value": ups * world, "unit": "updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # < 120 characters (the driver's record cuts the string): what is timed first, the shape after it
            "config": {"workload": "configs[1] PPO-Lag update: repeat 4, clip 0.5, KL stop off, batch 256, N=20000 (20x1000), "
                                   "obs 8/act 2, 256x256",
                       "seeds_per_gpu": 1, "parallelism": f"independent agents x{world}"},
            "grad_steps_per_update": int(grad_steps),
            "grad_steps_per_s": ups * grad_steps * world,
            "buffer_rows_per_s": ups * NROWS * world,
        }
        if per_rank is not None:
            out["ranks_seen"] = len(per_rank)
            out["per_rank_updates_per_s"] = [round(r["updates_per_s"], 3) for r in sorted(per_rank, key=lambda r: r["rank"])]
            out["sum_of_rank_rates"] = sum(r["updates_per_s"] for r in per_rank)
            out["timing_exchange"] = ("rccl (torch.distributed nccl)" if args.backend == "nccl" else "gloo") + \
                (f"; nccl could not start here ({comm_fallback})" if comm_fallback else "")
    legs = Legs(out, rank, total_budget_s=float(os.environ.get("FSRL_BENCH_LEG_BUDGET_S", "420")))

    # ---- roofline of the dominant kernel: HIP events around every ppo_fwd_bwd_kernel launch on
    #      the library's compute stream, over K more updates of the same workload
    def roofline_leg():
        eng.set_profiling(True)
        k_ms, k_raw, k_n, learn_ms, proc_list = 0.0, 0.0, 0, 0.0, []
        step_ms = {"fwdbwd": 0.0, "wgrad": 0.0, "adam": 0.0}
        prof_steps = max(1, min(args.steps, 5))
        for k in range(prof_steps):
            one_update(10_000 + k)
            tm = eng.last_timing()
            k_ms += tm["fwdbwd_ms"]; k_raw += tm["fwdbwd_raw_ms"]; k_n += tm["fwdbwd_launches"]
            learn_ms += tm["learn_ms"]; proc_list.append(tm["process_ms"])
        eng.set_profiling(False)
        # median x count: one update whose process_fn catches a host hiccup (seen: 7 ms instead of 0.25) must not move step_us
        proc_ms = float(np.median(proc_list)) * prof_steps
        avg_launch_s = (k_ms / max(k_n, 1)) * 1e-3
        rows_avg = NROWS / (grad_steps / REPEAT)        # 77 launches of 256 rows + 1 of 288 per pass
        fl = flops_fwdbwd_launch(rows_avg)
        achieved = fl / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        r = {"bound": "mfma", "kernel": "ppo_fwd_bwd_kernel<256>", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS,
             "unit": "TFLOP/s", "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": None,
             "avg_launch_us": avg_launch_s * 1e6, "avg_launch_us_raw_event_bracket": k_raw / max(k_n, 1) * 1e3,
             "launches_timed": int(k_n), "flops_per_launch": fl}
        # the bound that actually applies to a chain of dependent ~0.2 GFLOP launches ("report the achieved
        # fraction of the latency floor as well"): per optimiser step, the floors of its dependent launches -- MEASURED HERE,
        # in this run, on this device (fsrl_launch_floors: empty kernels with the step's three grids, block sizes and LDS
        # footprints, 300 triples back to back, ~5 ms) -- + the step's MFMA work at the chip's peak
        floors = eng.launch_floors(BATCH, 300)
        floors_us = floors["triple"]
        step_flops = fl + flops_wgrad_launch(rows_avg)
        floor_us = floors_us + step_flops / (F32_MFMA_PEAK_TFLOPS * 1e12) * 1e6
        # per optimiser step in the UN-instrumented timed region: the update minus process_fn (timed here with HIP events; the
        # event brackets of the profiling mode themselves cost ~8 us per step, so its own learn time is not used)
        step_us = (dt / args.steps * 1e3 - proc_ms / prof_steps) * 1e3 / grad_steps
        r["latency_floor_us"] = floor_us
        r["latency_floor_parts_us"] = {"three_empty_launches_measured_in_this_run": floors_us, "each_grid_behind_itself": floors,
                                       "mfma_at_peak": floor_us - floors_us}
        r["step_us"] = step_us
        r["frac_of_latency_floor"] = floor_us / step_us if step_us > 0 else None
        # what the state restore at the head of every timed update costs (three device-to-device copies, 2.4 MB, on the compute
        # stream): 50 of them back to back, one synchronisation
        eng.sync()
        t_r = time.perf_counter()
        for _ in range(50):
            eng.state_restore()
        eng.sync()
        r["state_restore_us_inside_every_timed_update"] = (time.perf_counter() - t_r) / 50 * 1e6
        pmc = pmc_traffic()
        if pmc is not None:
            r["traffic"], r["traffic_source"], r["traffic_profile_head"], r["traffic_profile_csrc_sha16"] = pmc
            r["csrc_sha16"] = csrc_sha16()
            r["traffic_profile_current"] = (r["traffic_profile_csrc_sha16"] == r["csrc_sha16"]) if r["traffic_profile_csrc_sha16"] else None
            if r["traffic_profile_current"] is not True:
                r["traffic_warning"] = ("the PMC capture was taken on other kernel sources than this run's (or carries no source "
                                        "hash): the traffic figure may be stale")
        if out is not None:
            out["phase_ms_instrumented"] = {
                "process_fn": proc_ms / prof_steps, "learn": learn_ms / prof_steps,
                "note": "profiling mode: three event records per optimiser step inside learn (~8 us each step), so learn here "
                        "exceeds the un-instrumented ms_per_step; only process_fn is used (step_us above)"}
        return r

    legs.run("roofline", roofline_leg, 60.0)

    if dist is not None and not args.headline_only:
        from fsrl_amd import parallel
        # ---- the other half of the headline metric at N > 1: every rank's whole training loop (host collector -> HIP store
        #      -> update) side by side for a few seconds on its own slice of the host cores; job env-steps/s = sum over ranks.
        #      A rank whose loop fails contributes NaNs, so the gather below is still joined by every rank.
        cores = parallel.pin_rank_cores(local_rank if not args.share_gpu else rank, world)
        # BASELINE configs[4]: 32 envs per rank (one seed per GPU)
        e2e = legs.run("end_to_end_rank", lambda: end_to_end(local_rank, seed, seconds=4.0, device_actor=True, envs=32), 90.0, store=False)
        mine = {"rank": float(rank), "cores": float(len(cores)),
                "env_steps_per_s": e2e["env_steps_per_s"] if e2e else float("nan"),
                "policy_updates_per_s": e2e["policy_updates_per_s"] if e2e else float("nan")}

        def job_leg():
            rows = parallel.allgather_metrics(mine)
            rows.sort(key=lambda r: r["rank"])
            ok = [r for r in rows if r["env_steps_per_s"] == r["env_steps_per_s"]]
            return {"env": e2e["env"] if e2e else None, "actor": e2e["actor"] if e2e else None,
                    "envs_per_rank": e2e["envs"] if e2e else None, "ranks": len(rows), "ranks_ok": len(ok),
                    "env_steps_per_s": sum(r["env_steps_per_s"] for r in ok),
                    "policy_updates_per_s": sum(r["policy_updates_per_s"] for r in ok),
                    "per_rank_env_steps_per_s": [round(r["env_steps_per_s"], 1) for r in rows],
                    "host_cores_per_rank": int(rows[0]["cores"])}
        legs.run("end_to_end_job", job_leg, 60.0)

        # ---- LAST (it is the one leg that has never run on hardware): the same exchange through the C ABI's own RCCL
        #      communicator (fsrl_comm_init + fsrl_metrics_allreduce), checked against torch.distributed's sum.  Every rank
        #      first proves it can reach RCCL, so a rank that cannot does not leave the others inside ncclCommInitRank; a
        #      rank that still hangs there trips the watchdog (30 s), which prints the line and ends every rank.
        if nccl_pg is not None and os.environ.get("FSRL_BENCH_LIB_COMM", "1") != "0":
            def lib_comm_leg():
                try:
                    eng.comm_unique_id(); ok, why = 1.0, None
                except Exception as e:                          # noqa: BLE001
                    ok, why = 0.0, f"unavailable: {e}"
                flag = torch.tensor([ok], device="cuda", dtype=torch.float64)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=nccl_pg)
                if float(flag.item()) != 1.0:
                    return why or "skipped: another rank cannot reach RCCL"
                eng.comm_init_from_torch()
                v = np.array([1.0, float(rank), args.steps / dt_local, float(seed)])
                got = eng.metrics_allreduce(v)
                want = torch.from_numpy(v).cuda()
                dist.all_reduce(want, op=dist.ReduceOp.SUM, group=nccl_pg)
                res = "ok" if np.array_equal(got, want.cpu().numpy()) and eng.comm_info() == (rank, world) else "mismatch"
                eng.comm_destroy()
                return res
            legs.run("lib_metrics_allreduce", lib_comm_leg, 30.0)

    if rank == 0 and world == 1 and not args.headline_only:       # rank 0 at N = 1 only
        if not args.no_cpu_baseline:
            legs.run("cpu_baseline", lambda: cpu_baseline(theta, inputs), 90.0)
            if isinstance(out.get("cpu_baseline"), dict) and "value" in out["cpu_baseline"]:
                out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
            # the reference's own GPU path (stock PyTorch on the device) on THIS chip, beside the CPU figure: context only
            legs.run("torch_rocm_baseline", lambda: torch_rocm_baseline(inputs), 60.0)
            # BASELINE configs[0]: the reference's own CPU-runnable case (128x128, 4 threads: fsrl/config/ppol_cfg.py:11,15)
            legs.run("cpu_baseline_c0", lambda: cpu_baseline(orthogonal_theta(seed, hid=128), inputs, seconds=8.0, threads=4,
                                                             hid=128), 60.0)
            legs.run("gpu_c0", lambda: gpu_config0(inputs, seed), 60.0)
            from fsrl_amd.parallel import usable_cpus
            allc = max(1, int(usable_cpus()))          # "... and with all host cores" (the cgroup quota counts)
            if allc > 4:
                legs.run("cpu_baseline_all_cores", lambda: cpu_baseline(theta, inputs, seconds=6.0, threads=allc), 60.0)
        # every other BASELINE configuration, each with its own roofline + cpu_baseline (configs[2], the TRPO-Lag user of the
        # same path, configs[3]) and the headline with the KL early stop on ("report both")
        legs.run("kl_on", lambda: kl_on_variant(theta, inputs, seed), 60.0)
        legs.run("cpo_c2", cpo_c2_leg, 90.0)
        legs.run("trpo_c1", trpo_c1_leg, 90.0)
        legs.run("sac_c3", sac_c3_leg, 120.0)
        legs.run("no_clip", lambda: no_clip_variant(theta, inputs), 60.0)
        legs.run("end_to_end", lambda: end_to_end(local_rank, seed, device_actor=True), 60.0)
        legs.run("end_to_end_host_actor", lambda: end_to_end(local_rank, seed, seconds=4.0, device_actor=False), 60.0)
        # the host vector-env side of the headline metric: worker processes x simulated step cost
        shm = []
        out["end_to_end_shmem"] = shm
        for w in (4, 32):
            for b in (0.0, 100.0):
                r = legs.run(f"end_to_end_shmem_w{w}_b{int(b)}",
                             lambda: end_to_end(local_rank, seed, seconds=3.0, device_actor=True, workers=w, busy_us=b, envs=32,
                                                cap_workers="auto"),
                             60.0, store=False)
                shm.append(r if r is not None else {"workers": w, "busy_us": b, "error": "leg failed or timed out"})
        # 32 envs at zero step cost with one process per env, as requested (cap_workers=False: the reference's layout) -- what "auto" avoids
        r = legs.run("end_to_end_shmem_w32_b0_uncapped",
                     lambda: end_to_end(local_rank, seed, seconds=3.0, device_actor=True, workers=32, busy_us=0.0, envs=32, cap_workers=False),
                     60.0, store=False)
        shm.append(r if r is not None else {"workers": 32, "busy_us": 0.0, "cap_workers": False, "error": "leg failed or timed out"})
        legs.run("grouped", lambda: grouped(4), 90.0)
        legs.run("grouped_k8", lambda: grouped(8), 90.0)
        legs.run("multi_seed", multi_seed, 90.0)
        legs.run("layered", lambda: layered_variant(inputs), 90.0)
    legs.emit()
    try:
        eng.close()
    except Exception:                                             # noqa: BLE001
        pass
    if dist is not None:
        legs.run("_teardown", lambda: (dist.barrier(), dist.destroy_process_group()), 30.0, store=False)


if __name__ == "__main__":
    main()
