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


def orthogonal_theta(seed, n_params_check=None):
    """PPOLagAgent init (ppo_lag_agent.py:147-153): orthogonal W, zero b, sigma_param = -0.5."""
    import torch
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


def _usable_cpus():
    from fsrl_amd.parallel import usable_cpus
    return usable_cpus()


def cpu_baseline(theta, inputs, seconds=12.0, threads=4):
    """The oracle (torch CPU fp32 port of the reference update) timed on the host cores,
    on a bounded sample: whole updates of the SAME workload until `seconds` elapsed.
    threads = 4: the reference default `thread=4` (fsrl/config/ppol_cfg.py:11)."""
    import torch
    from oracle.ppo_lag import OnPolicyData, PPOLagConfig, PPOLagOracle
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
            "sample": f"{n} full updates (312 grad steps each) of the same 20k-row workload, "
                      f"torch CPU fp32, {threads} threads; host: {os.cpu_count()} cpus visible, "
                      f"{_usable_cpus():g} usable (affinity / cgroup quota)"}


def end_to_end(local_rank, seed, seconds=6.0, device_actor=False, workers=0, busy_us=0.0, envs=ENVS):
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
        env = ShmemVectorEnv(env_num=envs, workers=workers, obs_dim=OBS, act_dim=ACT, episode_len=300, seed=seed, busy_us=busy_us)
    else:
        env = SyntheticSafetyVectorEnv(env_num=envs, obs_dim=OBS, act_dim=ACT, episode_len=300, seed=seed, busy_us=busy_us)
    agent = PPOLagAgent(env, cost_limit=10, device=f"cuda:{local_rank}", seed=seed, hidden_sizes=(HID, HID),
                        max_grad_norm=0.5, training_num=envs)
    agent.policy.train()
    buf = HipVectorReplayBuffer(agent.policy.engine, 100000, envs)
    col = FastCollector(agent.policy, env, buf, exploration_noise=True, device_actor=device_actor)
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
    out = {"env": "synthetic SafetyCarCircle-shaped dynamics (not PyBullet); " + kind, "envs": envs, "workers": workers,
           "busy_us": busy_us, "host_cpus_usable": usable_cpus(),
           "handshake": (("polled sequence numbers" if getattr(env, "spin_us", 0) > 0 else "semaphores") if workers > 0 else None),
           "env_bound_env_steps_per_s": (min(envs, usable_cpus()) / (busy_us * 1e-6) if busy_us > 0 else None),
           "actor": "device (fsrl_collect_step: one call per vector step, library RNG)" if device_actor else "host mirror (torch CPU, torch RNG)",
           "collects": collects, "env_steps_per_s": col.collect_step / dt,
           "collector_only_env_steps_per_s": col.collect_step / col.collect_time,
           "update_ms_per_collect": update_s / collects * 1e3,
           "policy_updates_per_s": collects / dt}
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

    def one(k):
        eng.set_params(theta); eng.optim_reset()
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


def pmc_traffic():
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE, separate runs; tools/capture_profiles.sh + tools/collect_profiles.py apply
    the guide's KiB unit and gfx950 x2 FETCH correction).  Counters cannot be read inside this process,
    so the figure is the one measured when the profile was captured; None when there is none."""
    here = os.path.dirname(os.path.abspath(__file__))
    for tag in ("r05", "r04", "r03", "r02", "r01"):
        f = os.path.join(here, "profiles", f"{tag}_pmc_traffic.json")
        if os.path.exists(f):
            for name, v in json.load(open(f)).items():
                if name.startswith("void ppo_fwd_bwd_kernel<256"):
                    return float(v["hbm_bytes_per_launch"]), f"profiles/{tag}_pmc_traffic.json (rocprofv3 --pmc, separate pass)"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # test hooks (tests/test_gpu_facade.py): run the N > 1 code path on a one-GPU box -- every rank on device 0, gloo
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--share-gpu", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
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

    def one_update(k):
        # every timed update is the same workload: orthogonal-init weights, fresh Adam state
        # (re-training 20x on ONE synthetic buffer with the KL stop off would diverge to inf;
        # the restore -- a 0.8 MB H2D copy and two memsets -- stays inside the timed region)
        eng.set_params(theta)
        eng.optim_reset()
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
    per_rank, lib_comm, dt_local = None, None, dt
    if dist is not None:
        # the SURVEY 8(e) exchange: every rank's figures gathered once (RCCL all_gather of a 5-double vector), then the
        # max-over-ranks time the contract asks for
        from fsrl_amd import parallel
        per_rank = parallel.allgather_metrics({"rank": float(rank), "seed": float(seed), "updates": float(args.steps),
                                               "grad_steps": float(grad_steps * args.steps), "seconds": dt,
                                               "updates_per_s": args.steps / dt})
        tt = torch.tensor([dt], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # the same exchange through the C ABI's own RCCL communicator (fsrl_comm_init + fsrl_metrics_allreduce), checked
        # against torch.distributed's sum; outside the timed region.  Every rank first proves it can reach RCCL, so a rank
        # that cannot does not leave the others waiting inside ncclCommInitRank.
        if args.backend == "nccl" and os.environ.get("FSRL_BENCH_LIB_COMM", "1") != "0":
            try:
                eng.comm_unique_id(); ok = 1.0
            except Exception as e:                          # noqa: BLE001
                ok, lib_comm = 0.0, f"unavailable: {e}"
            flag = torch.tensor([ok], device="cuda", dtype=torch.float64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) == 1.0:
                eng.comm_init_from_torch()
                mine = np.array([1.0, float(rank), args.steps / dt_local, float(seed)])
                got = eng.metrics_allreduce(mine)
                want = torch.from_numpy(mine).cuda()
                dist.all_reduce(want, op=dist.ReduceOp.SUM)
                lib_comm = "ok" if np.array_equal(got, want.cpu().numpy()) and eng.comm_info() == (rank, world) else "mismatch"
                eng.comm_destroy()
            elif ok == 1.0:
                lib_comm = "skipped: another rank cannot reach RCCL"
    # ---- the other half of the headline metric at N > 1: every rank's whole training loop (host collector -> HIP store ->
    #      update) side by side for a few seconds on its own slice of the host cores; job env-steps/s = the sum over ranks
    e2e_job = None
    if dist is not None:
        from fsrl_amd import parallel
        cores = parallel.pin_rank_cores(local_rank if not args.share_gpu else rank, world)
        dist.barrier()
        e2e = end_to_end(local_rank, seed, seconds=4.0, device_actor=True)
        rows = parallel.allgather_metrics({"rank": float(rank), "env_steps_per_s": e2e["env_steps_per_s"],
                                           "policy_updates_per_s": e2e["policy_updates_per_s"], "cores": float(len(cores))})
        rows.sort(key=lambda r: r["rank"])
        e2e_job = {"env": e2e["env"], "actor": e2e["actor"], "envs_per_rank": e2e["envs"], "ranks": len(rows),
                   "env_steps_per_s": sum(r["env_steps_per_s"] for r in rows),
                   "policy_updates_per_s": sum(r["policy_updates_per_s"] for r in rows),
                   "per_rank_env_steps_per_s": [round(r["env_steps_per_s"], 1) for r in rows],
                   "host_cores_per_rank": int(rows[0]["cores"])}
    # ---- roofline of the dominant kernel: HIP events around every ppo_fwd_bwd_kernel launch on
    #      the library's compute stream, over K more updates of the same workload
    eng.set_profiling(True)
    k_ms, k_raw, k_n, learn_ms, proc_ms = 0.0, 0.0, 0, 0.0, 0.0
    prof_steps = max(1, min(args.steps, 5))
    for k in range(prof_steps):
        one_update(10_000 + k)
        tm = eng.last_timing()
        k_ms += tm["fwdbwd_ms"]; k_raw += tm["fwdbwd_raw_ms"]; k_n += tm["fwdbwd_launches"]
        learn_ms += tm["learn_ms"]; proc_ms += tm["process_ms"]
    eng.set_profiling(False)
    avg_launch_s = (k_ms / max(k_n, 1)) * 1e-3
    # rows per launch: 77 launches of 256 rows + 1 of 288 per pass
    rows_avg = NROWS / (grad_steps / REPEAT)
    achieved = flops_fwdbwd_launch(rows_avg) / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0

    if rank == 0:
        ups = args.steps / dt
        out = {
            "metric": "policy-updates/sec (PPO-Lagrangian update(), SafetyCarCircle-v0 shape, 256x256 MLP, 20k-step buffer)",
            "value": ups * world, "unit": "updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: PPO-Lag update, obs 8 / act 2, 256x256, N=20000 rows "
                                   "(20 envs x 1000, 250-step episodes), batch 256, repeat 4, grad-clip 0.5, "
                                   "KL early stop off", "seeds_per_gpu": 1, "parallelism": f"independent agents x{world}"},
            "grad_steps_per_update": int(grad_steps),
            "grad_steps_per_s": ups * grad_steps * world,
            "buffer_rows_per_s": ups * NROWS * world,
            "phase_ms": {"process_fn": proc_ms / prof_steps, "learn": learn_ms / prof_steps},
            "roofline": {"bound": "mfma", "kernel": "ppo_fwd_bwd_kernel<256>", "achieved": achieved,
                         "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                         "avg_launch_us": avg_launch_s * 1e6,
                         "avg_launch_us_raw_event_bracket": k_raw / max(k_n, 1) * 1e3,
                         "launches_timed": int(k_n),
                         "flops_per_launch": flops_fwdbwd_launch(rows_avg)},
        }
        if per_rank is not None:
            out["ranks_seen"] = len(per_rank)
            if lib_comm is not None:
                out["lib_metrics_allreduce"] = lib_comm
            if e2e_job is not None:
                out["end_to_end_job"] = e2e_job
            out["per_rank_updates_per_s"] = [round(r["updates_per_s"], 3) for r in sorted(per_rank, key=lambda r: r["rank"])]
            out["sum_of_rank_rates"] = sum(r["updates_per_s"] for r in per_rank)
        pmc = pmc_traffic()
        if pmc is not None:
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = pmc
        if not args.no_cpu_baseline and world == 1:       # rank 0 at N = 1 only
            out["end_to_end"] = end_to_end(local_rank, seed, device_actor=True)
            out["end_to_end_host_actor"] = end_to_end(local_rank, seed, seconds=4.0, device_actor=False)
            # the host vector-env side of the headline metric: worker processes x simulated step cost
            out["end_to_end_shmem"] = [end_to_end(local_rank, seed, seconds=3.0, device_actor=True, workers=w, busy_us=b,
                                                  envs=32) for w in (4, 32) for b in (0.0, 100.0)]
            out["multi_seed"] = multi_seed()
            out["no_clip"] = no_clip_variant(theta, inputs)
            out["grouped"] = grouped()
            out["cpu_baseline"] = cpu_baseline(theta, inputs)
            from fsrl_amd.parallel import usable_cpus
            allc = max(1, int(usable_cpus()))          # "... and with all host cores" (the cgroup quota counts)
            if allc > 4:
                out["cpu_baseline_all_cores"] = cpu_baseline(theta, inputs, seconds=6.0, threads=allc)
            out["speedup_vs_cpu_port"] = out["value"] / world / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
