"""The multi-GPU layout: independent agents, one process per GPU, one seed each; the ONLY exchange is the
fixed epoch metric vector (sums -> means / totals on every rank).  RCCL (`backend="nccl"`) on GPUs, gloo on CPU.

The reference has no multi-GPU code; what shards is what its users run as separate jobs (one seed per job).  Here a
`torchrun --nproc-per-node N` launch (examples/train_multi_gpu.py, bench.py --gpus N) gives rank r: seed base + r,
device LOCAL_RANK, its own host envs on its own core slice, and `BaseTrainer._close_epoch` calls `reduce_epoch` once per
epoch.  Message: EPOCH_KEYS float64 values (< 256 B): latency-bound, no tuning."""
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

# The epoch vector: every entry is a per-rank SUM; job-level means are quotients of the reduced sums, so a
# rank that collected more episodes weighs more in the mean reward -- exactly what pooling the episodes would give.
EPOCH_KEYS = ("ranks", "n_st", "n_ep", "sum_rew", "sum_cost", "sum_len", "n_updates", "n_grad_steps", "sum_loss_total",
              "sum_kl", "collect_time", "update_time", "duration", "test_n_ep", "test_sum_rew", "test_sum_cost", "stop")


def is_distributed() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def rank_world() -> Tuple[int, int]:
    if not is_distributed():
        return 0, 1
    import torch.distributed as dist
    return dist.get_rank(), dist.get_world_size()


def init_from_env(backend: Optional[str] = None, share_gpu: bool = False) -> Tuple[int, int, int]:
    """Join the job `torchrun` (or the driver's `python -m torch.distributed.run`) started: RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* from the environment -> (rank, local_rank, world).  No-op (0, 0, 1) for a plain `python` launch.
    backend None: "nccl" (= RCCL over xGMI) when a GPU is visible, else "gloo".  share_gpu: every rank uses device 0
    (tests on a one-GPU box; RCCL refuses two ranks on one device, so pair it with gloo)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return 0, local_rank, 1
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL needs it)
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def usable_cpus() -> float:
    """CPUs this process can actually run on: the affinity mask, capped by the cgroup CPU quota (a container may see 256 CPUs
    and be allowed 16 CPU-seconds per second: cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`)."""
    try:
        n = float(len(os.sched_getaffinity(0)))
    except AttributeError:
        n = float(os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, float(quota) / float(period))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = float(f.read())
            if quota > 0:
                n = min(n, quota / period)
        except (OSError, ValueError):
            pass
    return max(n, 1.0)


def pin_rank_cores(local_rank: int, local_world: int, reserve: int = 0) -> List[int]:
    """Give this rank a disjoint slice of the host cores (its collector loop + env workers inherit it): SURVEY 8e
    "envs_per_rank host workers pinned to disjoint core sets".  Returns the slice; a no-op where affinity is unsupported."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return []
    per = max(1, (len(cores) - reserve) // max(local_world, 1))
    mine = cores[local_rank * per:(local_rank + 1) * per] or cores
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return []
    return mine


def _reduce_device():
    """Tensor placement for the collective: the current CUDA device under RCCL, the host under gloo."""
    import torch
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def allreduce_metrics(metrics: Dict[str, float], device=None, average: bool = True) -> Dict[str, float]:
    """Element-wise mean (or sum) of a flat scalar dict over all ranks; identity if the process
    group is not initialised.  Keys must match across ranks (sorted for a fixed layout)."""
    if not is_distributed():
        return dict(metrics)
    import torch
    import torch.distributed as dist
    keys = sorted(metrics)
    vec = torch.tensor([float(metrics[k]) for k in keys], dtype=torch.float64,
                       device=device if device is not None else _reduce_device())
    dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    if average:
        vec /= dist.get_world_size()
    return {k: float(v) for k, v in zip(keys, vec.tolist())}


def allgather_metrics(metrics: Dict[str, float], device=None):
    """Per-rank copies of the metric dict (keeps per-seed curves)."""
    if not is_distributed():
        return [dict(metrics)]
    import torch
    import torch.distributed as dist
    keys = sorted(metrics)
    vec = torch.tensor([float(metrics[k]) for k in keys], dtype=torch.float64,
                       device=device if device is not None else _reduce_device())
    out = [torch.zeros_like(vec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, vec)
    return [{k: float(v) for k, v in zip(keys, o.tolist())} for o in out]


def reduce_epoch(local: Dict[str, float], engine=None) -> Dict[str, float]:
    """ONE all-reduce(sum) of the fixed EPOCH_KEYS vector (missing keys count as 0) and the job-level figures derived
    from it on every rank:

        job/ranks, job/env_step, job/episodes                totals
        job/reward, job/cost, job/length                     episode-weighted means over all ranks' train episodes
        job/test_reward, job/test_cost                       likewise over the test episodes (if any)
        job/loss_total, job/kl                               means over all ranks' optimiser steps
        job/env_steps_per_s, job/updates_per_s               whole-job throughput: totals / max-free mean duration
        job/all_stop                                         1.0 when every rank's stop rule fired

    Identity layout without a process group (ranks = 1), so single-GPU runs log the same keys.  `engine`: when its context
    has joined the library's RCCL communicator (Engine.comm_init / comm_init_from_torch), the sum goes through the C ABI
    (`fsrl_metrics_allreduce`: ncclAllReduce on the engine's compute stream) instead of torch.distributed."""
    vec = np.array([float(local.get(k, 0.0)) for k in EPOCH_KEYS], np.float64)
    vec[0] = 1.0
    if engine is not None and engine.comm_info()[1] > 1:
        vec = engine.metrics_allreduce(vec)          # the library's own RCCL communicator (fsrl_metrics_allreduce)
    elif is_distributed():
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(vec).to(_reduce_device())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        vec = t.cpu().numpy()
    s = dict(zip(EPOCH_KEYS, vec.tolist()))
    ranks = max(s["ranks"], 1.0)
    dur = s["duration"] / ranks                        # ranks run side by side: wall time = the mean (≈ each) duration
    out = {"job/ranks": s["ranks"], "job/env_step": s["n_st"], "job/episodes": s["n_ep"],
           "job/reward": s["sum_rew"] / max(s["n_ep"], 1.0), "job/cost": s["sum_cost"] / max(s["n_ep"], 1.0),
           "job/length": s["sum_len"] / max(s["n_ep"], 1.0),
           "job/loss_total": s["sum_loss_total"] / max(s["n_grad_steps"], 1.0),
           "job/kl": s["sum_kl"] / max(s["n_grad_steps"], 1.0),
           "job/env_steps_per_s": s["n_st"] / max(dur, 1e-9), "job/updates_per_s": s["n_updates"] / max(dur, 1e-9),
           "job/collect_time": s["collect_time"] / ranks, "job/update_time": s["update_time"] / ranks,
           "job/all_stop": 1.0 if s["stop"] >= ranks else 0.0}
    if s["test_n_ep"] > 0:
        out["job/test_reward"] = s["test_sum_rew"] / s["test_n_ep"]
        out["job/test_cost"] = s["test_sum_cost"] / s["test_n_ep"]
    return out
