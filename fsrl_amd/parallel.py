"""The one exchange step of the multi-GPU layout: independent agents, one per GPU,
average / sum their epoch metric vector.  RCCL (`backend="nccl"`) on GPUs, gloo on CPU."""
from typing import Dict

import numpy as np


def is_distributed() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def allreduce_metrics(metrics: Dict[str, float], device=None, average: bool = True) -> Dict[str, float]:
    """Element-wise mean (or sum) of a flat scalar dict over all ranks; identity if the process
    group is not initialised.  Keys must match across ranks (sorted for a fixed layout)."""
    if not is_distributed():
        return dict(metrics)
    import torch
    import torch.distributed as dist
    keys = sorted(metrics)
    vec = torch.tensor([float(metrics[k]) for k in keys], dtype=torch.float64,
                       device=device if device is not None else "cpu")
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
                       device=device if device is not None else "cpu")
    out = [torch.zeros_like(vec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, vec)
    return [{k: float(v) for k, v in zip(keys, o.tolist())} for o in out]
