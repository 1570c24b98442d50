"""world_size-2 gloo test of the only exchange step of the multi-GPU layout: the epoch metric
all-reduce / all-gather between independent per-rank agents."""
import os
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from fsrl_amd.parallel import allgather_metrics, allreduce_metrics
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = {"train/reward": 10.0 * (rank + 1), "train/cost": float(rank), "n/st": 6000.0}
    mean = allreduce_metrics(m)
    total = allreduce_metrics({"n/st": m["n/st"]}, average=False)
    per_rank = allgather_metrics(m)
    q.put((rank, mean, total, per_rank))
    dist.destroy_process_group()


def test_metric_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, mean, total, per_rank in outs:
        assert mean == {"n/st": 6000.0, "train/cost": 0.5, "train/reward": 15.0}
        assert total == {"n/st": 12000.0}
        assert [d["train/reward"] for d in per_rank] == [10.0, 20.0]


def test_allreduce_is_identity_without_process_group():
    sys.path.insert(0, ROOT)
    from fsrl_amd.parallel import allreduce_metrics
    assert allreduce_metrics({"a": 1.5}) == {"a": 1.5}
