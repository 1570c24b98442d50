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


class _StubCollector:
    """collect() with rank-dependent, known statistics; no env, no engine"""

    def __init__(self, rank):
        self.rank, self.buffer = rank, None
        self.collect_step, self.collect_episode, self.collect_time = 0, 0, 0.0

    def reset_stat(self):
        self.collect_step, self.collect_episode, self.collect_time = 0, 0, 0.0

    def reset_buffer(self, keep_statistics=False):
        pass

    def collect(self, n_episode):
        n_ep = n_episode + self.rank                      # ranks collect different episode counts
        n_st = 10 * n_ep
        self.collect_step += n_st; self.collect_episode += n_ep; self.collect_time += 1e-3
        return {"n/ep": n_ep, "n/st": n_st, "rew": 10.0 * (self.rank + 1), "len": 10.0, "total_cost": 2.0 * n_ep * (self.rank + 1),
                "cost": 2.0 * (self.rank + 1), "truncated": 1.0, "terminated": 0.0}


class _StubPolicy:
    def __init__(self, logger):
        self.gradient_steps, self.logger = 0, logger

    def train(self): pass
    def eval(self): pass
    def pre_update_fn(self, **kw): pass
    def post_update_fn(self, **kw): pass

    def update(self, sample_size, buffer, batch_size=64, repeat=1):
        self.gradient_steps += 3
        self.logger.store(**{"loss/total": 1.0 + self.gradient_steps, "loss/kl": 0.01})


def _trainer_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from fsrl_amd import parallel
    r, lr, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and parallel.rank_world() == (rank, world)
    from fsrl_amd.trainer import OnpolicyTrainer
    from fsrl_amd.utils.logger import BaseLogger
    logger = BaseLogger(None)
    # rank 1's stop rule fires after the first epoch, rank 0's never: the job must keep BOTH ranks in lock step
    stop_fn = (lambda rew, cost: True) if rank == 1 else (lambda rew, cost: False)
    tr = OnpolicyTrainer(_StubPolicy(logger), _StubCollector(rank), None, max_epoch=3, batch_size=64, step_per_epoch=40,
                         repeat_per_collect=1, episode_per_collect=2, stop_fn=stop_fn, logger=logger, verbose=False)
    rows = []
    for ep, stat, info in tr:
        rows.append((ep, dict(tr.job_stats), {k: v for k, v in stat.items() if k.startswith("job/")}))
    q.put((rank, rows))
    dist.destroy_process_group()


def test_trainer_epoch_allreduce_world2_gloo():
    """Two independent trainers (stub collector / policy: no GPU here) under one process group: after every epoch each
    rank holds the SAME job-level figures, equal to the episode-weighted pooling of the two ranks' statistics; a stop
    rule that fires on one rank only does not desynchronise the collective (the job stops when all ranks' rules fired)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len(outs[0]) == len(outs[1]) == 3                       # rank 1's early stop_fn did not end its loop alone
    for e in range(3):
        j0, j1 = outs[0][e][1], outs[1][e][1]
        for k in ("job/ranks", "job/env_step", "job/episodes", "job/reward", "job/cost", "job/length", "job/loss_total", "job/kl"):
            assert j0[k] == j1[k], k                                   # identical on every rank
        # per epoch: rank 0 collects 2 episodes x 2 collects (40 steps), rank 1 collects 3 x 2 (60 steps)
        assert j0["job/ranks"] == 2.0 and j0["job/episodes"] == 4 + 6 and j0["job/env_step"] == 40 + 60
        assert abs(j0["job/reward"] - (4 * 10.0 + 6 * 20.0) / 10) < 1e-12       # pooled over episodes, not mean of means
        assert abs(j0["job/cost"] - (4 * 2.0 + 6 * 4.0) / 10) < 1e-12
        assert j0["job/length"] == 10.0 and abs(j0["job/kl"] - 0.01) < 1e-12 and j0["job/all_stop"] == 0.0
    assert outs[0][0][2].get("job/reward") == outs[0][0][1]["job/reward"]        # rank 0 logs the job row ...
    assert not outs[1][0][2]                                                     # ... the other ranks do not
