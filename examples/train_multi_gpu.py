#!/usr/bin/env python
"""BASELINE.json configs[4]: independent seeds sharded across the GPUs of one node -- one process, one seed, one agent and
one HIP context per GPU, `--envs` host envs per rank on the rank's own core slice; the ONLY exchange is the epoch metric
vector all-reduced over RCCL (fsrl_amd.parallel.reduce_epoch, called from BaseTrainer._close_epoch).

    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_multi_gpu.py \
        --algo ppol --envs 32 --epoch 5 --hidden 256

The reference has no multi-GPU path (its users start one job per seed); the partitioning follows SURVEY.md 8(e):
rank r -> seed base + r, device LOCAL_RANK, `envs` workers pinned to a disjoint core set.  Rank 0 prints one job-level row
per epoch (pooled reward / cost over all ranks' episodes, whole-job env-steps/s and updates/s) and, at the end, the
per-seed table gathered from all ranks.  `--backend gloo --share-gpu` runs every rank on device 0 (tests on a one-GPU box).
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fsrl_amd import parallel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="ppol", choices=["ppol", "cpo", "trpol", "focops", "sacl", "ddpgl", "cvpo"])
    ap.add_argument("--seed", type=int, default=0, help="base seed; rank r trains seed + r")
    ap.add_argument("--envs", type=int, default=32)
    ap.add_argument("--epoch", type=int, default=3)
    ap.add_argument("--step-per-epoch", type=int, default=9600)
    ap.add_argument("--episode-len", type=int, default=300)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--cost-limit", type=float, default=10.0)
    ap.add_argument("--workers", type=int, default=0, help="> 0: step the envs in that many worker processes (shared memory)")
    ap.add_argument("--busy-us", type=float, default=0.0, help="host time burnt per env step (a simulator's cost)")
    ap.add_argument("--backend", default=None, choices=[None, "nccl", "gloo"])
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--metrics-via", default="torch", choices=["torch", "lib"],
                    help="lib: the epoch vector goes through the C ABI's own RCCL communicator (fsrl_metrics_allreduce); "
                         "torch.distributed then only hands the 128-byte communicator id around")
    ap.add_argument("--logdir", default=None)
    ap.add_argument("--json", action="store_true", help="rank 0: print the final job summary as one JSON line")
    a = ap.parse_args()

    rank, local_rank, world = parallel.init_from_env(a.backend, share_gpu=a.share_gpu)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    cores = parallel.pin_rank_cores(int(os.environ.get("LOCAL_RANK", "0")), local_world)

    from fsrl_amd.agent import CPOAgent, CVPOAgent, DDPGLagAgent, FOCOPSAgent, PPOLagAgent, SACLagAgent, TRPOLagAgent
    from fsrl_amd.env import ShmemVectorEnv, SyntheticSafetyVectorEnv
    from fsrl_amd.utils import BaseLogger
    agents = {"ppol": PPOLagAgent, "cpo": CPOAgent, "trpol": TRPOLagAgent, "focops": FOCOPSAgent, "sacl": SACLagAgent,
              "ddpgl": DDPGLagAgent, "cvpo": CVPOAgent}
    seed = a.seed + rank
    if a.workers > 0:
        env = ShmemVectorEnv(env_num=a.envs, workers=a.workers, obs_dim=8, act_dim=2, episode_len=a.episode_len, seed=seed,
                             busy_us=a.busy_us, cores=cores)
    else:
        env = SyntheticSafetyVectorEnv(env_num=a.envs, obs_dim=8, act_dim=2, episode_len=a.episode_len, seed=seed,
                                       busy_us=a.busy_us)
    logdir = a.logdir or tempfile.mkdtemp(prefix="fsrl_amd_job_")
    logger = BaseLogger(logdir, name=f"{a.algo}-seed{seed}")            # every rank keeps its own seed's curve
    agent = agents[a.algo](env, logger, cost_limit=a.cost_limit, device=f"cuda:{local_rank}", seed=seed,
                           hidden_sizes=(a.hidden, a.hidden), training_num=a.envs)
    if a.metrics_via == "lib" and world > 1:
        agent.policy.engine.comm_init_from_torch()
    kw = dict(epoch=a.epoch, episode_per_collect=a.envs, step_per_epoch=a.step_per_epoch, device_actor=True,
              verbose=False, save_ckpt=False, show_progress=False)
    if a.algo in ("sacl", "ddpgl", "cvpo"):
        kw.update(update_per_step=0.2, batch_size=256)
    else:
        kw.update(repeat_per_collect=4, batch_size=256 if a.algo in ("ppol", "focops") else 99999)
    t0 = time.time()
    ep, stat, info = agent.learn(env, None, **kw)
    wall = time.time() - t0
    # ---- per-seed table: one all_gather at the end (per-seed curves stay in each rank's progress.txt)
    mine = {"seed": float(seed), "reward": float(stat.get("train/reward", 0.0)), "cost": float(stat.get("train/cost", 0.0)),
            "env_step": float(stat.get("update/env_step", 0.0)), "wall_s": wall}
    table = parallel.allgather_metrics(mine)
    if rank == 0:
        job = {k: v for k, v in info.items() if k.startswith("job/")}
        for row in sorted(table, key=lambda r: r["seed"]):
            print("seed %d: reward %.2f cost %.2f env_step %d (%.1f s)" % (row["seed"], row["reward"], row["cost"],
                                                                           row["env_step"], row["wall_s"]))
        if job:
            print("job: %d ranks, pooled reward %.2f cost %.2f, %.0f env-steps/s, %.1f updates/s (last epoch)" % (
                job["job/ranks"], job["job/reward"], job["job/cost"], job["job/env_steps_per_s"], job["job/updates_per_s"]))
        if a.json:
            print(json.dumps({"ranks": world, "algo": a.algo, "envs_per_rank": a.envs, "epochs": ep, "per_seed": table,
                              "job_last_epoch": job, "cores_rank0": len(cores)}))
    if hasattr(env, "close"):
        env.close()
    agent.policy.engine.close()
    if parallel.is_distributed():
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
