"""Grouped PPO-Lagrangian update: k independent policies of one network shape (multi-seed runs) stepped in lock step on one
MI355X -- `fsrl_group_ppo_update`, every launch of the minibatch step carrying all members ("optional within-GPU
batching of k seeds").  The reference runs seeds as separate jobs; per member this is PPOLagrangian.update
(fsrl/policy/ppo_lag.py:214-257 after base_policy.py:332-355): same arguments, same logger keys, same lr_scheduler step.

    group = PolicyGroup([agent.policy for agent in agents])
    ... every agent collects into ITS buffer, steps ITS PID multiplier (trainer.policy_update_fn does pre_update_fn) ...
    group.update(buffers, batch_size=256, repeat=4)

Random streams: the library shuffles (member i seeded from `seed`), so k grouped seeds are k independent runs but not the
reference's numpy permutation stream; pass `perms` for that."""
from typing import List, Optional, Sequence

import numpy as np

from fsrl_amd.engine import EngineGroup
from fsrl_amd.policy.ppo_lag import PPO_STAT_KEYS, PPOLagrangian


class PolicyGroup:
    def __init__(self, policies: Sequence[PPOLagrangian], seed: int = 1):
        self.policies = list(policies)
        assert all(isinstance(p, PPOLagrangian) for p in self.policies), "grouped updates: PPOLagrangian policies"
        # PPOLagrangian(reference_rng=True) burns the torch draws the reference wastes inside update(); a grouped update has no
        # single-policy random stream to follow
        assert not any(getattr(p, "_reference_rng", False) for p in self.policies), "reference_rng policies cannot be grouped"
        self.group = EngineGroup([p.engine for p in self.policies])
        self._seed, self._calls = int(seed), 0

    def close(self):
        self.group.close()

    def update(self, buffers, batch_size: int = 256, repeat: int = 4, perms: Optional[List] = None):
        pols = self.policies
        for p, b in zip(pols, buffers):
            assert getattr(b, "engine", None) is p.engine, "buffer i must be the HipVectorReplayBuffer of policy i"
            p.updating = True
        try:
            lr = [p.lagrangians_and_rescaling() if p.use_lagrangian else ([0.0] * (p.critics_num - 1), 1.0) for p in pols]
            lags = np.array([x[0] if len(x[0]) else [0.0] for x in lr], np.float64)
            resc = [x[1] for x in lr]
            self._calls += 1
            stats, stopped = self.group.ppo_update(lags, resc, batch_size, repeat, perms=perms,
                                                   seed=0 if perms is not None else self._seed + 7919 * self._calls)
        except BaseException:
            # a failed group update may have stepped some parameters already: the host mirrors are stale, nobody is updating
            for p in pols:
                p.updating = False
                p._mark_stale()
            raise
        out = []
        for p, st, sp in zip(pols, stats, stopped):
            drop = set()
            if not p.use_lagrangian or p.critics_num < 2:
                drop |= {"loss/lagrangian", "loss/actor_safety"}
            if p.critics_num < 2:
                drop.add("loss/vf1")
            cols = [j for j, k in enumerate(PPO_STAT_KEYS) if k not in drop]
            keys = [PPO_STAT_KEYS[j] for j in cols]
            table = getattr(p.logger, "store_rows", None)
            if table is not None:
                table(keys, st[:, cols])
            else:
                for row in st:
                    p.logger.store(**{k: float(row[j]) for j, k in zip(cols, keys)})
            if sp >= 0:
                p.logger.print("Early stop at step %d due to reaching max kl." % sp)
            p.gradient_steps += len(st)
            p.logger.store(gradient_steps=p.gradient_steps, tab="update")
            p._mark_stale()
            p._step_lr_scheduler()
            p.updating = False
            out.append({"gradient_steps": len(st), "early_stop_pass": sp})
        return out
