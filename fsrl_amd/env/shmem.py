"""Build-owned multi-process vector env with shared-memory observations: the host side of the headline metric.

north_star: "vectorized rollout collection (FastCollector over ShmemVectorEnv) stays on the host CPUs"; the reference builds
`ShmemVectorEnv([lambda: gym.make(task) for _ in range(training_num)])` (examples/mlp/train_ppol_agent.py:120-123), i.e.
tianshou's worker processes with the observations in shared memory.  tianshou, gymnasium and the simulators are absent
here, so this is the same ARCHITECTURE around the synthetic dynamics of `SyntheticSafetyVectorEnv`:

  * `workers` processes, each owning a contiguous slice of the `env_num` envs (one env per worker when
    workers == env_num, like tianshou; fewer workers batch their slice's steps);
  * obs / act / rew / cost / flags live in ONE `multiprocessing.shared_memory` block: the parent writes the actions of the
    active envs and bumps each touched worker's `go` sequence number in that block, the workers step their envs in parallel
    (burning `busy_us` per env step to stand in for a physics step), write results in place and bump their `done` number;
    the parent polls those -- no pickling, no pipes and no system call on the per-step path.  A worker that sees nothing for
    `spin_us` (the policy update, ~10 ms) parks on a semaphore and is woken by the next command, so idle workers do not burn
    their cores (32 semaphore posts + waits per vector step cost ~130 us; the sequence numbers ~2 us);
  * `cores`: the rank's core slice (fsrl_amd.parallel.pin_rank_cores); worker w is pinned to cores[w % len(cores)].

Calling convention of the collector: `len(env)`, `reset(ids=None) -> (obs, info)`, `step(act, ids) -> (obs, rew, terminated,
truncated, {"cost": cost})`, `close()`.  With workers == 1 the trajectories are bit-identical to the in-process env of the
same seed (tests/test_shmem_env.py)."""
import multiprocessing as mp
import os
from multiprocessing import shared_memory
from types import SimpleNamespace

import numpy as np

from fsrl_amd.env.synthetic import Box, SyntheticSafetyVectorEnv

_CMD_STEP, _CMD_RESET, _CMD_EXIT = 1, 2, 3


def _layout(env_num, obs_dim, act_dim):
    """name -> (offset, shape, dtype) of the arrays inside the shared block, 64-byte aligned"""
    fields = [("obs", (env_num, obs_dim), np.float32), ("act", (env_num, act_dim), np.float32), ("rew", (env_num, ), np.float64),
              ("cost", (env_num, ), np.float64), ("term", (env_num, ), np.uint8), ("trunc", (env_num, ), np.uint8),
              ("active", (env_num, ), np.uint8), ("cmd", (64, ), np.int32),
              # per-worker handshake words, one 64-byte line each (no false sharing between the pollers)
              ("go", (env_num, 8), np.int64), ("done", (env_num, 8), np.int64), ("parked", (env_num, 8), np.int64)]
    out, off = {}, 0
    for name, shape, dt in fields:
        out[name] = (off, shape, dt)
        off = (off + int(np.prod(shape)) * np.dtype(dt).itemsize + 63) // 64 * 64
    return out, off


def _views(buf, layout):
    return {name: np.ndarray(shape, dtype=dt, buffer=buf, offset=off) for name, (off, shape, dt) in layout.items()}


def _worker(w, lo, hi, shm_name, env_num, obs_dim, act_dim, episode_len, seed, busy_us, wake, done_sem, core, spin_us):
    if core is not None:
        try:
            os.sched_setaffinity(0, {core})
        except (AttributeError, OSError):
            pass
    shm = shared_memory.SharedMemory(name=shm_name)
    layout, _ = _layout(env_num, obs_dim, act_dim)
    v = _views(shm.buf, layout)
    env = SyntheticSafetyVectorEnv(env_num=hi - lo, obs_dim=obs_dim, act_dim=act_dim, episode_len=episode_len, seed=seed,
                                   busy_us=busy_us)
    go, done, parked = v["go"][w], v["done"][w], v["parked"][w]
    seen = 0
    spins = max(1, int(spin_us / 0.15))          # ~0.15 us per poll of a shared word from Python
    try:
        while True:
            n = 0
            while go[0] == seen:                 # poll; park after spin_us of nothing (spin_us = 0: semaphores only)
                n += 1
                if n >= spins or done_sem is not None:
                    parked[0] = 1
                    while go[0] == seen:         # a post can race the flag: the timeout bounds a missed wake-up to 2 ms
                        if done_sem is not None:
                            wake.acquire()       # semaphore mode: the parent posts for every command, nothing to miss
                        else:
                            wake.acquire(timeout=0.002)
                    parked[0] = 0
                    break
            seen = int(go[0])
            cmd = int(v["cmd"][0])
            if cmd == _CMD_EXIT:
                break
            local = np.flatnonzero(v["active"][lo:hi])
            if local.size:
                if cmd == _CMD_RESET:
                    obs, _ = env.reset(local)
                    v["obs"][lo + local] = obs
                else:
                    obs, rew, term, trunc, info = env.step(v["act"][lo + local], local)
                    g = lo + local
                    v["obs"][g] = obs; v["rew"][g] = rew; v["cost"][g] = info["cost"]
                    v["term"][g] = term; v["trunc"][g] = trunc
            done[0] = seen
            if done_sem is not None:
                done_sem.release()
    finally:
        del v
        shm.close()


class ShmemVectorEnv:
    def __init__(self, env_num=32, workers=None, obs_dim=8, act_dim=2, episode_len=300, seed=0, busy_us=0.0, cores=None,
                 start_method="spawn", spin_us=None):
        workers = env_num if workers is None else int(workers)
        assert 1 <= workers <= env_num
        self.env_num, self.obs_dim, self.act_dim, self.workers = env_num, obs_dim, act_dim, workers
        self.episode_len, self.busy_us = episode_len, busy_us
        self.observation_space = Box(-np.inf, np.inf, (obs_dim, ))
        self.action_space = Box(-1.0, 1.0, (act_dim, ))
        self.spec = SimpleNamespace(id="SyntheticSafety-v0", max_episode_steps=episode_len)
        self._layout, size = _layout(env_num, obs_dim, act_dim)
        self._shm = shared_memory.SharedMemory(create=True, size=size)
        self._v = _views(self._shm.buf, self._layout)
        for a in self._v.values():
            a[...] = 0
        ctx = mp.get_context(start_method)
        # contiguous slices: worker w owns envs [bounds[w], bounds[w + 1])
        self._bounds = [round(w * env_num / workers) for w in range(workers + 1)]
        self._owner = np.zeros(env_num, np.int32)
        # polling needs a CPU per worker plus one for the collector; on a smaller or quota-limited host (measured on the MI355X
        # box: 256 CPUs visible, cgroup quota 16: 32 pollers get throttled, 84k vs 110k env-steps/s at 100 us per step) the
        # workers park on semaphores right away and the parent blocks on theirs -- the classic handshake
        if spin_us is None:
            from fsrl_amd.parallel import usable_cpus
            usable = min(usable_cpus(), len(cores)) if cores else usable_cpus()      # affinity AND the cgroup CPU quota
            spin_us = 500.0 if workers + 1 <= usable else 0.0
        self.spin_us = float(spin_us)
        self._wake = [ctx.Semaphore(0) for _ in range(workers)]
        self._done = [ctx.Semaphore(0) for _ in range(workers)] if self.spin_us <= 0 else None
        self._seq = 0
        self._procs = []
        for w in range(workers):
            lo, hi = self._bounds[w], self._bounds[w + 1]
            self._owner[lo:hi] = w
            wseed = seed if workers == 1 else seed * 7919 + w
            core = None if not cores else list(cores)[w % len(cores)]
            p = ctx.Process(target=_worker, args=(w, lo, hi, self._shm.name, env_num, obs_dim, act_dim, episode_len, wseed,
                                                  busy_us, self._wake[w], self._done[w] if self._done else None, core,
                                                  self.spin_us), daemon=True)
            p.start()
            self._procs.append(p)
        self._closed = False

    def __len__(self):
        return self.env_num

    def _run(self, cmd, ids):
        v = self._v
        v["active"][:] = 0
        v["active"][ids] = 1
        v["cmd"][0] = cmd
        touched = np.unique(self._owner[ids])
        self._seq += 1
        seq = self._seq
        v["go"][touched, 0] = seq                       # the command is in place: release the pollers
        for w in (touched if self._done is not None else touched[v["parked"][touched, 0] != 0]):
            self._wake[w].release()                      # parked workers (first step after an update) need the post
        if self._done is not None:                       # semaphore mode: every touched worker parked and posts when done
            for w in touched:
                if not self._done[w].acquire(timeout=60):
                    raise RuntimeError(f"env worker {w} did not answer (exitcode {self._procs[w].exitcode})")
            return
        done = v["done"]
        n = 0
        while not (done[touched, 0] == seq).all():
            n += 1
            if n % 4096 == 0:                            # every ~ms: liveness of the workers, and an overall deadline
                dead = [int(w) for w in touched if not self._procs[w].is_alive()]
                if dead or n > 4096 * 60000:
                    raise RuntimeError(f"env workers {dead or list(map(int, touched))} did not answer")
                for w in touched[v["parked"][touched, 0] != 0]:
                    self._wake[w].release()

    def reset(self, ids=None, **kwargs):
        ids = np.arange(self.env_num) if ids is None else np.asarray(ids)
        self._run(_CMD_RESET, ids)
        return self._v["obs"][ids].copy(), {}

    def step(self, act, ids=None):
        ids = np.arange(self.env_num) if ids is None else np.asarray(ids)
        self._v["act"][ids] = np.asarray(act, np.float32).reshape(len(ids), self.act_dim)
        self._run(_CMD_STEP, ids)
        v = self._v
        return (v["obs"][ids].copy(), v["rew"][ids].copy(), v["term"][ids].astype(bool), v["trunc"][ids].astype(bool),
                {"cost": v["cost"][ids].copy()})

    def close(self):
        if self._closed:
            return
        self._closed = True
        self._v["cmd"][0] = _CMD_EXIT
        self._seq += 1
        self._v["go"][:self.workers, 0] = self._seq
        for g in self._wake:
            g.release()
        for p in self._procs:
            p.join(5)
            if p.is_alive():
                p.terminate()
        self._v = None
        self._shm.close()
        try:
            self._shm.unlink()
        except FileNotFoundError:
            pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
