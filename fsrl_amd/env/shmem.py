"""Build-owned multi-process vector env with shared-memory observations: the host side of the headline metric.

north_star: "vectorized rollout collection (FastCollector over ShmemVectorEnv) stays on the host CPUs"; the reference builds
`ShmemVectorEnv([lambda: gym.make(task) for _ in range(training_num)])` (examples/mlp/train_ppol_agent.py:120-123), i.e.
tianshou's worker processes with the observations in shared memory.  tianshou, gymnasium and the simulators are absent
here, so this is the same ARCHITECTURE around either the synthetic dynamics of `SyntheticSafetyVectorEnv` (the bench) or the
caller's own env factories (`ShmemVectorEnv([lambda: make(task) for _ in range(n)])`, duck-typed gym envs, fsrl_amd/env/venv.py):

  * `workers` processes, each owning a contiguous slice of the `env_num` envs (one env per worker when
    workers == env_num, like tianshou; fewer workers batch their slice's steps);
  * obs / act / rew / cost / flags live in ONE `multiprocessing.shared_memory` block: the parent writes the actions of the
    active envs, the workers step their envs in parallel (an env step costs `busy_us` of host time, the stand-in for a
    physics step) and write the results in place -- no pickling, no pipes;
  * the per-step handshake is NATIVE (fsrl_amd/env/csrc/fsrl_env.c, libfsrl_env.so): a command is one release store of a
    generation word + one FUTEX_WAKE for every worker sleeping on it; completion is one counter the workers decrement, the
    collector sleeps on it once and the last worker wakes it.  Workers block inside the C call, not in the interpreter.
    (Round 2: a semaphore pair per worker = 32 posts + 32 blocking waits from Python per vector step, ~130 us.)
  * two LANES: the workers are split in two halves with their own generation / completion words, so the collector can
    keep one half stepping while it computes the other half's actions (`step_async` / `step_wait`; FastCollector's
    split-phase loop).  `step()` / `reset()` drive both lanes at once;
  * `cores`: the rank's core slice (fsrl_amd.parallel.pin_rank_cores); worker w is pinned to cores[w % len(cores)].

Calling convention of the collector: `len(env)`, `reset(ids=None) -> (obs, info)`, `step(act, ids) -> (obs, rew, terminated,
truncated, {"cost": cost})`, `close()`.  With workers == 1 the trajectories are bit-identical to the in-process env of the
same seed (tests/test_shmem_env.py)."""
import ctypes as C
import multiprocessing as mp
import os
from multiprocessing import shared_memory
from types import SimpleNamespace

import numpy as np

from fsrl_amd.env.synthetic import Box, SyntheticSafetyVectorEnv

_CMD_STEP, _CMD_RESET, _CMD_EXIT = 1, 2, 3
_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libfsrl_env.so")


def _load_lib():
    """libfsrl_env.so: the futex handshake.  There is no Python fallback: without the library the env does not start."""
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} not found: build it with fsrl_amd/env/csrc/build.sh (or __graft_entry__.build())")
    lib = C.CDLL(_LIB_PATH)
    u32p = C.c_void_p
    lib.fsrl_env_post.argtypes = [u32p, u32p, C.c_uint32, C.c_uint32]; lib.fsrl_env_post.restype = None
    lib.fsrl_env_wait_go.argtypes = [u32p, C.c_uint32, C.c_uint32, C.c_int32]; lib.fsrl_env_wait_go.restype = C.c_uint32
    lib.fsrl_env_done.argtypes = [u32p]; lib.fsrl_env_done.restype = None
    lib.fsrl_env_wait_done.argtypes = [u32p, C.c_uint32, C.c_int32]; lib.fsrl_env_wait_done.restype = C.c_int32
    return lib


def _layout(env_num, obs_dim, act_dim, workers):
    """name -> (offset, shape, dtype) of the arrays inside the shared block, 64-byte aligned"""
    fields = [("obs", (env_num, obs_dim), np.float32), ("act", (env_num, act_dim), np.float32), ("rew", (env_num, ), np.float64),
              ("cost", (env_num, ), np.float64), ("term", (env_num, ), np.uint8), ("trunc", (env_num, ), np.uint8),
              ("active", (env_num, ), np.uint8),
              # handshake words, one 64-byte line each: per lane (generation | pending workers | command), per worker the
              # generation of the last command it takes part in
              ("hs", (2, 3, 16), np.uint32), ("want", (workers, 16), np.uint32),
              # per worker: non-zero once it raised inside its env (the collector checks it after every wait)
              ("err", (workers, 16), np.uint32)]
    out, off = {}, 0
    for name, shape, dt in fields:
        out[name] = (off, shape, dt)
        off = (off + int(np.prod(shape)) * np.dtype(dt).itemsize + 63) // 64 * 64
    return out, off


def _views(buf, layout):
    return {name: np.ndarray(shape, dtype=dt, buffer=buf, offset=off) for name, (off, shape, dt) in layout.items()}


def _make_env(spec, lo, hi):
    """the worker's own envs: the synthetic dynamics, or the user's factories for envs [lo, hi)"""
    if spec["kind"] == "synthetic":
        return SyntheticSafetyVectorEnv(env_num=hi - lo, obs_dim=spec["obs_dim"], act_dim=spec["act_dim"], episode_len=spec["episode_len"],
                                        seed=spec["seed"], busy_us=spec["busy_us"])
    import cloudpickle
    from fsrl_amd.env.venv import EnvList
    return EnvList([cloudpickle.loads(b)() for b in spec["fns"]], seed=spec["seed"], first_index=lo)


def _worker(w, lane, lo, hi, shm_name, env_num, workers, obs_dim, act_dim, spec, core, spin):
    if core is not None:
        try:
            os.sched_setaffinity(0, {core})
        except (AttributeError, OSError):
            pass
    lib = _load_lib()
    shm = shared_memory.SharedMemory(name=shm_name)
    layout, _ = _layout(env_num, obs_dim, act_dim, workers)
    v = _views(shm.buf, layout)
    env = _make_env(spec, lo, hi)
    gen_p, pend_p = v["hs"][lane, 0].ctypes.data, v["hs"][lane, 1].ctypes.data
    cmd_w, want_w = v["hs"][lane, 2], v["want"][w]
    parent = os.getppid()
    sl, all_local = slice(lo, hi), np.arange(hi - lo)
    v_active, v_obs, v_act, v_rew, v_cost, v_term, v_trunc = (v[k] for k in ("active", "obs", "act", "rew", "cost", "term", "trunc"))
    seen = 0
    in_cmd = False
    try:
        while True:
            in_cmd = False
            g = lib.fsrl_env_wait_go(gen_p, seen, spin, 1000)
            if g == seen:                          # a second without a command: is the collector still there?
                if os.getppid() != parent:
                    break
                continue
            seen = g
            cmd = int(cmd_w[0])
            if cmd == _CMD_EXIT:
                break
            if int(want_w[0]) != g:                # this command does not touch my envs
                continue
            in_cmd = True
            act_w = v_active[sl]
            if act_w.all():                        # the whole slice (the steady state): basic slices, no index arrays
                local, gi = all_local, sl
            else:
                local = np.flatnonzero(act_w)
                gi = lo + local
            if len(local):
                if cmd == _CMD_RESET:
                    obs, _ = env.reset(local)
                    v_obs[gi] = obs
                else:
                    obs, rew, term, trunc, info = env.step(v_act[gi], local)
                    v_obs[gi] = obs; v_rew[gi] = rew; v_cost[gi] = info["cost"]
                    v_term[gi] = term; v_trunc[gi] = trunc
            lib.fsrl_env_done(pend_p)
    except BaseException:
        # an exception inside the user's env: say so in the shared block and release the collector (it checks the error
        # words after every wait and raises), then let the traceback reach stderr
        v["err"][w, 0] = 1
        if in_cmd:
            lib.fsrl_env_done(pend_p)
        raise
    finally:
        env.close()
        del v, cmd_w, want_w, v_active, v_obs, v_act, v_rew, v_cost, v_term, v_trunc
        shm.close()


AUTO_CHEAP_US = 25.0        # cap_workers="auto": an env step below this is all handshake
AUTO_CHEAP_WORKERS = 4


def _probe_step_cost_us(env, act_dim, steps=8):
    """host time of one env.step on a throw-away instance (median of a few steps from a reset state); None when the env cannot be
    stepped here"""
    import time
    from fsrl_amd.env.venv import _reset_one, _step_one
    try:
        _reset_one(env, None, {})
        sp = getattr(env, "action_space", None)
        lo, hi = np.asarray(getattr(sp, "low", -1.0), np.float64), np.asarray(getattr(sp, "high", 1.0), np.float64)
        act = np.where(np.isfinite(lo + hi), 0.5 * (lo + hi), 0.0).astype(np.float32).reshape(-1)[:act_dim]
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            out = _step_one(env, act)
            ts.append(time.perf_counter() - t0)
            if out[2] or out[3]:
                _reset_one(env, None, {})
        return float(np.median(ts)) * 1e6
    except Exception:                                       # noqa: BLE001 -- a probe must never take the construction down
        return None


class ShmemVectorEnv:
    def __init__(self, env_num=32, workers=None, obs_dim=8, act_dim=2, episode_len=300, seed=None, busy_us=0.0, cores=None,
                 start_method="spawn", spin_us=None, env_fns=None, cap_workers=False):
        """Two ways to say what the workers step: the synthetic dynamics (`env_num`, `obs_dim`, ... ; the bench), or
        `env_fns` -- one factory per env, the reference's `ShmemVectorEnv([lambda: gym.make(task) for _ in range(n)])`
        (also accepted as the first positional argument).  Factories travel to the spawned workers with cloudpickle, so
        lambdas and closures work; the envs are duck-typed (fsrl_amd/env/venv.py).  `seed` in this mode follows tianshou's
        `venv.seed(s)`: env i is reset with seed s + i the first time (None = the env's own default)."""
        if env_fns is None and not isinstance(env_num, (int, np.integer)):
            env_fns, env_num = env_num, None
        fns_pickled = None
        if env_fns is None and seed is None:
            seed = 0                                          # the synthetic dynamics are always seeded (the bench's default)
        if env_fns is not None:
            import cloudpickle
            env_fns = list(env_fns)
            env_num = len(env_fns)
            fns_pickled = [cloudpickle.dumps(fn) for fn in env_fns]
            probe = env_fns[0]()                              # like tianshou: one throw-away instance for the spaces
            self.observation_space, self.action_space = probe.observation_space, probe.action_space
            self.spec = getattr(probe, "spec", None)
            obs_dim, act_dim = int(np.prod(self.observation_space.shape)), int(np.prod(self.action_space.shape))
            step_cost_us = _probe_step_cost_us(probe, act_dim) if cap_workers == "auto" else None
            if hasattr(probe, "close"):
                probe.close()
            del probe
            episode_len = getattr(self.spec, "max_episode_steps", None)
        else:
            self.observation_space = Box(-np.inf, np.inf, (obs_dim, ))
            self.action_space = Box(-1.0, 1.0, (act_dim, ))
            self.spec = SimpleNamespace(id="SyntheticSafety-v0", max_episode_steps=episode_len)
        workers = env_num if workers is None else int(workers)
        assert 1 <= workers <= env_num
        # cap_workers=True: never more worker PROCESSES than CPUs this process may use (affinity and cgroup quota; the rank's core
        # slice when `cores` is given) -- the envs are spread over the workers, several per process.  It pays for envs that cost
        # (almost) nothing per step, where a vector step is the wake-up of the sleeping workers; it LOSES for envs with a real
        # step cost: one process per env on fewer CPUs runs in rounds either way, but with a process per env the two lanes of the
        # split-phase collect keep every CPU busy while the other lane's actor call is in flight (MI355X box, 16 usable CPUs,
        # 32 envs x 100 us: 0.68 of the env bound with 16 processes, 0.83-0.89 with 32).  Default off = tianshou's one process
        # per env.
        # cap_workers="auto" (r5): decided by what ONE env step costs -- measured on the throw-away probe instance (env_fns) or known
        # (the synthetic dynamics: busy_us).  An env cheaper than AUTO_CHEAP_US per step is all handshake: it runs on
        # AUTO_CHEAP_WORKERS processes (32 envs x 0 us on this box: 4 processes 600-680 k env-steps/s, 16 processes 290-420 k, 32
        # processes 240-330 k); every other env keeps the requested process count (one per env: the reference's layout, and what
        # fills the CPUs at a real step cost).  `worker_mode` says what was chosen and why.
        self.workers_requested = workers
        self.worker_mode = f"as requested: {workers} processes"
        if cap_workers == "auto":
            cost = float(busy_us) if env_fns is None else step_cost_us
            if cost is not None and cost < AUTO_CHEAP_US and workers > AUTO_CHEAP_WORKERS:
                self.worker_mode = (f"auto: an env step costs {cost:.1f} us (< {AUTO_CHEAP_US:g}): {AUTO_CHEAP_WORKERS} processes "
                                    f"instead of {workers}")
                workers = AUTO_CHEAP_WORKERS
            else:
                self.worker_mode = (f"auto: an env step costs {cost if cost is not None else float('nan'):.1f} us: "
                                    f"{workers} processes as requested")
        elif cap_workers:
            from fsrl_amd.parallel import usable_cpus
            cap = max(1, int(min(usable_cpus(), len(cores)) if cores else usable_cpus()))
            workers = max(1, min(workers, cap))
            self.worker_mode = f"capped at the usable CPUs: {workers} processes"
        self.env_num, self.obs_dim, self.act_dim, self.workers = env_num, obs_dim, act_dim, workers
        self.episode_len, self.busy_us = episode_len, busy_us
        self._lib = _load_lib()
        self._layout, size = _layout(env_num, obs_dim, act_dim, workers)
        self._shm = shared_memory.SharedMemory(create=True, size=size)
        self._v = _views(self._shm.buf, self._layout)
        for a in self._v.values():
            a[...] = 0
        ctx = mp.get_context(start_method)
        # contiguous slices: worker w owns envs [bounds[w], bounds[w + 1]); lane 0 = the first half of the workers
        self._bounds = [round(w * env_num / workers) for w in range(workers + 1)]
        self._owner = np.zeros(env_num, np.int32)
        self.n_lanes = 2 if workers >= 2 else 1
        half = (workers + 1) // 2 if self.n_lanes == 2 else workers
        self._lane_of_worker = np.array([0 if w < half else 1 for w in range(workers)], np.int32)
        # spinning before the futex sleep needs a CPU per worker plus one for the collector; on a smaller or quota-limited
        # host (the MI355X box: 256 CPUs visible, cgroup quota 16) nobody spins
        if spin_us is None:
            from fsrl_amd.parallel import usable_cpus
            usable = min(usable_cpus(), len(cores)) if cores else usable_cpus()      # affinity AND the cgroup CPU quota
            spin_us = 500.0 if workers + 1 <= usable else 0.0  # longer than the collector's own work between two commands
        self.spin_us = float(spin_us)
        self._spin = int(self.spin_us * 30)                   # ~30 ns per pause-and-load
        self._gen = [0, 0]
        self._inflight = [None, None]                         # per lane: env ids of the command in flight
        self._procs = []
        for w in range(workers):
            lo, hi = self._bounds[w], self._bounds[w + 1]
            self._owner[lo:hi] = w
            if fns_pickled is None:
                spec = dict(kind="synthetic", obs_dim=obs_dim, act_dim=act_dim, episode_len=episode_len, busy_us=busy_us,
                            seed=seed if workers == 1 else seed * 7919 + w)
            else:
                spec = dict(kind="fns", fns=fns_pickled[lo:hi], seed=seed)
            core = None if not cores else list(cores)[w % len(cores)]
            p = ctx.Process(target=_worker, args=(w, int(self._lane_of_worker[w]), lo, hi, self._shm.name, env_num, workers, obs_dim,
                                                  act_dim, spec, core, self._spin), daemon=True)
            p.start()
            self._procs.append(p)
        # the native collector probes these with waitid(): only meaningful when this process IS the workers' parent (spawn / fork).
        # A fork server's children are not ours -- pass no pids then (the error words and the 60 s timeout remain)
        self._pids = np.array([p.pid if start_method in ("spawn", "fork") else 0 for p in self._procs], np.int32)
        self.lane_of_env = self._lane_of_worker[self._owner]
        self.lanes = [np.flatnonzero(self.lane_of_env == l) for l in range(self.n_lanes)]
        # lanes are contiguous env / worker ranges: the steady state (a command for a whole lane) uses basic slices
        self._lane_env = [slice(int(l[0]), int(l[-1]) + 1) for l in self.lanes]
        self._lane_wrk = [slice(int(np.flatnonzero(self._lane_of_worker == l)[0]), int(np.flatnonzero(self._lane_of_worker == l)[-1]) + 1)
                          for l in range(self.n_lanes)]
        hs = self._v["hs"]
        self._gen_p = [hs[l, 0].ctypes.data for l in range(2)]
        self._pend_p = [hs[l, 1].ctypes.data for l in range(2)]
        self._closed = False

    def __len__(self):
        return self.env_num

    # ---- the handshake: post a command to a lane, wait for it
    def mark_broken(self, why):
        """sticky: every later command raises (the handshake words can no longer be trusted)"""
        self._broken = str(why)

    def _post(self, lane, cmd, ids):
        if getattr(self, "_broken", None):
            raise RuntimeError(f"ShmemVectorEnv is unusable: {self._broken}")
        assert self._inflight[lane] is None, "a command is already in flight on this lane"
        v = self._v
        self._gen[lane] = (self._gen[lane] + 1) & 0x7FFFFFFF or 1
        g = self._gen[lane]
        es = self._lane_env[lane]
        if len(ids) == es.stop - es.start:               # the whole lane
            v["active"][es] = 1
            ws = self._lane_wrk[lane]
            v["want"][ws, 0] = g
            n = ws.stop - ws.start
        else:
            v["active"][es] = 0
            v["active"][ids] = 1
            touched = np.unique(self._owner[ids])
            v["want"][touched, 0] = g
            n = len(touched)
        v["hs"][lane, 2, 0] = cmd
        self._inflight[lane] = ids
        self._lib.fsrl_env_post(self._gen_p[lane], self._pend_p[lane], n, g)

    def _wait(self, lane):
        ids = self._inflight[lane]
        assert ids is not None, "nothing in flight on this lane"
        waited = 0
        while self._lib.fsrl_env_wait_done(self._pend_p[lane], self._spin, 2000) != 0:
            waited += 2
            dead = [w for w in np.unique(self._owner[ids]) if not self._procs[w].is_alive()]
            if dead or waited >= 60:
                self._inflight[lane] = None
                raise RuntimeError(f"env workers {dead or 'of lane %d' % lane} did not answer")
        self._inflight[lane] = None
        if self._v["err"][:, 0].any():
            bad = np.flatnonzero(self._v["err"][:, 0]).tolist()
            raise RuntimeError(f"env workers {bad} raised inside their env (traceback on their stderr); the vector env cannot be used any more")
        return ids

    def _split(self, ids):
        """ids (any order) -> [(lane, positions into ids)] for the lanes they touch"""
        lane = self.lane_of_env[ids]
        return [(l, np.flatnonzero(lane == l)) for l in range(self.n_lanes) if (lane == l).any()]

    def _run(self, cmd, ids):
        parts = self._split(ids)
        for l, pos in parts:
            self._post(l, cmd, ids[pos])
        for l, _ in parts:
            self._wait(l)

    def reset(self, ids=None, **kwargs):
        if kwargs:                    # the workers call env.reset(local): nothing carries keyword arguments to them
            raise TypeError(f"ShmemVectorEnv.reset does not forward reset kwargs to its workers: {sorted(kwargs)}")
        ids = np.arange(self.env_num) if ids is None else np.asarray(ids)
        self._run(_CMD_RESET, ids)
        return self._v["obs"][ids].copy(), {}

    def _results(self, ids):
        v = self._v
        # a contiguous ascending range: basic slices.  The length test alone also holds for a permuted range like [0, 2, 1, 3]
        # (step(act, ids) accepts any order), so the order is checked as well
        if len(ids) and len(ids) == int(ids[-1]) - int(ids[0]) + 1 and (len(ids) == 1 or bool((np.diff(ids) == 1).all())):
            ids = slice(int(ids[0]), int(ids[-1]) + 1)
            return (v["obs"][ids].copy(), v["rew"][ids].copy(), v["term"][ids].astype(bool), v["trunc"][ids].astype(bool),
                    {"cost": v["cost"][ids].copy()})
        return (v["obs"][ids], v["rew"][ids], v["term"][ids].astype(bool), v["trunc"][ids].astype(bool), {"cost": v["cost"][ids]})

    def step(self, act, ids=None):
        ids = np.arange(self.env_num) if ids is None else np.asarray(ids)
        self._v["act"][ids] = np.asarray(act, np.float32).reshape(len(ids), self.act_dim)
        self._run(_CMD_STEP, ids)
        return self._results(ids)

    # ---- native collector loop (fsrl_collect_run): the shared block as a C struct
    def native_desc(self):
        """struct fsrl_shm_env for fsrl_collect_run (include/fsrl_hip.h); the generation counters travel with it (sync_native
        after the call).  No command may be in flight."""
        if getattr(self, "_broken", None):
            raise RuntimeError(f"ShmemVectorEnv is unusable: {self._broken}")
        assert self._inflight == [None, None]
        from fsrl_amd._lib import ShmEnv
        if getattr(self, "_desc", None) is None:
            v = self._v
            self._owner32 = np.ascontiguousarray(self._owner, np.int32)
            self._low32 = np.ascontiguousarray(self._lane_of_worker, np.int32)
            d = ShmEnv()
            for k in ("obs", "act", "rew", "cost", "term", "trunc", "active", "hs", "want"):
                setattr(d, k, v[k].ctypes.data)
            d.owner, d.lane_of_worker = self._owner32.ctypes.data, self._low32.ctypes.data
            d.err, d.pids = v["err"].ctypes.data, self._pids.ctypes.data
            d.env_num, d.obs_dim, d.act_dim, d.workers, d.n_lanes = self.env_num, self.obs_dim, self.act_dim, self.workers, self.n_lanes
            d.spin = self._spin
            self._desc = d
        self._desc.gen[0], self._desc.gen[1] = self._gen[0], self._gen[1]
        return self._desc

    def sync_native(self):
        self._gen[0], self._gen[1] = int(self._desc.gen[0]), int(self._desc.gen[1])

    # ---- split-phase interface: one lane steps while the collector works on the other
    def step_async(self, act, ids):
        """Start a step of `ids` (all in ONE lane) and return at once; `step_wait(ids)` collects it."""
        ids = np.asarray(ids)
        lane = int(self.lane_of_env[ids[0]])
        assert (self.lane_of_env[ids] == lane).all(), "step_async: the ids of one call belong to one lane"
        self._v["act"][ids] = np.asarray(act, np.float32).reshape(len(ids), self.act_dim)
        self._post(lane, _CMD_STEP, ids)

    def step_wait(self, ids):
        ids = np.asarray(ids)
        got = self._wait(int(self.lane_of_env[ids[0]]))
        assert len(got) == len(ids)
        return self._results(ids)

    def close(self):
        if self._closed:
            return
        self._closed = True
        for l in range(2):
            self._v["hs"][l, 2, 0] = _CMD_EXIT
            self._gen[l] = (self._gen[l] + 1) & 0x7FFFFFFF or 1
            self._lib.fsrl_env_post(self._gen_p[l], self._pend_p[l], 0, self._gen[l])
        for p in self._procs:
            p.join(5)
            if p.is_alive():
                p.terminate()
        self._v = None
        self._desc = None
        self._gen_p = self._pend_p = None
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
