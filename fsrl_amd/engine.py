"""numpy-facing wrapper of one libfsrl_hip context (one per GPU / per agent).

This is plumbing: every method forwards to one C-ABI entry point of include/fsrl_hip.h.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib

_f32p, _f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)
_u8p, _i32p, _i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64)


def _ptr(a, t):
    return a.ctypes.data_as(t) if a is not None else None


@dataclass
class EngineConfig:
    """Mirror of struct fsrl_config; defaults = PPOLagAgent defaults (ppo_lag_agent.py:82-116)."""
    obs_dim: int = 8
    act_dim: int = 2
    hidden: int = 128                 # two hidden layers of this width ...
    hidden_sizes: Optional[Sequence[int]] = None   # ... or the agents' hidden_sizes (wins over `hidden`): two layers of at most 256
                                                   # units run on the fused kernels; any other tuple (1 .. 8 layers, any width)
                                                   # makes a layered context (include/fsrl_hip.h fsrl_config.n_hidden)
    force_layered: bool = False       # tests: a two-layer network through the layered kernels too
    n_critics: int = 2
    env_num: int = 20
    buffer_size: int = 100000
    max_action: float = 1.0
    gamma: float = 0.99
    gae_lambda: float = 0.95
    eps_clip: float = 0.2
    dual_clip: Optional[float] = None
    vf_coef: float = 0.25
    max_grad_norm: Optional[float] = None
    target_kl: float = 0.02
    norm_adv: bool = True
    use_lagrangian: bool = True
    lr: float = 5e-4
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-8
    recompute_adv: bool = False
    unbounded: bool = False           # on-policy actor head without max_action * tanh
    rew_norm: bool = False            # reward_normalization (base_policy.py:430-444)
    value_clip: bool = False          # PPO clipped value loss (needs rew_norm)
    algo: int = _lib.ALGO_PPO_LAG

    def to_c(self):
        c = _lib.Config()
        c.algo, c.obs_dim, c.act_dim, c.hidden = self.algo, self.obs_dim, self.act_dim, self.hidden
        if self.hidden_sizes is not None:
            hs = [int(h) for h in self.hidden_sizes]
            if not 1 <= len(hs) <= 8:
                raise ValueError(f"the HIP path runs MLPs with 1 to 8 hidden layers, got hidden_sizes={tuple(hs)}")
            c.hidden, c.n_hidden = 0, len(hs)
            for i, h in enumerate(hs):
                c.hidden_sizes[i] = h
            c.force_layered = int(self.force_layered)
        c.n_critics, c.env_num, c.buffer_size = self.n_critics, self.env_num, self.buffer_size
        c.max_action, c.gamma, c.gae_lambda = self.max_action, self.gamma, self.gae_lambda
        c.eps_clip, c.dual_clip = self.eps_clip, (self.dual_clip or 0.0)
        c.vf_coef, c.max_grad_norm = self.vf_coef, (self.max_grad_norm or 0.0)
        tk = self.target_kl
        c.target_kl = 0.0 if (tk is None or not np.isfinite(tk) or tk >= 1e8) else tk
        c.norm_adv, c.use_lagrangian = int(self.norm_adv), int(self.use_lagrangian)
        c.lr, c.beta1, c.beta2, c.adam_eps = self.lr, self.beta1, self.beta2, self.adam_eps
        c.recompute_adv = int(self.recompute_adv)
        c.unbounded, c.rew_norm, c.value_clip = int(self.unbounded), int(self.rew_norm), int(self.value_clip)
        return c


class Engine:
    def __init__(self, cfg: EngineConfig, device: int = 0):
        self.lib = _lib.load()
        self.cfg = cfg
        self._ctx = C.c_void_p()
        ccfg = cfg.to_c()
        _lib.check(self.lib.fsrl_ctx_create(int(device), C.byref(ccfg), C.byref(self._ctx)))
        self.n_params = int(self.lib.fsrl_param_count(self._ctx))
        self._act_stage = None
        self._collect_stage = None
        self._run_stage = None      # cached staging arrays + ctypes pointers of the collector's hot calls
        self._push_stage = None

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self.lib.fsrl_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _lib.check(self.lib.fsrl_sync(self._ctx))

    # ---------------------------------------------------------------- parameters
    def set_params(self, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        _lib.check(self.lib.fsrl_params_set(self._ctx, _ptr(flat, _f32p), flat.size))

    def get_params(self):
        out = np.empty(self.n_params, np.float32)
        _lib.check(self.lib.fsrl_params_get(self._ctx, _ptr(out, _f32p), out.size))
        return out

    def get_grads(self):
        out = np.empty(self.n_params, np.float32)
        _lib.check(self.lib.fsrl_grads_get(self._ctx, _ptr(out, _f32p), out.size))
        return out

    def state_snapshot(self):
        """Checkpoint parameters + Adam state in HBM (device-to-device, asynchronous)."""
        _lib.check(self.lib.fsrl_state_snapshot(self._ctx))

    def state_restore(self):
        _lib.check(self.lib.fsrl_state_restore(self._ctx))

    def optim_reset(self):
        _lib.check(self.lib.fsrl_optim_reset(self._ctx))

    def ret_rms_get(self) -> np.ndarray:
        """[n_critics, 3] running (mean, var, count) of the normalised returns (BasePolicy.ret_rms)."""
        out = np.zeros(3 * self.cfg.n_critics, np.float64)
        _lib.check(self.lib.fsrl_ret_rms_get(self._ctx, _ptr(out, _f64p), out.size))
        return out.reshape(-1, 3)

    def ret_rms_set(self, rms):
        v = np.ascontiguousarray(rms, np.float64).reshape(-1)
        _lib.check(self.lib.fsrl_ret_rms_set(self._ctx, _ptr(v, _f64p), v.size))

    def set_lr(self, group: int, lr: float):
        """Learning rate of one optimiser (fsrl_set_lr): what lr_scheduler.step() produced on the host."""
        _lib.check(self.lib.fsrl_set_lr(self._ctx, int(group), float(lr)))

    def get_lr(self, group: int = 0) -> float:
        return float(self.lib.fsrl_get_lr(self._ctx, int(group)))

    def ppo_abort(self):
        _lib.check(self.lib.fsrl_ppo_abort(self._ctx))

    # ---------------------------------------------------------------- store
    def push(self, env_ids, obs, act, rew, cost, terminated, truncated, obs_next):
        """VectorReplayBuffer.add for k rows -> (ptr, ep_rew, ep_len, ep_idx).  Called once per vector step:
        inputs are copied into cached staging arrays whose ctypes pointers are built once."""
        k = len(env_ids)
        st = self._push_stage
        if st is None or st["cap"] < k:
            cap, Do, Da = max(k, self.cfg.env_num), self.cfg.obs_dim, self.cfg.act_dim
            arr = dict(ids=np.empty(cap, np.int32), obs=np.empty((cap, Do), np.float32), act=np.empty((cap, Da), np.float32),
                       rew=np.empty(cap, np.float64), cost=np.empty(cap, np.float64), term=np.empty(cap, np.uint8),
                       trunc=np.empty(cap, np.uint8), nxt=np.empty((cap, Do), np.float32), ptr=np.empty(cap, np.int64),
                       er=np.empty(cap, np.float64), el=np.empty(cap, np.int32), ei=np.empty(cap, np.int64))
            types = dict(ids=_i32p, obs=_f32p, act=_f32p, rew=_f64p, cost=_f64p, term=_u8p, trunc=_u8p, nxt=_f32p, ptr=_i64p,
                         er=_f64p, el=_i32p, ei=_i64p)
            st = self._push_stage = dict(cap=cap, a=arr, p={n: _ptr(arr[n], types[n]) for n in arr})
        a, p = st["a"], st["p"]
        obs = np.asarray(obs, np.float32).reshape(k, -1); act = np.asarray(act, np.float32).reshape(k, -1)
        assert obs.shape[1] == self.cfg.obs_dim and act.shape[1] == self.cfg.act_dim
        a["ids"][:k] = env_ids; a["obs"][:k] = obs; a["act"][:k] = act; a["rew"][:k] = rew
        a["term"][:k] = terminated; a["trunc"][:k] = truncated
        a["nxt"][:k] = np.asarray(obs_next, np.float32).reshape(k, -1)
        if cost is not None:
            a["cost"][:k] = cost
        _lib.check(self.lib.fsrl_store_push(self._ctx, p["ids"], k, p["obs"], p["act"], p["rew"],
                                            p["cost"] if cost is not None else None, p["term"], p["trunc"], p["nxt"],
                                            p["ptr"], p["er"], p["el"], p["ei"]))
        return a["ptr"][:k].copy(), a["er"][:k].copy(), a["el"][:k].copy(), a["ei"][:k].copy()

    def reset_store(self, keep_statistics=False):
        _lib.check(self.lib.fsrl_store_reset(self._ctx, int(keep_statistics)))

    def __len__(self):
        return int(self.lib.fsrl_store_len(self._ctx))

    def store_read(self, indices):
        """buffer[indices] -> dict(obs, act, rew, cost, terminated, truncated, obs_next) on the host."""
        idx = np.ascontiguousarray(indices, np.int64).reshape(-1)
        n, Do, Da = idx.size, self.cfg.obs_dim, self.cfg.act_dim
        out = dict(obs=np.empty((n, Do), np.float32), act=np.empty((n, Da), np.float32), rew=np.empty(n, np.float64),
                   cost=np.empty(n, np.float64), terminated=np.empty(n, np.uint8), truncated=np.empty(n, np.uint8),
                   obs_next=np.empty((n, Do), np.float32))
        _lib.check(self.lib.fsrl_store_read(self._ctx, _ptr(idx, _i64p), n, _ptr(out["obs"], _f32p), _ptr(out["act"], _f32p),
                                            _ptr(out["rew"], _f64p), _ptr(out["cost"], _f64p), _ptr(out["terminated"], _u8p),
                                            _ptr(out["truncated"], _u8p), _ptr(out["obs_next"], _f32p)))
        out["terminated"] = out["terminated"].astype(bool); out["truncated"] = out["truncated"].astype(bool)
        return out

    def store_configure(self, total_size: int, buffer_num: int):
        """VectorReplayBuffer(total_size, buffer_num): re-cut the allocated store (empties it)."""
        _lib.check(self.lib.fsrl_store_configure(self._ctx, int(total_size), int(buffer_num)))

    def store_geometry(self):
        """-> (rows per sub-buffer, sub-buffers in use)"""
        sub, num = C.c_int64(), C.c_int32()
        _lib.check(self.lib.fsrl_store_geometry(self._ctx, C.byref(sub), C.byref(num)))
        return int(sub.value), int(num.value)

    def sample0(self):
        n = C.c_int64()
        cap = len(self)
        out = np.empty(max(cap, 1), np.int64)
        _lib.check(self.lib.fsrl_store_sample0(self._ctx, _ptr(out, _i64p), out.size, C.byref(n)))
        return out[:n.value]

    # ---------------------------------------------------------------- inference
    def actor_forward(self, obs):
        obs = np.ascontiguousarray(obs, np.float32).reshape(-1, self.cfg.obs_dim)
        k = obs.shape[0]
        mu = np.empty((k, self.cfg.act_dim), np.float32)
        sigma = np.empty((k, self.cfg.act_dim), np.float32)
        _lib.check(self.lib.fsrl_actor_forward(self._ctx, _ptr(obs, _f32p), k, _ptr(mu, _f32p),
                                               _ptr(sigma, _f32p)))
        return mu, sigma

    def actor_sample(self, obs, deterministic=False, seed=0):
        """Collector-time actions a ~ pi(.|obs): actor on the device, noise from the library RNG.
        Hot loop of the collector: staging arrays and their ctypes pointers are cached."""
        obs = np.asarray(obs, np.float32).reshape(-1, self.cfg.obs_dim)
        k = obs.shape[0]
        st = self._act_stage
        if st is None or st[0].shape[0] < k:
            cap = max(k, self.cfg.env_num)
            so, sa = np.empty((cap, self.cfg.obs_dim), np.float32), np.empty((cap, self.cfg.act_dim), np.float32)
            st = self._act_stage = (so, sa, _ptr(so, _f32p), _ptr(sa, _f32p))
        np.copyto(st[0][:k], obs)
        _lib.check(self.lib.fsrl_actor_sample(self._ctx, st[2], k, int(deterministic), int(seed), st[3]))
        return st[1][:k].copy()

    def collect_step(self, prev, obs_act, deterministic=False, bound_method=1, low=None, high=None):
        """One vector step of the collector in one C call (fsrl_collect_step): store the finished transitions `prev` =
        (env_ids, obs, act, rew, cost, terminated, truncated, obs_next) or None, and return (act, env_act, ep_rew,
        ep_len) for the observations `obs_act` (None: store only).  Staging arrays / ctypes pointers are cached."""
        Do, Da = self.cfg.obs_dim, self.cfg.act_dim
        st = self._collect_stage
        if st is None:
            cap = self.cfg.env_num
            arr = dict(ids=np.empty(cap, np.int32), obs=np.empty((cap, Do), np.float32), act=np.empty((cap, Da), np.float32),
                       rew=np.empty(cap, np.float64), cost=np.empty(cap, np.float64), term=np.empty(cap, np.uint8),
                       trunc=np.empty(cap, np.uint8), nxt=np.empty((cap, Do), np.float32), ptr=np.empty(cap, np.int64),
                       er=np.empty(cap, np.float64), el=np.empty(cap, np.int32), ei=np.empty(cap, np.int64),
                       oa=np.empty((cap, Do), np.float32), ao=np.empty((cap, Da), np.float32), eo=np.empty((cap, Da), np.float32),
                       lo=np.empty(Da, np.float32), hi=np.empty(Da, np.float32))
            types = dict(ids=_i32p, obs=_f32p, act=_f32p, rew=_f64p, cost=_f64p, term=_u8p, trunc=_u8p, nxt=_f32p, ptr=_i64p,
                         er=_f64p, el=_i32p, ei=_i64p, oa=_f32p, ao=_f32p, eo=_f32p, lo=_f32p, hi=_f32p)
            st = self._collect_stage = dict(a=arr, p={n: _ptr(arr[n], types[n]) for n in arr})
        a, p = st["a"], st["p"]
        k = 0
        if prev is not None:
            ids, obs, act, rew, cost, term, trunc, nxt = prev
            k = len(ids)
            a["ids"][:k] = ids; a["obs"][:k] = obs; a["act"][:k] = act; a["rew"][:k] = rew; a["cost"][:k] = cost
            a["term"][:k] = term; a["trunc"][:k] = trunc; a["nxt"][:k] = nxt
        ka = 0
        if obs_act is not None:
            ka = len(obs_act)
            a["oa"][:ka] = obs_act
        if low is not None:
            a["lo"][:] = low; a["hi"][:] = high
        _lib.check(self.lib.fsrl_collect_step(
            self._ctx, p["ids"], k, p["obs"], p["act"], p["rew"], p["cost"], p["term"], p["trunc"], p["nxt"], p["ptr"],
            p["er"], p["el"], p["ei"], p["oa"], ka, int(deterministic), int(bound_method),
            p["lo"] if low is not None else None, p["hi"] if low is not None else None, p["ao"], p["eo"]))
        return a["ao"][:ka].copy(), a["eo"][:ka].copy(), a["er"][:k], a["el"][:k]

    def collect_run(self, env_desc, ready, obs, act, env_act, deterministic=False, bound_method=1, low=None, high=None,
                    max_steps=1 << 30):
        """The collector's inner loop over a worker-process env in one C call (fsrl_collect_run): from the envs `ready` with
        observations `obs`, policy actions `act` and mapped actions `env_act`, step the env / store / evaluate the actor until
        a vector step finishes an episode.  -> (steps, cost_sum, obs, act, rew, cost, terminated, truncated, obs_next) where
        obs / act are the inputs of that last step and rew ... obs_next its (unstored) results."""
        Do, Da = self.cfg.obs_dim, self.cfg.act_dim
        n = len(ready)
        st = self._run_stage
        if st is None or st["n"] < n:
            cap = self.cfg.env_num
            arr = dict(ids=np.empty(cap, np.int32), obs=np.empty((cap, Do), np.float32), act=np.empty((cap, Da), np.float32),
                       eact=np.empty((cap, Da), np.float32), rew=np.empty(cap, np.float64), cost=np.empty(cap, np.float64),
                       term=np.empty(cap, np.uint8), trunc=np.empty(cap, np.uint8), nxt=np.empty((cap, Do), np.float32),
                       lo=np.empty(Da, np.float32), hi=np.empty(Da, np.float32))
            types = dict(ids=_i32p, obs=_f32p, act=_f32p, eact=_f32p, rew=_f64p, cost=_f64p, term=_u8p, trunc=_u8p, nxt=_f32p,
                         lo=_f32p, hi=_f32p)
            st = self._run_stage = dict(n=cap, a=arr, p={k: _ptr(arr[k], types[k]) for k in arr}, steps=C.c_int32(), csum=C.c_double())
        a, p = st["a"], st["p"]
        a["ids"][:n] = ready; a["obs"][:n] = obs; a["act"][:n] = act; a["eact"][:n] = env_act
        if low is not None:
            a["lo"][:] = low; a["hi"][:] = high
        _lib.check(self.lib.fsrl_collect_run(
            self._ctx, C.byref(env_desc), p["ids"], n, p["obs"], p["act"], p["eact"], int(deterministic), int(bound_method),
            p["lo"] if low is not None else None, p["hi"] if low is not None else None, int(max_steps), C.byref(st["steps"]),
            C.byref(st["csum"]), p["rew"], p["cost"], p["term"], p["trunc"], p["nxt"]))
        return (st["steps"].value, st["csum"].value, a["obs"][:n].copy(), a["act"][:n].copy(), a["rew"][:n].copy(),
                a["cost"][:n].copy(), a["term"][:n].astype(bool), a["trunc"][:n].astype(bool), a["nxt"][:n].copy())

    def collect_episodes(self, env_desc, ready, obs, n_episode, deterministic=False, bound_method=1, low=None, high=None, split=False):
        """FastCollector.collect(n_episode) over a worker-process env in ONE C call (fsrl_collect_episodes): vector steps, store,
        actor, resets, episode accounting, surplus envs.  split: the two-lane split-phase form (fsrl_collect_episodes_split).
        -> dict(steps, total_cost, terminated, truncated, ep_rews, ep_lens, t_env, t_act)"""
        Do, Da = self.cfg.obs_dim, self.cfg.act_dim
        n = len(ready)
        ids = np.ascontiguousarray(ready, np.int32)
        ob = np.ascontiguousarray(obs, np.float32).reshape(n, Do)
        lo = np.ascontiguousarray(low, np.float32) if low is not None else None
        hi = np.ascontiguousarray(high, np.float32) if low is not None else None
        steps, cost = C.c_int64(), C.c_double()
        nt, ntr, nep = C.c_int32(), C.c_int32(), C.c_int32()
        ep_rew, ep_len = np.zeros(int(n_episode), np.float64), np.zeros(int(n_episode), np.int32)
        fn = self.lib.fsrl_collect_episodes_split if split else self.lib.fsrl_collect_episodes
        _lib.check(fn(
            self._ctx, C.byref(env_desc), _ptr(ids, _i32p), n, _ptr(ob, _f32p), int(n_episode), int(deterministic), int(bound_method),
            _ptr(lo, _f32p) if lo is not None else None, _ptr(hi, _f32p) if hi is not None else None, C.byref(steps), C.byref(cost),
            C.byref(nt), C.byref(ntr), _ptr(ep_rew, _f64p), _ptr(ep_len, _i32p), C.byref(nep)))
        k = nep.value
        tm = np.zeros(2, np.float64)
        _lib.check(self.lib.fsrl_collect_timing(self._ctx, _ptr(tm, _f64p)))
        return dict(steps=int(steps.value), total_cost=float(cost.value), terminated=int(nt.value), truncated=int(ntr.value),
                    ep_rews=ep_rew[:k], ep_lens=ep_len[:k], t_env=float(tm[0]), t_act=float(tm[1]))

    def actor_set_resident(self, on=True, idle_timeout_us=0.0):
        """The collector's actor as a resident workgroup (include/fsrl_hip.h: fsrl_actor_set_resident); on by default."""
        _lib.check(self.lib.fsrl_actor_set_resident(self._ctx, int(bool(on)), float(idle_timeout_us)))

    def actor_release(self):
        """End the resident actor kernel now (a collect is over) instead of at the next stream work / its idle timeout."""
        self.lib.fsrl_actor_release(self._ctx)

    def actor_resident_stats(self):
        out = np.zeros(3, np.int64)
        _lib.check(self.lib.fsrl_actor_resident_stats(self._ctx, _ptr(out, _i64p)))
        return dict(launches=int(out[0]), requests=int(out[1]), live=bool(out[2]))

    def store_sizes(self, n=None):
        n = self.cfg.env_num if n is None else int(n)
        out = np.empty(n, np.int64)
        _lib.check(self.lib.fsrl_store_sizes(self._ctx, _ptr(out, _i64p), n))
        return out

    # ---------------------------------------------------------------- FOCOPS (on the PPO begin / pass / end calls)
    def focops_init(self, actor_lr=5e-4, critic_lr=1e-3, l2_reg=1e-3, delta=0.02, eta=0.02, tem_lambda=0.95,
                    max_grad_norm=0.5):
        cfg = _lib.FocopsConfig(actor_lr, critic_lr, l2_reg, delta, eta, tem_lambda, float(max_grad_norm or 0.0))
        _lib.check(self.lib.fsrl_focops_init(self._ctx, C.byref(cfg)))

    def focops_set_plan(self, four_launch: int = 0):
        """A/B and tests: 1 = the four-launch minibatch step (split-K weight gradients), 0 = three launches (default)"""
        _lib.check(self.lib.fsrl_focops_set_plan(self._ctx, int(four_launch)))

    def focops_update(self, nu, nu_loss, batch_size, repeat, perms=None, seed=0):
        """-> (stats [steps, 8]: nu_loss, nu_value, actor_loss, kl, entropy, vf0, vf1, vf_total; stopped pass or -1)."""
        _lib.check(self.lib.fsrl_focops_set_nu(self._ctx, float(nu), float(nu_loss)))
        stats, stopped = self.ppo_update([0.0], 1.0, batch_size, repeat, perms=perms, seed=seed)
        return stats[:, :_lib.FOCOPS_NSTATS], stopped

    # ---------------------------------------------------------------- PPO-Lagrangian
    def ppo_begin(self, lagrangians: Sequence[float], rescaling: float, batch_size: int) -> int:
        lag = np.ascontiguousarray(lagrangians, np.float64).reshape(-1)
        n = C.c_int64()
        _lib.check(self.lib.fsrl_ppo_begin(self._ctx, _ptr(lag, _f64p) if lag.size else None,
                                           float(rescaling), int(batch_size), C.byref(n)))
        self._n = n.value
        return n.value

    def ppo_pass(self, perm=None, seed: int = 0, wait: bool = True) -> bool:
        """One pass of minibatch steps -> True if the KL early stop fired.  wait=False only enqueues the pass (returns
        False); ppo_pass_result() then collects the verdict -- the caller can prepare the next pass meanwhile."""
        stopped = C.c_int32()
        if perm is not None:
            perm = np.ascontiguousarray(perm, np.int64)
            assert perm.size == self._n, "permutation length != batch length"
        _lib.check(self.lib.fsrl_ppo_pass(self._ctx, _ptr(perm, _i64p), int(seed), C.byref(stopped) if wait else None))
        return bool(stopped.value)

    def ppo_pass_result(self) -> bool:
        stopped = C.c_int32()
        _lib.check(self.lib.fsrl_ppo_pass_result(self._ctx, C.byref(stopped)))
        return bool(stopped.value)

    def ppo_end(self):
        n = C.c_int64()
        _lib.check(self.lib.fsrl_ppo_end(self._ctx, None, 0, C.byref(n)))  # count only: not allowed
        return n.value

    def ppo_end_stats(self, max_steps: int):
        n = C.c_int64()
        out = np.empty((max(max_steps, 1), _lib.PPO_NSTATS), np.float32)
        _lib.check(self.lib.fsrl_ppo_end(self._ctx, _ptr(out, _f32p), out.shape[0], C.byref(n)))
        return out[:n.value]

    def ppo_set_plan(self, tall_tiles: int = -1):
        """32-row tiles in the forward / backward launch of a minibatch step: -1 automatic, 0 none, n > 0 a count (A/B; same bits)"""
        _lib.check(self.lib.fsrl_ppo_set_plan(self._ctx, int(tall_tiles)))

    def ppo_update(self, lagrangians, rescaling, batch_size, repeat, perms=None, seed=0):
        """-> (stats [steps, 11] float32, stopped_pass or -1)."""
        n = self.ppo_begin(lagrangians, rescaling, batch_size)
        stopped_pass = -1
        steps_per_pass = max(1, -(-n // max(batch_size, 1)))
        try:
            for k in range(repeat):
                if self.ppo_pass(None if perms is None else perms[k], seed + k if seed else 0):
                    stopped_pass = k
                    break
            stats = self.ppo_end_stats(steps_per_pass * max(repeat, 1))
        except BaseException:
            self.ppo_abort()                     # a bad permutation / HIP error must not wedge the begin ... end state
            raise
        return stats, stopped_pass

    def batch_get(self, which: str):
        n = self._n
        cols = 1 if which == "logp_old" else self.cfg.n_critics
        out = np.empty((n, cols), np.float32)
        _lib.check(self.lib.fsrl_batch_get(self._ctx, which.encode(), _ptr(out, _f32p), out.size))
        return out[:, 0] if which == "logp_old" else out

    def gae_return(self, v, v_next, rew, end_flag, gamma, gae_lambda):
        v = np.ascontiguousarray(v, np.float32); v_next = np.ascontiguousarray(v_next, np.float32)
        rew = np.ascontiguousarray(rew, np.float64)
        end = np.ascontiguousarray(end_flag).astype(np.uint8)
        out = np.empty(rew.size, np.float64)
        _lib.check(self.lib.fsrl_gae_return(self._ctx, _ptr(v, _f32p), _ptr(v_next, _f32p),
                                            _ptr(rew, _f64p), _ptr(end, _u8p), rew.size, float(gamma),
                                            float(gae_lambda), _ptr(out, _f64p)))
        return out

    def launch_floors(self, mb_rows=256, iters=200):
        """fsrl_launch_floors: {"fwdbwd", "wgrad", "adam", "triple"} in microseconds, measured now on this device."""
        out = np.zeros(4, np.float64)
        _lib.check(self.lib.fsrl_launch_floors(self._ctx, int(mb_rows), int(iters), _ptr(out, _f64p)))
        return dict(zip(("fwdbwd", "wgrad", "adam", "triple"), out.tolist()))

    def nstep_return(self, metric, end_flag, target_q, indices, gamma, n_step):
        """nstep_return (base_policy.py:543-567) on the device; target_q [bsz, ...] float32, indices [n_step, bsz]."""
        metric = np.ascontiguousarray(metric, np.float64)
        end = np.ascontiguousarray(end_flag).astype(np.uint8)
        tq = np.ascontiguousarray(target_q, np.float32)
        shape = tq.shape
        bsz = shape[0] if tq.ndim else 0
        tq2 = tq.reshape(bsz, -1) if bsz else tq.reshape(0, 1)
        idx = np.ascontiguousarray(indices, np.int64).reshape(int(n_step), bsz) if n_step >= 1 else np.zeros((0, bsz), np.int64)
        out = np.empty(tq2.shape, np.float64)
        _lib.check(self.lib.fsrl_nstep_return(self._ctx, _ptr(metric, _f64p), _ptr(end, _u8p), metric.size,
                                              _ptr(tq2, _f32p), _ptr(idx, _i64p), bsz, tq2.shape[1], float(gamma),
                                              int(n_step), _ptr(out, _f64p)))
        return out.reshape(shape)

    # ---------------------------------------------------------------- CPO / TRPO-Lagrangian
    def tr_begin(self, target_kl=0.01, backtrack_coeff=0.8, damping=0.1, l2_reg=0.0, critic_lr=1e-3,
                 max_backtracks=10, optim_critic_iters=10, cg_iters=10, norm_adv=True,
                 cost_limit=float("inf")) -> int:
        cfg = _lib.TrConfig(target_kl, backtrack_coeff, damping, l2_reg, critic_lr, max_backtracks,
                            optim_critic_iters, cg_iters, int(norm_adv),
                            cost_limit if np.isfinite(cost_limit) else 1e300)
        n = C.c_int64()
        _lib.check(self.lib.fsrl_tr_begin(self._ctx, C.byref(cfg), C.byref(n)))
        self._n = n.value
        return n.value

    @property
    def n_actor_params(self) -> int:
        return int(self.lib.fsrl_actor_param_count(self._ctx))

    def tr_linesearch_evals(self, cap=64):
        out = np.zeros(cap, np.int32)
        n = int(self.lib.fsrl_tr_linesearch_evals(self._ctx, _ptr(out, _i32p), cap))
        return out[:n]

    def tr_grad(self, which: int):
        out = np.empty(self.n_actor_params, np.float32)
        _lib.check(self.lib.fsrl_tr_grad(self._ctx, int(which), _ptr(out, _f32p), out.size))
        return out

    def tr_hvp(self, v):
        v = np.ascontiguousarray(v, np.float32)
        out = np.empty_like(v)
        _lib.check(self.lib.fsrl_tr_hvp(self._ctx, _ptr(v, _f32p), _ptr(out, _f32p), v.size))
        return out

    def tr_hvp_cached(self, v):
        """H v again at the theta / batch of the previous tr_hvp call: the cached-activation kernel of the CG solves."""
        v = np.ascontiguousarray(v, np.float32)
        out = np.empty_like(v)
        _lib.check(self.lib.fsrl_tr_hvp_cached(self._ctx, _ptr(v, _f32p), _ptr(out, _f32p), v.size))
        return out

    def tr_set_plan(self, tile_rows: int = 0, hvp: int = 0, wgrad: int = 0):
        """Kernel plan of the full-batch path (A/B timing, bit-identity tests); 0 / 0 = automatic.  tile_rows + 64: critic steps on the
        compute stream; + 128: read-backs by copy + stream synchronisation instead of polled completion words."""
        _lib.check(self.lib.fsrl_tr_set_plan(self._ctx, int(tile_rows), int(hvp), int(wgrad)))

    def tr_set_tile_split(self, n32_tile: int = -1, n32_hvp: int = -1):
        """A/B only: forced 32-row tile counts of the co-resident tile / cached-HVP launches (-1 = the dispatch simulation)."""
        _lib.check(self.lib.fsrl_tr_set_tile_split(self._ctx, int(n32_tile), int(n32_hvp)))

    def tr_set_co_delay(self, tile_periods: int = -1, hvp_periods: int = -1):
        """A/B only: start offset of every CU's second co-resident workgroup (periods of 8 128 cycles; -1 = default)."""
        _lib.check(self.lib.fsrl_tr_set_co_delay(self._ctx, int(tile_periods), int(hvp_periods)))

    def tr_eval(self):
        out = np.zeros(8, np.float64)
        _lib.check(self.lib.fsrl_tr_eval(self._ctx, _ptr(out, _f64p)))
        return out

    def _tr_perms(self, perms, repeat, batch_size):
        """perms: None or [repeat][N] -> (pointer or None, keep-alive array, minibatches per repeat)"""
        n = max(int(self._n), 0)
        b = max(1, min(int(batch_size), max(n, 1)))
        nmb = max(1, n // b) if n else 1                # Batch.split(merge_last=True): the remainder joins the last chunk
        if perms is None:
            return None, None, nmb
        a = np.ascontiguousarray(np.stack([np.asarray(p, np.int64) for p in perms]), np.int64)
        assert a.shape == (repeat, n), "perms must be [repeat][N]"
        return _ptr(a, _i64p), a, nmb

    def cpo_learn(self, ave_cost_return: float, repeat: int, batch_size: int = 2**31 - 1, perms=None, seed: int = 0):
        """CPO.learn (cpo.py:353-370) on the batch of the last tr_begin.  batch_size < N: Batch.split minibatches of the
        permutations `perms` ([repeat][N]; None = library shuffle), one stats row per minibatch."""
        pp, keep, nmb = self._tr_perms(perms, repeat, batch_size)
        out = np.empty((repeat * nmb, _lib.CPO_NSTATS), np.float32)
        rows = C.c_int64()
        _lib.check(self.lib.fsrl_cpo_learn_mb(self._ctx, float(ave_cost_return), int(repeat), int(min(batch_size, 2**31 - 1)), pp,
                                              int(seed), _ptr(out, _f32p), out.shape[0], C.byref(rows)))
        return out[:rows.value]

    def trpo_learn(self, lagrangians, rescaling: float, repeat: int, batch_size: int = 2**31 - 1, perms=None, seed: int = 0):
        lag = np.ascontiguousarray(lagrangians, np.float64).reshape(-1)
        pp, keep, nmb = self._tr_perms(perms, repeat, batch_size)
        out = np.empty((repeat * nmb, _lib.TRPO_NSTATS), np.float32)
        rows = C.c_int64()
        _lib.check(self.lib.fsrl_trpo_learn_mb(self._ctx, _ptr(lag, _f64p) if lag.size else None, float(rescaling), int(repeat),
                                               int(min(batch_size, 2**31 - 1)), pp, int(seed), _ptr(out, _f32p), out.shape[0],
                                               C.byref(rows)))
        return out[:rows.value]

    # ---------------------------------------------------------------- SAC-Lagrangian
    def sac_init(self, actor_lr=5e-4, critic_lr=1e-3, alpha_lr=3e-4, tau=0.05, alpha=0.005,
                 target_entropy=None, n_step=2, auto_alpha=True, use_lagrangian=True, deterministic=False,
                 exploration_sigma=0.1):
        """deterministic=True selects DDPG-Lagrangian (deterministic actor + target actor, single critics)."""
        te = -float(self.cfg.act_dim) if target_entropy is None else float(target_entropy)
        cfg = _lib.SacConfig(actor_lr, critic_lr, alpha_lr, tau, alpha, te, int(n_step), int(auto_alpha),
                             int(use_lagrangian), int(deterministic), float(exploration_sigma))
        _lib.check(self.lib.fsrl_sac_init(self._ctx, C.byref(cfg)))
        self.n_sac_actor = int(self.lib.fsrl_sac_param_count(self._ctx, 0))
        self.n_sac_critics = int(self.lib.fsrl_sac_param_count(self._ctx, 1))

    def sac_set_plan(self, plan: int = 0):
        """A/B and tests: bit 0 = split-K weight gradients at every batch size (default: the one-workgroup-per-tile kernel up to
        512 rows); bit 1 = sample and gather as two launches, bit 2 = the n-step targets as a launch of their own (default: 10
        launches per update, not 12), bit 3 = prefetch the next update's sample on the side stream (measured slower: off by default),
        bit 4 = sample + gather as a launch of their own (round 4: 10 launches; r5 default: inside the actors' forward launch, 9),
        bit 5 = no rider blocks (r6 default above 512 rows: the actor's weight-gradient launch also draws the NEXT update's batch),
        bit 6 = the critics' split-K weight-gradient launch in plain block order (r6 default: XCD-aware, 2.3x less memory-side traffic)"""
        _lib.check(self.lib.fsrl_sac_set_plan(self._ctx, int(plan)))

    def sac_set_params(self, actor_flat, critics_flat, log_alpha=0.0):
        a = np.ascontiguousarray(actor_flat, np.float32); c = np.ascontiguousarray(critics_flat, np.float32)
        _lib.check(self.lib.fsrl_sac_params_set(self._ctx, _ptr(a, _f32p), a.size, _ptr(c, _f32p), c.size,
                                                float(log_alpha)))

    def sac_put_params(self, which: int, flat):
        """Overwrite one parameter set (0 actor, 1 critics, 2 critics_old, 3 actor_old); nothing else changes."""
        a = np.ascontiguousarray(flat, np.float32)
        _lib.check(self.lib.fsrl_sac_params_put(self._ctx, int(which), _ptr(a, _f32p), a.size))

    def sac_get_params(self, which: int):
        """which: 0 actor, 1 critics, 2 critics_old, 3 actor_old (DDPG-Lag) -> (flat params, alpha)."""
        out = np.empty(self.n_sac_actor if which in (0, 3) else self.n_sac_critics, np.float32)
        alpha = C.c_float()
        _lib.check(self.lib.fsrl_sac_params_get(self._ctx, int(which), _ptr(out, _f32p), out.size, C.byref(alpha)))
        return out, float(alpha.value)

    def sac_update(self, batch_size, lagrangians, rescaling, indices=None, eps_target=None, eps_pi=None, seed=0,
                   sync=True):
        """One SAC-Lag update.  indices/eps_* given together = caller RNG (parity); all None = device
        RNG.  sync=False only enqueues the work: the statistics row is fetched later by sac_drain()."""
        lag = np.ascontiguousarray(lagrangians, np.float64).reshape(-1)
        idx = None if indices is None else np.ascontiguousarray(indices, np.int64)
        et = None if eps_target is None else np.ascontiguousarray(eps_target, np.float32)
        ep = None if eps_pi is None else np.ascontiguousarray(eps_pi, np.float32)
        if idx is not None:
            assert idx.size == batch_size
        out = np.empty(_lib.SAC_NSTATS, np.float32) if sync else None
        _lib.check(self.lib.fsrl_sac_update(self._ctx, int(batch_size), _ptr(idx, _i64p), _ptr(et, _f32p),
                                            _ptr(ep, _f32p), int(seed), _ptr(lag, _f64p) if lag.size else None,
                                            float(rescaling), _ptr(out, _f32p)))
        return out

    def sac_drain(self, max_rows=4096):
        """Statistics rows [n, SAC_NSTATS] of the sync=False updates since the last drain."""
        out = np.empty((int(max_rows), getattr(self, "_ring_cols", _lib.SAC_NSTATS)), np.float32)
        n = int(self.lib.fsrl_sac_stats_drain(self._ctx, _ptr(out, _f32p), int(max_rows)))
        if n < 0:
            _lib.check(n)
        return out[:n]

    def sac_last_sample(self, batch_size):
        idx = np.empty(int(batch_size), np.int64)
        et = np.empty((int(batch_size), self.cfg.act_dim), np.float32); ep = np.empty_like(et)
        _lib.check(self.lib.fsrl_sac_last_sample(self._ctx, _ptr(idx, _i64p), _ptr(et, _f32p), _ptr(ep, _f32p),
                                                 int(batch_size)))
        return idx, et, ep

    def sac_actor_forward(self, obs):
        obs = np.ascontiguousarray(obs, np.float32).reshape(-1, self.cfg.obs_dim)
        k = obs.shape[0]
        mu = np.empty((k, self.cfg.act_dim), np.float32); sigma = np.empty_like(mu)
        _lib.check(self.lib.fsrl_sac_actor_forward(self._ctx, _ptr(obs, _f32p), k, _ptr(mu, _f32p),
                                                   _ptr(sigma, _f32p)))
        return mu, sigma

    # ---------------------------------------------------------------- CVPO (on the SAC replay context)
    def cvpo_init(self, qc_thres, actor_lr=5e-4, critic_lr=1e-3, tau=0.05, n_step=2, double_critic=False,
                  sample_act_num=16, estep_iter_num=1, mstep_iter_num=1, estep_kl=0.02, estep_dual_max=20.0,
                  estep_dual_lr=0.02, mstep_kl_mu=0.005, mstep_kl_std=0.0005, mstep_dual_max=0.5, mstep_dual_lr=0.1):
        """fsrl_cvpo_init (cvpo.py:71-163).  Parameters then move through the sac_* accessors (which=3: actor_old)."""
        cfg = _lib.CvpoConfig(actor_lr, critic_lr, tau, int(n_step), int(double_critic), int(sample_act_num),
                              int(estep_iter_num), int(mstep_iter_num), estep_kl, estep_dual_max, estep_dual_lr,
                              mstep_kl_mu, mstep_kl_std, mstep_dual_max, mstep_dual_lr, float(qc_thres))
        _lib.check(self.lib.fsrl_cvpo_init(self._ctx, C.byref(cfg)))
        self.n_sac_actor = int(self.lib.fsrl_sac_param_count(self._ctx, 0))
        self.n_sac_critics = int(self.lib.fsrl_sac_param_count(self._ctx, 1))
        self._ring_cols = _lib.CVPO_NSTATS
        self._cvpo_k = int(sample_act_num)

    def cvpo_pre_update(self):
        _lib.check(self.lib.fsrl_cvpo_pre_update(self._ctx))

    def cvpo_post_update(self):
        _lib.check(self.lib.fsrl_cvpo_post_update(self._ctx))

    def cvpo_set_thres(self, qc_thres):
        _lib.check(self.lib.fsrl_cvpo_set_thres(self._ctx, float(qc_thres)))

    def cvpo_update(self, batch_size, indices=None, eps_target=None, eps_particles=None, seed=0, sync=True):
        """One CVPO.update.  indices / eps_target [B,Da] / eps_particles [K,B,Da] together = caller RNG; all None =
        device RNG.  sync=False only enqueues; rows come back through sac_drain()."""
        idx = None if indices is None else np.ascontiguousarray(indices, np.int64)
        et = None if eps_target is None else np.ascontiguousarray(eps_target, np.float32)
        ek = None if eps_particles is None else np.ascontiguousarray(eps_particles, np.float32)
        if idx is not None:
            assert idx.size == batch_size and et.size == batch_size * self.cfg.act_dim
            assert ek.size == self._cvpo_k * batch_size * self.cfg.act_dim
        out = np.empty(_lib.CVPO_NSTATS, np.float32) if sync else None
        _lib.check(self.lib.fsrl_cvpo_update(self._ctx, int(batch_size), _ptr(idx, _i64p), _ptr(et, _f32p),
                                             _ptr(ek, _f32p), int(seed), _ptr(out, _f32p)))
        return out

    def cvpo_duals(self):
        """(eta, lambda, mstep_dual_mu, mstep_dual_std)"""
        out = np.empty(4, np.float32)
        _lib.check(self.lib.fsrl_cvpo_duals_get(self._ctx, _ptr(out, _f32p)))
        return out

    def cvpo_last_particles(self, batch_size):
        out = np.empty((self._cvpo_k, int(batch_size), self.cfg.act_dim), np.float32)
        _lib.check(self.lib.fsrl_cvpo_last_particles(self._ctx, _ptr(out, _f32p), out.size))
        return out

    # ---------------------------------------------------------------- metric exchange (RCCL through the C ABI)
    def comm_unique_id(self) -> bytes:
        buf = np.zeros(128, np.uint8)
        _lib.check(self.lib.fsrl_comm_unique_id(_ptr(buf, _u8p), 128))
        return buf.tobytes()

    def comm_init(self, rank: int, world: int, uid: bytes):
        buf = np.frombuffer(uid, np.uint8).copy()
        _lib.check(self.lib.fsrl_comm_init(self._ctx, int(rank), int(world), _ptr(buf, _u8p), buf.size))

    def comm_init_from_torch(self):
        """Join the library's own RCCL communicator using the process group torch.distributed already has (any backend,
        gloo included) only to hand rank 0's 128-byte id around; afterwards fsrl_metrics_allreduce needs torch no more."""
        import torch.distributed as dist
        assert dist.is_initialized(), "init a torch.distributed process group first (it only carries the id)"
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [self.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        self.comm_init(rank, world, box[0])

    def comm_info(self):
        r, w = C.c_int32(), C.c_int32()
        _lib.check(self.lib.fsrl_comm_info(self._ctx, C.byref(r), C.byref(w)))
        return int(r.value), int(w.value)

    def metrics_allreduce(self, vec):
        """Element-wise sum of a float64 vector (<= 64 entries) over the ranks; the identity without a communicator."""
        v = np.ascontiguousarray(vec, np.float64).copy()
        _lib.check(self.lib.fsrl_metrics_allreduce(self._ctx, _ptr(v, _f64p), v.size))
        return v

    def comm_destroy(self):
        _lib.check(self.lib.fsrl_comm_destroy(self._ctx))

    def set_profiling(self, on: bool):
        _lib.check(self.lib.fsrl_set_profiling(self._ctx, int(on)))

    def last_timing(self):
        out = np.zeros(5, np.float64)
        _lib.check(self.lib.fsrl_last_timing(self._ctx, _ptr(out, _f64p), 5))
        return dict(process_ms=out[0], learn_ms=out[1], fwdbwd_ms=out[2], fwdbwd_launches=int(out[3]),
                    fwdbwd_raw_ms=out[4])


class EngineGroup:
    """k PPO-Lagrangian engines of one network shape on one GPU, updated in lock step (fsrl_group_*): every launch of
    the minibatch step carries all members.  Members keep their own store, parameters and random streams; use the
    engines as usual for everything else (push, collect_step, get_params ...)."""

    def __init__(self, engines):
        self.engines = list(engines)
        assert self.engines, "a group needs at least one engine"
        self.lib = self.engines[0].lib
        k = len(self.engines)
        arr = (C.c_void_p * k)(*[e._ctx for e in self.engines])
        self._g = C.c_void_p()
        _lib.check(self.lib.fsrl_group_create(arr, k, C.byref(self._g)))

    def close(self):
        if getattr(self, "_g", None) is not None and self._g:
            self.lib.fsrl_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_plan(self, tall_tiles=-1):
        """32-row tiles per (member, network) in the forward / backward launch: -1 automatic, 0 none, n > 0 a count (A/B; same bits)."""
        _lib.check(self.lib.fsrl_group_set_plan(self._g, int(tall_tiles)))

    def ppo_update(self, lagrangians, rescalings, batch_size, repeat, perms=None, seed=0):
        """k x Engine.ppo_update.  lagrangians: [k][n_critics - 1]; rescalings: [k]; perms: None or per member a list /
        array [repeat][N_i].  -> (list of stats arrays [steps_i, 11], list of stopped passes (-1 = none))."""
        k = len(self.engines)
        lag = np.ascontiguousarray(lagrangians, np.float64).reshape(k, -1)
        resc = np.ascontiguousarray(rescalings, np.float64).reshape(k)
        sizes = [len(e) for e in self.engines]
        cap = max(1, max(-(-n // max(batch_size, 1)) for n in sizes)) * max(repeat, 1)
        stats = [np.empty((cap, _lib.PPO_NSTATS), np.float32) for _ in range(k)]
        sp = (_f32p * k)(*[_ptr(s, _f32p) for s in stats])
        pp, keep = None, []
        if perms is not None:
            for i in range(k):
                a = np.ascontiguousarray(np.stack([np.asarray(p, np.int64) for p in perms[i]]), np.int64)
                assert a.shape == (repeat, sizes[i]), "perms[i] must be [repeat][N_i]"
                keep.append(a)
            pp = (_i64p * k)(*[_ptr(a, _i64p) for a in keep])
        nst = np.zeros(k, np.int64)
        stop = np.zeros(k, np.int32)
        _lib.check(self.lib.fsrl_group_ppo_update(self._g, _ptr(lag, _f64p) if lag.size else None, _ptr(resc, _f64p),
                                                  int(batch_size), int(repeat), pp, int(seed), sp, cap, _ptr(nst, _i64p),
                                                  _ptr(stop, _i32p)))
        for e, n in zip(self.engines, sizes):
            e._n = n
        return [s[:int(n)] for s, n in zip(stats, nst)], [int(x) for x in stop]
