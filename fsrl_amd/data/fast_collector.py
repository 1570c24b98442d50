"""Episode-count-exact vectorised rollout collection on the host CPUs.

Behaviour of fsrl/data/fast_collector.py:192-408 (policy forward under no_grad -> exploration
noise -> map_action -> env.step(ids) -> cost from info -> buffer.add -> per-env reset, surplus
envs dropped so exactly n_episode episodes are collected, statistics dict), written against
numpy dicts instead of tianshou Batches.  The buffer is the HIP-resident store proxy, so every
vector step costs one staged push (no device sync)."""
import time
from typing import Any, Dict, Optional

import numpy as np
import torch

from fsrl_amd.data.batch import Batch


class FastCollector:
    def __init__(self, policy, env, buffer=None, preprocess_fn=None, exploration_noise: bool = False,
                 device_actor: bool = False, fused_step: bool = True, split_phase=False, native_loop: bool = True,
                 resident_actor: Optional[bool] = None):
        # device_actor=True: actions come from fsrl_actor_sample (actor on the MI355X, library RNG) and rows go
        # straight to fsrl_store_push -- no torch call and no Batch objects per vector step.  False keeps the
        # host mirror of the actor with torch's random stream (what the reference consumes).
        self.device_actor = device_actor and getattr(policy, "engine", None) is not None
        # resident_actor (device_actor only; None = the library's default, on): the actor kernel stays on the device for the length of a
        # collect and is rung through a doorbell in pinned memory instead of being launched once per vector step
        # (fsrl_actor_set_resident, include/fsrl_hip.h).  Same actions either way.
        if self.device_actor and resident_actor is not None:
            policy.engine.actor_set_resident(bool(resident_actor))
        # fused_step (device_actor only): one fsrl_collect_step per vector step instead of fsrl_actor_sample + fsrl_store_push
        self.fused_step = fused_step
        # native_loop (device_actor + fused_step over the worker-process env).  True: a collect(n_episode) is ONE library call
        # (fsrl_collect_episodes: handshake with the env workers, store, actor, resets, episode accounting).  "run": the vector
        # steps in which no episode ends run inside the library (fsrl_collect_run), Python sees the episode boundaries.  False:
        # one fsrl_collect_step per vector step from Python.
        self.native_loop = native_loop
        # split_phase (device_actor + fused_step over an env with step_async / step_wait and two lanes: the worker-process env):
        # the envs are stepped in two halves, the actor for one half runs while the other half's workers step.  Rows of an
        # env stay chronological, so sample(0) order and episode-exact collection are as in the plain loop; the ORDER in which
        # the library's noise stream is consumed changes, so this is a throughput mode like perms=None, not a parity mode.
        # split_phase="auto": the first collect runs the plain loop and times its two halves; the split loop takes over when
        # a vector step of the env costs more than the actor call it would hide (an env that costs nothing gains nothing
        # from two half-width actor launches per vector step).
        from fsrl_amd.env.venv import as_vector_env
        env = as_vector_env(env)             # a single env is wrapped like the reference does (fast_collector.py:55-58)
        can_split = hasattr(env, "step_async") and getattr(env, "n_lanes", 1) == 2
        self._split_auto = split_phase == "auto" and can_split
        self.split_phase = bool(split_phase is True and can_split)
        self._t_env = self._t_act = 0.0
        self._split_decided = False
        self.env = env
        self.env_num = len(env)
        self.policy = policy
        self.buffer = buffer
        self.preprocess_fn = preprocess_fn
        self.exploration_noise = exploration_noise
        self._action_space = env.action_space
        if buffer is not None:
            assert buffer.buffer_num >= self.env_num
        self.reset(False)

    # ------------------------------------------------------------------ resets
    def reset(self, reset_buffer: bool = True, gym_reset_kwargs: Optional[Dict[str, Any]] = None) -> None:
        self.reset_env(gym_reset_kwargs)
        if reset_buffer:
            self.reset_buffer()
        self.reset_stat()

    def reset_stat(self) -> None:
        self.collect_step, self.collect_episode, self.collect_time = 0, 0, 0.0

    def reset_buffer(self, keep_statistics: bool = False) -> None:
        if self.buffer is not None:
            self.buffer.reset(keep_statistics=keep_statistics)

    def reset_env(self, gym_reset_kwargs: Optional[Dict[str, Any]] = None) -> None:
        obs, info = self.env.reset(**(gym_reset_kwargs or {}))
        self._obs = np.asarray(obs)

    # ------------------------------------------------------------------ collect
    def collect(self, n_episode: int = 1, random: bool = False, render: bool = False,
                no_grad: bool = True, gym_reset_kwargs: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
        """fsrl/data/fast_collector.py:252-368.  With the actor on the device the resident actor kernel (fsrl_actor_set_resident) lives
        for the length of this call: the native loops end it themselves, the interpreted ones through fsrl_actor_release here."""
        try:
            return self._collect(n_episode, random, render, no_grad, gym_reset_kwargs)
        finally:
            if self.device_actor:
                self.policy.engine.actor_release()

    def _collect(self, n_episode: int = 1, random: bool = False, render: bool = False,
                 no_grad: bool = True, gym_reset_kwargs: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
        if n_episode is None:
            raise TypeError("Please specify n_episode in FastCollector.collect().")
        assert n_episode > 0
        ready = np.arange(min(self.env_num, n_episode))
        obs = self._obs[:len(ready)]
        t0 = time.time()
        step_count, total_cost, episode_count = 0, 0.0, 0
        term_count, trunc_count = 0, 0
        ep_rews, ep_lens = [], []
        eng = self.policy.engine if self.device_actor else None
        if eng is not None and hasattr(self.policy, "_drain"):
            self.policy._drain()
        if eng is not None and not random and self.buffer is not None and self.fused_step:
            if self.split_phase:
                return self._collect_split(eng, n_episode, ready, obs, t0, gym_reset_kwargs)
            return self._collect_fused(eng, n_episode, ready, obs, t0, gym_reset_kwargs)
        while True:
            data = None if eng is not None and not random else Batch(obs=obs, info={})
            if eng is not None and not random:
                act = eng.actor_sample(obs, deterministic=self.policy._deterministic_eval and not self.policy.training)
            elif random:
                act = np.stack([self._action_space.sample() for _ in ready])
                act = self.policy.map_action_inverse(act)
            else:
                with torch.no_grad():
                    result = self.policy(data, None)
                act = np.asarray(result.act.numpy() if torch.is_tensor(result.act) else result.act)
                if self.exploration_noise:
                    act = self.policy.exploration_noise(act, data)
            obs_next, rew, terminated, truncated, info = self.env.step(self.policy.map_action(act), ready)
            terminated, truncated = np.asarray(terminated, bool), np.asarray(truncated, bool)
            done = terminated | truncated
            cost = np.asarray(info.get("cost", np.zeros(len(ready))), np.float64) if isinstance(info, dict) \
                else np.array([i.get("cost", 0.0) for i in info], np.float64)
            total_cost += float(cost.sum())
            step_count += len(ready)
            if self.buffer is not None and eng is not None:
                ptr, ep_rew, ep_len, ep_idx = eng.push(ready, obs, act, rew, cost, terminated, truncated, obs_next)
                self.buffer.note_add(ready)
            elif self.buffer is not None:
                ptr, ep_rew, ep_len, ep_idx = self.buffer.add(
                    Batch(obs=obs, act=act, rew=np.asarray(rew, np.float64), cost=cost,
                          terminated=terminated, truncated=truncated, obs_next=obs_next), buffer_ids=ready)
            else:
                ep_rew, ep_len = self._track_episodes(ready, rew, done)
            obs = np.asarray(obs_next).copy()
            if done.any():
                local = np.where(done)[0]
                episode_count += len(local)
                ep_lens.append(np.asarray(ep_len)[local])
                ep_rews.append(np.asarray(ep_rew)[local])
                term_count += int(terminated.sum())
                trunc_count += int(truncated.sum())
                obs_reset, _ = self.env.reset(ready[local], **(gym_reset_kwargs or {}))
                obs[local] = obs_reset
                surplus = len(ready) - (n_episode - episode_count)
                if surplus > 0:  # drop finished envs that are no longer needed (unbiased tail)
                    mask = np.ones(len(ready), bool)
                    mask[local[:surplus]] = False
                    ready, obs = ready[mask], obs[mask]
            if episode_count >= n_episode:
                break
        self.collect_step += step_count
        self.collect_episode += episode_count
        self.collect_time += max(time.time() - t0, 1e-9)
        self.reset_env()
        rews, lens = np.concatenate(ep_rews), np.concatenate(ep_lens)
        done_count = term_count + trunc_count
        return {"n/ep": episode_count, "n/st": step_count, "rew": float(rews.mean()),
                "len": float(lens.mean()), "total_cost": total_cost,
                "cost": total_cost / episode_count, "truncated": trunc_count / done_count,
                "terminated": term_count / done_count}

    def _collect_fused(self, eng, n_episode, ready, obs, t0, gym_reset_kwargs):
        """The device-actor loop on fsrl_collect_step: per vector step ONE library call (store the finished transitions
        while the actor for the next observations is in flight, then noise + map_action) and env.step."""
        pol = self.policy
        det = bool(pol._deterministic_eval and not pol.training)
        space = pol.action_space
        bound = {"": 0, "clip": 1, "tanh": 2}[pol.action_bound_method] if space is not None else 0
        low = np.asarray(space.low, np.float32) if (space is not None and pol.action_scaling) else None
        high = np.asarray(space.high, np.float32) if low is not None else None
        step_count, total_cost, episode_count, term_count, trunc_count = 0, 0.0, 0, 0, 0
        ep_rews, ep_lens = [], []
        obs = np.asarray(obs, np.float32)
        if self.native_loop is True and hasattr(self.env, "native_desc") and not gym_reset_kwargs:
            return self._collect_native(eng, n_episode, ready, obs, t0, det, bound, low, high, split=False)
        act, env_act, _, _ = eng.collect_step(None, obs, det, bound, low, high)
        clock = time.perf_counter
        t_env = t_act = 0.0
        # worker-process env: the steps in which no episode ends run inside the library (fsrl_collect_run); this loop sees the
        # boundary steps only
        native = self.native_loop and hasattr(self.env, "native_desc")
        # split_phase="auto" without a native loop: time 16 interpreted steps first.  With the native loop there is nothing to
        # decide: it beats the interpreted split loop in every measured configuration (4 / 32 workers x 0 / 100 us)
        probing = self._split_auto and not self._split_decided and not native
        probe_left = 16
        while True:
            if probing and probe_left == 0:
                probing, self._split_decided = False, True
                self._t_env, self._t_act = t_env, t_act
                self.split_phase = t_env > 1.25 * t_act           # takes over from the next collect() on
            if native and not probing:
                try:
                    k_steps, csum, obs, act, rew, cost, terminated, truncated, obs_next = eng.collect_run(
                        self.env.native_desc(), ready, obs, act, env_act, det, bound, low, high)
                except BaseException:
                    # as in _collect_native: a failed native run may have left a command half-handled on a lane (dead worker,
                    # HIP error after a posted command) -- the env refuses every later command instead of answering with stale data
                    self.env.sync_native()
                    self.env.mark_broken("a native collect run failed while env commands were in flight")
                    raise
                self.env.sync_native()
                done = terminated | truncated
                total_cost += csum
                step_count += len(ready) * k_steps
            else:
                tc0 = clock()
                obs_next, rew, terminated, truncated, info = self.env.step(env_act, ready)
                tc1 = clock()
                t_env += tc1 - tc0
                terminated, truncated = np.asarray(terminated, bool), np.asarray(truncated, bool)
                done = terminated | truncated
                cost = np.asarray(info.get("cost", np.zeros(len(ready))), np.float64) if isinstance(info, dict) \
                    else np.array([i.get("cost", 0.0) for i in info], np.float64)
                total_cost += float(cost.sum())
                step_count += len(ready)
                probe_left -= 1
            prev = (ready, obs, act, rew, cost, terminated, truncated, obs_next)
            nxt, nready = obs_next, ready
            if done.any():
                local = np.where(done)[0]
                n_done = len(local)
                term_count += int(terminated.sum()); trunc_count += int(truncated.sum())
                nxt = np.array(obs_next, np.float32)
                obs_reset, _ = self.env.reset(ready[local], **(gym_reset_kwargs or {}))
                nxt[local] = obs_reset
                surplus = len(ready) - (n_episode - episode_count - n_done)
                if surplus > 0:      # drop finished envs that are no longer needed (unbiased tail)
                    mask = np.ones(len(ready), bool)
                    mask[local[:surplus]] = False
                    nready, nxt = ready[mask], nxt[mask]
                last = episode_count + n_done >= n_episode
                tc2 = clock()
                act, env_act, ep_rew, ep_len = eng.collect_step(prev, None if last else nxt, det, bound, low, high)
                t_act += clock() - tc2
                episode_count += n_done
                ep_lens.append(ep_len[local].copy()); ep_rews.append(ep_rew[local].copy())
                if last:
                    break
            else:
                tc2 = clock()
                act, env_act, _, _ = eng.collect_step(prev, nxt, det, bound, low, high)
                t_act += clock() - tc2
            obs, ready = np.asarray(nxt, np.float32), nready
        self.buffer.sync_sizes()
        self.collect_step += step_count
        self.collect_episode += episode_count
        self.collect_time += max(time.time() - t0, 1e-9)
        self.reset_env()
        rews, lens = np.concatenate(ep_rews), np.concatenate(ep_lens)
        done_count = term_count + trunc_count
        return {"n/ep": episode_count, "n/st": step_count, "rew": float(rews.mean()),
                "len": float(lens.mean()), "total_cost": total_cost,
                "cost": total_cost / episode_count, "truncated": trunc_count / done_count,
                "terminated": term_count / done_count}

    def _collect_native(self, eng, n_episode, ready, obs, t0, det, bound, low, high, split):
        """worker-process env: the WHOLE collect -- steps, store, actor, resets, episode accounting, surplus envs -- in one library
        call (fsrl_collect_episodes, or its two-lane split-phase form).  The interpreter is not entered while the call runs: a
        KeyboardInterrupt is delivered when the collect returns (a collect of n_episode episodes; native_loop="run" or False
        returns to Python at every episode boundary / vector step instead).  The env's generation counters travel through the call:
        they are read back even when it fails, so the env object stays consistent with its shared block (a worker that raised
        or died fails the call within half a second; the env itself is then unusable and says so on its next command)."""
        try:
            r = eng.collect_episodes(self.env.native_desc(), ready, obs, n_episode, det, bound, low, high, split=split)
        except BaseException:
            # a failed native collect may have left a command half-handled on a lane: the env refuses every later command
            self.env.sync_native()
            self.env.mark_broken("a native collect failed while env commands were in flight")
            raise
        self.env.sync_native()
        self.buffer.sync_sizes()
        self.collect_step += r["steps"]
        self.collect_episode += len(r["ep_rews"])
        self.collect_time += max(time.time() - t0, 1e-9)
        if self._split_auto and not self._split_decided:
            # the first collect ran single-phase and timed its two halves in C: the split loop takes over when the env's vector
            # step outweighs the store + actor call it can hide behind the other lane's step
            self._split_decided = True
            self._t_env, self._t_act = r["t_env"], r["t_act"]
            self.split_phase = r["t_env"] > 1.25 * r["t_act"]
        self.reset_env()
        done_count = r["terminated"] + r["truncated"]
        return {"n/ep": len(r["ep_rews"]), "n/st": r["steps"], "rew": float(r["ep_rews"].mean()),
                "len": float(r["ep_lens"].mean()), "total_cost": r["total_cost"],
                "cost": r["total_cost"] / len(r["ep_rews"]), "truncated": r["truncated"] / done_count,
                "terminated": r["terminated"] / done_count}

    def _collect_split(self, eng, n_episode, ready, obs, t0, gym_reset_kwargs):
        """The fused loop over the two lanes of a worker-process env: while lane A's workers step, the collector stores lane
        B's finished transitions, evaluates the actor on lane B's next observations and starts lane B's step.  Episode
        accounting is global (envs of either lane are dropped once the remaining envs cover the episodes still needed), so
        exactly n_episode episodes are collected, as in the reference (fast_collector.py:341-362)."""
        pol, env = self.policy, self.env
        det = bool(pol._deterministic_eval and not pol.training)
        space = pol.action_space
        bound = {"": 0, "clip": 1, "tanh": 2}[pol.action_bound_method] if space is not None else 0
        low = np.asarray(space.low, np.float32) if (space is not None and pol.action_scaling) else None
        high = np.asarray(space.high, np.float32) if low is not None else None
        if self.native_loop is True and hasattr(env, "native_desc") and not gym_reset_kwargs:
            return self._collect_native(eng, n_episode, ready, np.asarray(obs, np.float32), t0, det, bound, low, high, split=True)
        step_count, total_cost, episode_count, term_count, trunc_count = 0, 0.0, 0, 0, 0
        ep_rews, ep_lens = [], []
        obs = np.asarray(obs, np.float32)
        lane_of = env.lane_of_env
        groups = []
        for lane in range(2):
            m = lane_of[ready] == lane
            g = {"ready": ready[m], "obs": obs[m], "act": None, "busy": False}
            if len(g["ready"]):
                g["act"], env_act, _, _ = eng.collect_step(None, g["obs"], det, bound, low, high)
                env.step_async(env_act, g["ready"])
                g["busy"] = True
            groups.append(g)
        while episode_count < n_episode:
            assert groups[0]["busy"] or groups[1]["busy"], "split-phase collect: no env left but episodes are missing"
            for g in groups:
                if not g["busy"]:
                    continue
                rdy = g["ready"]
                obs_next, rew, terminated, truncated, info = env.step_wait(rdy)
                g["busy"] = False
                done = terminated | truncated
                cost = np.asarray(info["cost"], np.float64)
                total_cost += float(cost.sum())
                step_count += len(rdy)
                prev = (rdy, g["obs"], g["act"], rew, cost, terminated, truncated, obs_next)
                nxt, nready = obs_next, rdy
                if done.any():
                    local = np.where(done)[0]
                    n_done = len(local)
                    term_count += int(terminated.sum()); trunc_count += int(truncated.sum())
                    nxt = np.array(obs_next, np.float32)
                    obs_reset, _ = env.reset(rdy[local], **(gym_reset_kwargs or {}))
                    nxt[local] = obs_reset
                    active = len(groups[0]["ready"]) + len(groups[1]["ready"])
                    surplus = active - (n_episode - episode_count - n_done)
                    if surplus > 0:      # drop finished envs that are no longer needed (unbiased tail)
                        mask = np.ones(len(rdy), bool)
                        mask[local[:surplus]] = False
                        nready, nxt = rdy[mask], nxt[mask]
                    act, env_act, ep_rew, ep_len = eng.collect_step(prev, nxt if len(nready) else None, det, bound, low, high)
                    episode_count += n_done
                    ep_lens.append(ep_len[local].copy()); ep_rews.append(ep_rew[local].copy())
                else:
                    act, env_act, _, _ = eng.collect_step(prev, nxt, det, bound, low, high)
                g["ready"], g["obs"], g["act"] = nready, np.asarray(nxt, np.float32), act
                if len(nready):
                    env.step_async(env_act, nready)
                    g["busy"] = True
        assert not (groups[0]["busy"] or groups[1]["busy"])
        self.buffer.sync_sizes()
        self.collect_step += step_count
        self.collect_episode += episode_count
        self.collect_time += max(time.time() - t0, 1e-9)
        self.reset_env()
        rews, lens = np.concatenate(ep_rews), np.concatenate(ep_lens)
        done_count = term_count + trunc_count
        return {"n/ep": episode_count, "n/st": step_count, "rew": float(rews.mean()),
                "len": float(lens.mean()), "total_cost": total_cost,
                "cost": total_cost / episode_count, "truncated": trunc_count / done_count,
                "terminated": term_count / done_count}

    # episode bookkeeping when no buffer is attached (evaluation)
    def _track_episodes(self, ready, rew, done):
        if not hasattr(self, "_acc_rew") or len(self._acc_rew) != self.env_num:
            self._acc_rew, self._acc_len = np.zeros(self.env_num), np.zeros(self.env_num, int)
        self._acc_rew[ready] += np.asarray(rew, np.float64)
        self._acc_len[ready] += 1
        ep_rew, ep_len = np.zeros(len(ready)), np.zeros(len(ready), int)
        d = np.where(done)[0]
        ep_rew[d], ep_len[d] = self._acc_rew[ready[d]], self._acc_len[ready[d]]
        self._acc_rew[ready[d]], self._acc_len[ready[d]] = 0.0, 0
        return ep_rew, ep_len
