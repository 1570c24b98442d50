/* fsrl_env.h -- C ABI of libfsrl_env.so: the per-step handshake of the worker-process vector env.
 *
 * What it replaces in the reference: FSRL trains on tianshou's ShmemVectorEnv (examples/mlp/train_ppol_agent.py:120-123;
 * fsrl/agent/base_agent.py:225-242 hands it to FastCollector): one process per env, observations in shared memory, and per
 * env and vector step a `pipe.send(("step", action))` + `pipe.recv()` pair from the collector's interpreter
 * (fsrl/data/fast_collector.py:283-301 calls `self.env.step(action_remap, ready_env_ids)`).  Here the command and its
 * completion are words inside the env's own shared block (fsrl_amd/env/shmem.py lays it out):
 *
 *     gen      one per lane; the collector bumps it to publish a command (actions, active mask and the per-worker
 *              participation stamps are plain stores made BEFORE the bump: release / acquire pair on gen)
 *     pending  one per lane; number of workers that still have to finish the command; the last one wakes the collector
 *
 * All functions are plain C on plain pointers, Linux futex underneath, no HIP.  The same source is compiled into
 * libfsrl_hip.so as static functions for fsrl_collect_run (include/fsrl_hip.h).                                       */
#ifndef FSRL_ENV_H
#define FSRL_ENV_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* collector: pending := n_workers, gen := new_gen (release), FUTEX_WAKE every sleeper of gen                          */
void fsrl_env_post(uint32_t* gen, uint32_t* pending, uint32_t n_workers, uint32_t new_gen);
/* collector: block until *pending == 0 (acquire); `spin` polls before sleeping; 0 = done, -1 = timeout_ms elapsed     */
int32_t fsrl_env_wait_done(uint32_t* pending, uint32_t spin, int32_t timeout_ms);
/* worker: block until *gen != seen; returns the new generation (acquire), or `seen` after timeout_ms without one      */
uint32_t fsrl_env_wait_go(uint32_t* gen, uint32_t seen, uint32_t spin, int32_t timeout_ms);
/* worker: my share of the command is written: count down, the last one wakes the collector                            */
void fsrl_env_done(uint32_t* pending);
/* CLOCK_MONOTONIC in ns / busy-wait until a deadline on it (the simulated cost of an env step)                         */
int64_t fsrl_env_now_ns(void);
int64_t fsrl_env_burn_until(int64_t deadline_ns);

#ifdef __cplusplus
}
#endif
#endif
