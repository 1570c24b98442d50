/* fsrl_env.c -- the per-step handshake of the worker-process vector env (fsrl_amd/env/shmem.py), native.
 *
 * The reference's training envs are tianshou's ShmemVectorEnv (examples/mlp/train_ppol_agent.py:120-123): one process per
 * env, a pipe send + recv per env and vector step.  Round 2 of this repo used one semaphore pair per worker: 32 posts and 32
 * blocking waits from the Python interpreter per vector step (~130 us).  Here a command is ONE release store of a generation
 * word + one FUTEX_WAKE for everybody waiting on it, and completion is ONE counter the workers decrement; the collector
 * sleeps on it once and is woken by the last worker.  Workers and collector block inside these calls (no interpreter work
 * while waiting); all words live in the env's shared-memory block.  Acquire / release ordering is explicit, so the payload
 * (actions, observations) written with plain stores on either side is visible on any host, not only under x86 TSO.
 *
 * Linux only (futex).  Built by __graft_entry__.build() / fsrl_amd/env/csrc/build.sh into fsrl_amd/libfsrl_env.so.      */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <errno.h>
#include <limits.h>
#include <linux/futex.h>
#include <stdint.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

/* FSRL_ENV_API: empty for libfsrl_env.so (exported); `static` when libfsrl_hip.so's native collector loop
 * (fsrl_amd/csrc/host_collect.inc) includes this file, so that the same code posts and waits from C++ too.  The
 * __atomic builtins are what both gcc (C11) and hipcc (C++17) spell the same way.                                    */
#ifndef FSRL_ENV_API
#define FSRL_ENV_API
#endif

static long fsrl_futex(uint32_t* addr, int op, uint32_t val, const struct timespec* ts) {
    return syscall(SYS_futex, addr, op, val, ts, (void*)0, 0);
}
static inline void cpu_relax(void) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}

/* collector: the command (actions, active mask, per-worker participation stamps) is in place -> publish it.
 * pending := number of workers that take part; gen := new_gen (release); wake every waiter of gen.                 */
FSRL_ENV_API void fsrl_env_post(uint32_t* gen, uint32_t* pending, uint32_t n_workers, uint32_t new_gen) {
    __atomic_store_n(pending, n_workers, __ATOMIC_RELAXED);
    __atomic_store_n(gen, new_gen, __ATOMIC_RELEASE);
    fsrl_futex(gen, FUTEX_WAKE, INT_MAX, (const struct timespec*)0);
}

/* worker: block until *gen != seen; returns the new generation (acquire), or `seen` after timeout_ms without one
 * (the caller checks that its parent is alive and calls again).  spin: polls before sleeping (0 on hosts with fewer CPUs
 * than workers).                                                                                                   */
FSRL_ENV_API uint32_t fsrl_env_wait_go(uint32_t* gen, uint32_t seen, uint32_t spin, int32_t timeout_ms) {
    for (uint32_t i = 0; i < spin; ++i) {
        const uint32_t g = __atomic_load_n(gen, __ATOMIC_ACQUIRE);
        if (g != seen) return g;
        cpu_relax();
    }
    struct timespec ts; ts.tv_sec = timeout_ms / 1000; ts.tv_nsec = (long)(timeout_ms % 1000) * 1000000L;
    for (;;) {
        const uint32_t g = __atomic_load_n(gen, __ATOMIC_ACQUIRE);
        if (g != seen) return g;
        const long rc = fsrl_futex(gen, FUTEX_WAIT, seen, timeout_ms > 0 ? &ts : (const struct timespec*)0);
        if (rc == -1 && errno == ETIMEDOUT) return __atomic_load_n(gen, __ATOMIC_ACQUIRE);
    }
}

/* worker: my part of the command is written -> count down; the last one wakes the collector.                        */
FSRL_ENV_API void fsrl_env_done(uint32_t* pending) {
    if (__atomic_fetch_sub(pending, 1u, __ATOMIC_ACQ_REL) == 1u)
        fsrl_futex(pending, FUTEX_WAKE, 1, (const struct timespec*)0);
}

/* collector: block until *pending == 0 (acquire).  0 = done, -1 = timeout_ms elapsed (worker died?).                 */
FSRL_ENV_API int32_t fsrl_env_wait_done(uint32_t* pending, uint32_t spin, int32_t timeout_ms) {
    for (uint32_t i = 0; i < spin; ++i) {
        if (__atomic_load_n(pending, __ATOMIC_ACQUIRE) == 0u) return 0;
        cpu_relax();
    }
    struct timespec t0, now;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        const uint32_t p = __atomic_load_n(pending, __ATOMIC_ACQUIRE);
        if (p == 0u) return 0;
        struct timespec ts; ts.tv_sec = 0; ts.tv_nsec = 50 * 1000000L;     /* re-check liveness every 50 ms */
        fsrl_futex(pending, FUTEX_WAIT, p, &ts);
        clock_gettime(CLOCK_MONOTONIC, &now);
        const long ms = (now.tv_sec - t0.tv_sec) * 1000L + (now.tv_nsec - t0.tv_nsec) / 1000000L;
        if (timeout_ms > 0 && ms > timeout_ms) return __atomic_load_n(pending, __ATOMIC_ACQUIRE) == 0u ? 0 : -1;
    }
}

/* burn host time until `deadline_ns` on CLOCK_MONOTONIC (the simulated cost of an env step: the worker's numpy dynamics
 * count towards it); returns the current time.  now_ns: the same clock, for the caller to form the deadline.          */
FSRL_ENV_API int64_t fsrl_env_now_ns(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (int64_t)t.tv_sec * 1000000000LL + t.tv_nsec;
}
FSRL_ENV_API int64_t fsrl_env_burn_until(int64_t deadline_ns) {
    int64_t n;
    while ((n = fsrl_env_now_ns()) < deadline_ns) cpu_relax();
    return n;
}
