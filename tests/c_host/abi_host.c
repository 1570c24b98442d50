/* A host of the C ABI with no Python and no torch in the process: plain C, compiled with gcc against include/fsrl_hip.h and
 * linked to libfsrl_hip.so only.  It does what a reference-side binding does for one PPO-Lagrangian update()
 * (fsrl/policy/base_policy.py:332-355): create a context, fill the store in lock step like FastCollector
 * (fast_collector.py:333), run the update with the library's own permutation stream, read parameters and statistics back.
 * Output: one line of hexadecimal checksums (FNV-1a over the raw bytes) that tests/test_gpu_c_host.py compares with the
 * same call sequence issued from Python through ctypes -- bit for bit.
 *
 *   gcc -O2 -std=c99 -Iinclude tests/c_host/abi_host.c -Lfsrl_amd -lfsrl_hip -Wl,-rpath,$PWD/fsrl_amd -lm -o abi_host
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fsrl_hip.h"

#define CHK(call)                                                                 \
    do {                                                                          \
        int rc_ = (call);                                                         \
        if (rc_ != 0) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, fsrl_last_error());     \
            return 1;                                                             \
        }                                                                         \
    } while (0)

static uint64_t lcg_state = 88172645463325252ULL;
static double lcg_uniform(void) { /* xorshift64*: the Python side of the test restates these three lines */
    lcg_state ^= lcg_state >> 12; lcg_state ^= lcg_state << 25; lcg_state ^= lcg_state >> 27;
    return (double)((lcg_state * 2685821657736338717ULL) >> 11) / 9007199254740992.0;
}
static uint64_t fnv1a(const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ULL; }
    return h;
}

int main(void) {
    enum { E = 4, T = 150, DO = 8, DA = 2, H = 64, B = 64, REPEAT = 3 };
    fsrl_config cfg;
    fsrl_config_default(&cfg);
    cfg.obs_dim = DO; cfg.act_dim = DA; cfg.hidden = H; cfg.n_critics = 2; cfg.env_num = E; cfg.buffer_size = 4000;
    cfg.max_grad_norm = 0.5f; cfg.target_kl = 0.0f; /* KL early stop off */
    fsrl_ctx* ctx = NULL;
    CHK(fsrl_ctx_create(0, &cfg, &ctx));
    const int64_t np = fsrl_param_count(ctx);
    float* theta = (float*)malloc((size_t)np * sizeof(float));
    for (int64_t i = 0; i < np; ++i) theta[i] = (float)((lcg_uniform() - 0.5) * 0.2);
    CHK(fsrl_params_set(ctx, theta, np));

    /* lock-step rollout: every env finishes an episode (truncated) every 50 steps, env 3 terminates at step 120 */
    int32_t ids[E];
    float obs[E * DO], nxt[E * DO], act[E * DA];
    double rew[E], cost[E];
    uint8_t term[E], trunc[E];
    int64_t ptr[E], ep_idx[E];
    double ep_rew[E];
    int32_t ep_len[E];
    for (int e = 0; e < E; ++e) ids[e] = e;
    for (int i = 0; i < E * DO; ++i) obs[i] = (float)(lcg_uniform() * 2.0 - 1.0);
    for (int t = 0; t < T; ++t) {
        for (int i = 0; i < E * DO; ++i) nxt[i] = (float)(lcg_uniform() * 2.0 - 1.0);
        for (int i = 0; i < E * DA; ++i) act[i] = (float)(lcg_uniform() * 0.6 - 0.3);
        for (int e = 0; e < E; ++e) {
            rew[e] = lcg_uniform();
            cost[e] = lcg_uniform() < 0.1 ? 1.0 : 0.0;
            term[e] = (uint8_t)(e == 3 && t == 120);
            trunc[e] = (uint8_t)((t + 1) % 50 == 0 && !term[e]);
        }
        CHK(fsrl_store_push(ctx, ids, E, obs, act, rew, cost, term, trunc, nxt, ptr, ep_rew, ep_len, ep_idx));
        memcpy(obs, nxt, sizeof(obs));
    }
    if (fsrl_store_len(ctx) != (int64_t)E * T) { fprintf(stderr, "store holds %lld rows\n", (long long)fsrl_store_len(ctx)); return 1; }

    const double lag[1] = {0.75};
    float stats[64 * FSRL_PPO_NSTATS];
    int64_t n_steps = 0;
    int32_t stopped = -2;
    CHK(fsrl_ppo_update(ctx, lag, 1.0 / 1.75, B, REPEAT, NULL, 12345ULL, stats, 64, &n_steps, &stopped));
    CHK(fsrl_params_get(ctx, theta, np));
    for (int64_t i = 0; i < np; ++i)
        if (!isfinite(theta[i])) { fprintf(stderr, "non-finite parameter %lld\n", (long long)i); return 1; }
    int32_t rank = -1, world = -1;
    CHK(fsrl_comm_info(ctx, &rank, &world));
    printf("params %lld steps %lld stopped %d theta %016llx stats %016llx kl_last %.9g rank %d world %d\n", (long long)np,
           (long long)n_steps, (int)stopped, (unsigned long long)fnv1a(theta, (size_t)np * sizeof(float)),
           (unsigned long long)fnv1a(stats, (size_t)n_steps * FSRL_PPO_NSTATS * sizeof(float)),
           (double)stats[(n_steps - 1) * FSRL_PPO_NSTATS + 5], (int)rank, (int)world);
    CHK(fsrl_ctx_destroy(ctx));
    free(theta);
    return 0;
}
