/*
 * fsrl_hip.h -- C ABI of libfsrl_hip.so, the MI355X (gfx950) policy-update engine that sits
 * behind FSRL's Python seam (fsrl.agent / fsrl.policy / fsrl.trainer).
 *
 * The reference (liuzuxin/FSRL) is pure Python and has no FFI; its seam is the duck-typed
 * contract between trainer/collector and policy+buffer.  Each entry
 * point below names the reference interface it replaces (file:line under the reference
 * tree).  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative code on failure (FSRL_E*);
 *     fsrl_last_error() returns a thread-local human-readable message.
 *   - all pointer arguments are HOST pointers owned by the caller for the duration of
 *     the call; the library owns every device allocation (store, parameters, Adam
 *     moments, activations) and copies in/out.
 *   - one context per GPU, not re-entrant; single caller thread (like the reference).
 *   - parameters travel as ONE flat float32 vector in torch `parameters()` order:
 *       actor  : sigma_param[Da] W1[H1,Do] b1[H1] W2[H2,H1] b2[H2] Wmu[Da,H2] bmu[Da]
 *       critic : W1[H1,Do] b1[H1] W2[H2,H1] b2[H2] W3[1,H2] b3[1]     (x n_critics)
 *     (tianshou-0.5 Net+ActorProb / Net+Critic as built at
 *      fsrl/agent/ppo_lag_agent.py:136-153; weights row-major [out,in] like nn.Linear).
 */
#ifndef FSRL_HIP_H
#define FSRL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSRL_OK 0
#define FSRL_EINVAL (-22)   /* shape / flag violation (reference: assert / TypeError)  */
#define FSRL_ENOMEM (-12)
#define FSRL_EHIP (-5)      /* a HIP runtime call failed; see fsrl_last_error()        */
#define FSRL_ESTATE (-1)    /* call order violation (e.g. pass before begin)           */

#define FSRL_ALGO_PPO_LAG 0
#define FSRL_ALGO_TRPO_LAG 1
#define FSRL_ALGO_CPO 2
#define FSRL_ALGO_SAC_LAG 3

#define FSRL_MAX_CRITICS 4
#define FSRL_PPO_NSTATS 11  /* rescaling, lagrangian, actor_safety, actor_rew, actor_total,
                               kl, vf0, vf1, vf_total, total, entropy  (logger keys of
                               fsrl/policy/ppo_lag.py:169-170,204-211,245-247 and
                               fsrl/policy/lagrangian_base.py:158-165)                  */

typedef struct fsrl_ctx fsrl_ctx;

/* Hyper-parameters; field names follow PPOLagAgent.__init__ (fsrl/agent/ppo_lag_agent.py:82-116)
 * and PPOLagrangian.__init__ (fsrl/policy/ppo_lag.py:85-133). */
typedef struct fsrl_config {
    int32_t algo;            /* FSRL_ALGO_*                                              */
    int32_t obs_dim;         /* Do  (1..128)                                             */
    int32_t act_dim;         /* Da  (1..16)                                              */
    int32_t hidden;          /* H1 == H2 == hidden, one of 64 / 128 / 256                */
    int32_t n_critics;       /* 1 + number of cost constraints (2 for one cost)          */
    int32_t env_num;         /* number of per-env sub-buffers (VectorReplayBuffer)       */
    int64_t buffer_size;     /* total rows requested; sub-buffers get ceil(total/env_num)*/
    float max_action;        /* env.action_space.high[0]                                 */
    double gamma;            /* 0.99                                                     */
    double gae_lambda;       /* 0.95                                                     */
    float eps_clip;          /* 0.2                                                      */
    float dual_clip;         /* 0 = off, else > 1                                        */
    float vf_coef;           /* 0.25                                                     */
    float max_grad_norm;     /* 0 = off (agent default None), cfg default 0.5            */
    float target_kl;         /* 0.02; early stop when pass-mean KL > 1.5*target_kl       */
    int32_t norm_adv;        /* per-minibatch advantage normalisation                    */
    int32_t use_lagrangian;
    float lr;                /* Adam, one optimiser over actor+critics                   */
    float beta1, beta2, adam_eps;
} fsrl_config;

const char* fsrl_last_error(void);
void fsrl_config_default(fsrl_config* cfg); /* reference defaults for PPO-Lag            */

/* ---- context ------------------------------------------------------------------------ */
/* replaces: PPOLagAgent.__init__ building nets + optimiser + policy on `device`
 * (fsrl/agent/ppo_lag_agent.py:127-200). */
int fsrl_ctx_create(int device_id, const fsrl_config* cfg, fsrl_ctx** out);
int fsrl_ctx_destroy(fsrl_ctx* ctx);
int fsrl_sync(fsrl_ctx* ctx);                       /* drain both internal streams       */

/* ---- parameters  (policy.state_dict()/load_state_dict(), base_agent.py:293-297) ------ */
int64_t fsrl_param_count(const fsrl_ctx* ctx);
int fsrl_params_set(fsrl_ctx* ctx, const float* flat, int64_t n);
int fsrl_params_get(fsrl_ctx* ctx, float* flat, int64_t n);
int fsrl_grads_get(fsrl_ctx* ctx, float* flat, int64_t n);   /* last minibatch gradient  */
int fsrl_optim_reset(fsrl_ctx* ctx);                /* zero Adam moments and step count  */

/* ---- HIP-resident transition store (tianshou VectorReplayBuffer as used at
 *      fsrl/agent/base_agent.py:279, fsrl/data/fast_collector.py:333-335,
 *      fsrl/trainer/onpolicy.py:109) ------------------------------------------------- */
/* add(batch, buffer_ids) -> (ptr, ep_rew, ep_len, ep_idx).  Rows are staged in pinned host
 * memory and copied with hipMemcpyAsync on the side stream; the call does not wait.       */
int fsrl_store_push(fsrl_ctx* ctx, const int32_t* env_ids, int32_t k, const float* obs,
                    const float* act, const double* rew, const double* cost,
                    const uint8_t* terminated, const uint8_t* truncated,
                    const float* obs_next, int64_t* ptr_out, double* ep_rew_out,
                    int32_t* ep_len_out, int64_t* ep_idx_out);
int fsrl_store_reset(fsrl_ctx* ctx, int keep_statistics);   /* buffer.reset()            */
int64_t fsrl_store_len(const fsrl_ctx* ctx);                /* len(buffer)               */
/* buffer.sample_indices(0): env-major, chronological inside each sub-buffer.             */
int fsrl_store_sample0(fsrl_ctx* ctx, int64_t* indices_out, int64_t cap, int64_t* n_out);

/* ---- actor inference for the collector (policy.forward under no_grad,
 *      fsrl/policy/base_policy.py:178-190; sampling stays on the host RNG) ------------ */
int fsrl_actor_forward(fsrl_ctx* ctx, const float* obs, int32_t k, float* mu_out,
                       float* sigma_out);

/* ---- PPO-Lagrangian update = BasePolicy.update (base_policy.py:332-355) -------------- */
/* begin: buffer.sample(0) + PPOLagrangian.process_fn (ppo_lag.py:134-150): gathers the
 * batch in sample(0) order, V_i(obs), V_i(obs_next), float64 GAE per critic, logp_old.
 * lagrangians[n_critics-1] and rescaling come from LagrangianPolicy (lagrangian_base.py:
 * 145-166) and are host float64 like in the reference.  *n_out = rows in the batch.      */
int fsrl_ppo_begin(fsrl_ctx* ctx, const double* lagrangians, double rescaling,
                   int32_t batch_size, int64_t* n_out);
/* one pass of PPOLagrangian.learn's outer loop (ppo_lag.py:217-255): minibatches of
 * batch_size in the order `perm` (= np.random.permutation(n), merge_last=True); fwd, loss,
 * bwd, grad clip, Adam for each; *stopped_out = 1 if the pass-mean approx-KL exceeded
 * 1.5*target_kl (the caller must then stop calling, like the reference's break).
 * perm == NULL: the library draws its own permutation (xoshiro256**, seeded by `seed`).  */
int fsrl_ppo_pass(fsrl_ctx* ctx, const int64_t* perm, uint64_t seed, int32_t* stopped_out);
/* end: drains per-minibatch stats ([n_steps][FSRL_PPO_NSTATS] float32, row-major).       */
int fsrl_ppo_end(fsrl_ctx* ctx, float* stats_out, int64_t cap_steps, int64_t* n_steps_out);
/* convenience: begin + `repeat` passes (perms = [repeat][n] or NULL) + end.              */
int fsrl_ppo_update(fsrl_ctx* ctx, const double* lagrangians, double rescaling,
                    int32_t batch_size, int32_t repeat, const int64_t* perms, uint64_t seed,
                    float* stats_out, int64_t cap_steps, int64_t* n_steps_out,
                    int32_t* stopped_pass_out);

/* read back process_fn products of the current batch: which = "values" | "rets" | "advs"
 * ([n][n_critics] float32, like torch.stack(.., -1)) | "logp_old" ([n]).                 */
int fsrl_batch_get(fsrl_ctx* ctx, const char* which, float* out, int64_t cap);

/* ---- stand-alone float64 GAE scan on the device (gae_return, base_policy.py:524-540);
 *      v_next already masked.  Bit-exact with the sequential reference.                  */
int fsrl_gae_return(fsrl_ctx* ctx, const float* v, const float* v_next, const double* rew,
                    const uint8_t* end_flag, int64_t n, double gamma, double gae_lambda,
                    double* adv_out);

/* ---- timing of the last update, measured with hipEvents on the compute stream --------- */
/* out[0] = process_fn ms, out[1] = learn ms (all passes), out[2] = fused fwd/bwd kernel
 * total ms over the update (sum of per-launch event pairs when profiling is enabled),
 * out[3] = number of fwd/bwd launches, out[4] (if n >= 5) = the raw event-bracket sum (no
 * overhead correction).                                                                   */
int fsrl_set_profiling(fsrl_ctx* ctx, int enable);
int fsrl_last_timing(fsrl_ctx* ctx, double* out, int32_t n);

/* ---- metrics exchange across the node's GPUs is done by the Python host with
 *      torch.distributed (RCCL); the library has no collective of its own.              */

#ifdef __cplusplus
}
#endif
#endif /* FSRL_HIP_H */
