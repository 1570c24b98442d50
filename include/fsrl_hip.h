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
#define FSRL_ALGO_FOCOPS 4

#define FSRL_MAX_CRITICS 4
#define FSRL_MAX_HIDDEN 8      /* hidden layers of a layered context (fsrl_config.n_hidden)                          */
#define FSRL_MAX_WIDTH 4096    /* widest hidden layer of a layered context                                            */
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
    int32_t hidden;          /* two hidden layers of this width; with hidden1 / hidden2 set: 0, or the padded width       */
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
    int32_t recompute_adv;   /* PPO-Lag, FOCOPS: recompute V, GAE, returns with the CURRENT critics before every pass after
                                the first (ppo_lag.py:218-221, focops.py:223-226 recompute_advantage); logp_old stays */
    int32_t unbounded;       /* on-policy actor: 1 = mean head without max_action * tanh (ActorProb(unbounded=True),
                                tianshou 0.5 utils/net/continuous.py; fsrl/agent/ppo_lag_agent.py:134)                */
    int32_t rew_norm;        /* reward_normalization (base_policy.py:114, 430-444): critics learn returns divided by the
                                running std of the returns; on-policy contexts                                        */
    int32_t value_clip;      /* PPO: clipped value loss (ppo_lag.py:158-164); needs rew_norm like the reference      */
    int32_t hidden1, hidden2; /* hidden_sizes = (hidden1, hidden2) of the agents (fsrl/agent/ppo_lag_agent.py:91,136), any
                                widths in [1, 256]; 0, 0 = (hidden, hidden).  Every flat parameter vector of the API (set /
                                get, gradients, trust-region vectors, the replay agents' actor / critic vectors) has the
                                caller's layout; the kernels run at 64 / 128 / 256 with zero-padded units               */
    int32_t n_hidden;        /* 0, or the length of hidden_sizes[] below: `hidden_sizes` of the agents as ANY tuple
                                (fsrl/agent/ppo_lag_agent.py:91,136-145; tianshou Net(hidden_sizes=...)), 1 .. FSRL_MAX_HIDDEN
                                layers of 1 .. FSRL_MAX_WIDTH units; leave hidden1 / hidden2 0 with it.  Two layers of at most
                                256 units select the fused kernels exactly like hidden1 / hidden2.  Anything else makes a
                                LAYERED context (every algorithm): the same entry points (store, collector actor,
                                fsrl_ppo_* / fsrl_tr_* / fsrl_sac_* / fsrl_cvpo_*, parameters, snapshot, lr), the same float64 scans and logged rows, but the network
                                math runs one MFMA GEMM launch per Linear (2 L + 5 launches per minibatch step instead of 3)
                                on activations kept in HBM.  Grouped updates and fsrl_launch_floors refuse such a
                                context with FSRL_EINVAL.                                                   */
    int32_t hidden_sizes[FSRL_MAX_HIDDEN];
    int32_t force_layered;   /* tests: run a two-layer network of at most 256 units through the layered kernels as well   */
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
/* Device-resident checkpoint of the training state of an ON-POLICY context (PPO-Lag, FOCOPS, CPO, TRPO-Lag: parameters with
 * their W2 mirrors, Adam moments, step counts, and with reward_normalization the running return statistics), kept in HBM; the
 * collector's noise stream and the library's shuffle stream are NOT part of it.  Replay contexts (SAC / DDPG / CVPO keep actor,
 * Q, target and alpha state of their own) are refused with FSRL_EINVAL. snapshot / restore are device-to-device copies on the compute stream, no host round
 * trip and no synchronisation.  The reference has no counterpart (copy.deepcopy(policy.state_dict()) is the host-side
 * equivalent); bench.py restores the same start state before every timed update with it.                                */
int fsrl_state_snapshot(fsrl_ctx* ctx);
int fsrl_state_restore(fsrl_ctx* ctx);
/* lr_scheduler.step() at the end of BasePolicy.update (fsrl/policy/base_policy.py:352-354): the caller's torch
 * scheduler owns the schedule, this call moves the new rate of one optimiser into the engine; it applies from the next
 * optimiser step.  group: on-policy contexts 0 (the one Adam; for CPO / TRPO-Lag the critics' Adam); FOCOPS 0 actor,
 * 1 critics; replay contexts (SAC / DDPG / CVPO) 0 actor, 1 critics, 2 alpha.  fsrl_get_lr returns < 0 for a bad group. */
/* BasePolicy.ret_rms (fsrl/policy/base_policy.py:111, 442-444): per critic the running (mean, var, count) of the normalised
 * returns, rows of 3 doubles, n = 3 * n_critics.  Only in contexts created with rew_norm.  The reference keeps these outside
 * state_dict; a host that wants them to survive a restart carries them itself. */
int fsrl_ret_rms_get(fsrl_ctx* ctx, double* out, int32_t n);
int fsrl_ret_rms_set(fsrl_ctx* ctx, const double* in, int32_t n);
int fsrl_set_lr(fsrl_ctx* ctx, int32_t group, float lr);
float fsrl_get_lr(const fsrl_ctx* ctx, int32_t group);

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
/* buffer[indices] (tianshou ReplayBufferManager.__getitem__ as FSRL's evaluation / dataset tooling uses it): the rows
 * stored at the given slots, copied to the host; any output pointer may be NULL.  Not on the training path.            */
int fsrl_store_read(fsrl_ctx* ctx, const int64_t* indices, int64_t n, float* obs_out, float* act_out, double* rew_out,
                    double* cost_out, uint8_t* terminated_out, uint8_t* truncated_out, float* obs_next_out);
/* VectorReplayBuffer(total_size, buffer_num) built inside learn() (fsrl/agent/base_agent.py:279,
 * fsrl/agent/sac_lag_agent.py learn): re-cut the store into buffer_num sub-buffers of ceil(total_size / buffer_num)
 * rows -- slot numbering, wrap-around and sample indices then are the reference's for that geometry.  It must fit the
 * allocation of fsrl_ctx_create (cfg.buffer_size rows, cfg.env_num sub-buffers), else FSRL_EINVAL; empties the store. */
int fsrl_store_configure(fsrl_ctx* ctx, int64_t total_size, int32_t buffer_num);
int fsrl_store_geometry(const fsrl_ctx* ctx, int64_t* sub_size_out, int32_t* buffer_num_out);
/* buffer.sample_indices(0): env-major, chronological inside each sub-buffer.             */
int fsrl_store_sample0(fsrl_ctx* ctx, int64_t* indices_out, int64_t cap, int64_t* n_out);

/* ---- actor inference for the collector (policy.forward under no_grad,
 *      fsrl/policy/base_policy.py:178-190; sampling stays on the host RNG) ------------ */
int fsrl_actor_forward(fsrl_ctx* ctx, const float* obs, int32_t k, float* mu_out,
                       float* sigma_out);
/* Collector-time sampling (fsrl/data/fast_collector.py:283-300 -> policy.forward -> dist.sample()): actor on
 * the device, k x act_dim draws from the library RNG (seed != 0 re-keys it).  On-policy contexts:
 * a = mu + exp(sigma_param) * eps; SAC contexts: a = tanh(mu + sigma(s) * eps).  deterministic != 0: the
 * mean.  act_out[k][act_dim] is the raw policy output (what the buffer stores; map_action stays with the
 * caller).  Not torch's random stream -- the host mirror of the actor remains for stream-exact runs.   */
int fsrl_actor_sample(fsrl_ctx* ctx, const float* obs, int32_t k, int32_t deterministic, uint64_t seed,
                      float* act_out);
/* One vector step of FastCollector.collect (fsrl/data/fast_collector.py:283-368) with the actor on the device, in
 * ONE call: (1) launch the actor on obs_act[k_act] -- the observations the NEXT actions are for (obs_next with the
 * reset observations of finished envs put in, surplus envs dropped); (2) while it runs, store the k transitions that
 * just finished exactly as fsrl_store_push does (k = 0 on the first step of a collect); (3) wait, draw the noise
 * as fsrl_actor_sample does (same stream: the two calls one after the other give the same numbers), and map the
 * action for env.step as BasePolicy.map_action does (base_policy.py:226-256): bound_method 0 none / 1 clip to
 * [-1, 1] / 2 tanh, then low + (high - low) * (a + 1) / 2 when act_low / act_high are given.
 * act_out[k_act][act_dim]: the policy's action (what the buffer stores); env_act_out: what the env takes.        */
int fsrl_collect_step(fsrl_ctx* ctx, const int32_t* env_ids, int32_t k, const float* obs, const float* act,
                      const double* rew, const double* cost, const uint8_t* terminated, const uint8_t* truncated,
                      const float* obs_next, int64_t* ptr_out, double* ep_rew_out, int32_t* ep_len_out,
                      int64_t* ep_idx_out, const float* obs_act, int32_t k_act, int32_t deterministic,
                      int32_t bound_method, const float* act_low, const float* act_high, float* act_out,
                      float* env_act_out);
/* The worker-process vector env as the native collector loop sees it: pointers into the env's shared-memory block
 * (fsrl_amd/env/shmem.py lays it out; tianshou's ShmemVectorEnv is what the reference uses,
 * examples/mlp/train_ppol_agent.py:120-123) and the geometry of its two handshake lanes.                            */
typedef struct fsrl_shm_env {
    float* obs;                 /* [env_num][obs_dim]  written by the workers                          */
    float* act;                 /* [env_num][act_dim]  written by the collector                        */
    double* rew; double* cost;  /* [env_num]                                                           */
    uint8_t* term; uint8_t* trunc; uint8_t* active;     /* [env_num]                                   */
    uint32_t* hs;               /* [2][3][16]: per lane generation | pending workers | command         */
    uint32_t* want;             /* [workers][16]: generation of the last command a worker takes part in */
    const int32_t* owner;       /* [env_num] worker of each env                                        */
    const int32_t* lane_of_worker;  /* [workers]                                                       */
    int32_t env_num, obs_dim, act_dim, workers, n_lanes;
    uint32_t gen[2];            /* in / out: last generation posted per lane                           */
    uint32_t spin;              /* polls before the collector sleeps on a lane's completion word       */
    const uint32_t* err;        /* [workers][16] or NULL: non-zero once a worker raised inside its env */
    const int32_t* pids;        /* [workers] or NULL: the workers' process ids (children of the caller): a killed worker
                                 * fails the native collect within 0.5 s instead of after the 60 s handshake timeout */
} fsrl_shm_env;
/* FastCollector.collect's inner loop (fast_collector.py:252-340) over that env, in C: starting from the envs `ready`
 * with observations obs[n], policy actions act[n] and mapped actions env_act[n] (the outputs of the last
 * fsrl_collect_step), repeat { env.step -> store the transitions, evaluate the actor on the new observations
 * (= fsrl_collect_step) } until a vector step finishes an episode or max_steps steps were taken.  THAT step's results
 * come back unstored in rew_out / cost_out / term_out / trunc_out / obs_next_out[n] (the caller resets the finished envs,
 * drops surplus ones and stores the rows with its next fsrl_collect_step, fast_collector.py:341-362); obs / act /
 * env_act then hold the inputs of that step.  *steps_out: vector steps taken (>= 1), *cost_sum_out: sum of their costs.
 * Same random stream and the same stored rows as the per-step calls.                                               */
int fsrl_collect_run(fsrl_ctx* ctx, fsrl_shm_env* env, const int32_t* ready, int32_t n, float* obs, float* act,
                     float* env_act, int32_t deterministic, int32_t bound_method, const float* act_low,
                     const float* act_high, int32_t max_steps, int32_t* steps_out, double* cost_sum_out, double* rew_out,
                     double* cost_out, uint8_t* term_out, uint8_t* trunc_out, float* obs_next_out);
/* replaces: FastCollector.collect(n_episode=...) as a whole over the worker-process env (fsrl/data/fast_collector.py:283-368): the
 * loop of fsrl_collect_run PLUS the episode boundaries -- reset of the finished envs (a reset command to their workers), episode
 * accounting, dropping of surplus envs (fast_collector.py:341-362).  ready / obs: the envs to start from and their current
 * observations.  Outputs: env steps taken, the summed cost, terminated / truncated counts, and per finished episode its
 * reward and length ([n_episode] each, in the order the loop met them); *episodes_out == n_episode on success.  The same
 * fsrl_collect_step calls in the same order as the interpreted loop: same rows in the same slots, same noise stream. */
int fsrl_collect_episodes(fsrl_ctx* ctx, fsrl_shm_env* env, const int32_t* ready, int32_t n, const float* obs, int32_t n_episode,
                          int32_t deterministic, int32_t bound_method, const float* act_low, const float* act_high,
                          int64_t* steps_out, double* total_cost_out, int32_t* term_count_out, int32_t* trunc_count_out,
                          double* ep_rew_out, int32_t* ep_len_out, int32_t* episodes_out);

/* The same collect, split-phase over the env's two lanes (needs n_lanes == 2): one lane's workers step while the collector stores
 * the other lane's transitions and evaluates the actor on its next observations; exactly n_episode episodes, global surplus rule
 * (fast_collector.py:341-362).  Rows and noise stream equal the interpreted split loop's (FastCollector._collect_split).   */
int fsrl_collect_episodes_split(fsrl_ctx* ctx, fsrl_shm_env* env, const int32_t* ready, int32_t n, const float* obs, int32_t n_episode,
                                int32_t deterministic, int32_t bound_method, const float* act_low, const float* act_high,
                                int64_t* steps_out, double* total_cost_out, int32_t* term_count_out, int32_t* trunc_count_out,
                                double* ep_rew_out, int32_t* ep_len_out, int32_t* episodes_out);
/* out2[0] = seconds the last fsrl_collect_episodes[_split] waited for the env workers, out2[1] = seconds in its store + actor calls */
int fsrl_collect_timing(fsrl_ctx* ctx, double* out2);
/* The collector's actor as a RESIDENT workgroup (contexts with the fused two-layer networks, up to 64 rows per call):
 * instead of one kernel launch per vector step (fast_collector.py:283-300: `self.policy(self.data, last_state)` per step) one
 * workgroup stays on its CU for the length of a collect; the host rings a doorbell in pinned memory with the observations next to
 * it and spins on a completion word.  The kernel ends on the next library call that enqueues other work on the context's stream
 * (which is thereby ordered behind it), at the end of fsrl_collect_episodes[_split], and by itself after idle_timeout_us without a
 * doorbell (default 2000; <= 0 keeps the value).  on = 0: one launch per call.  Same actions bit for bit either way.           */
int fsrl_actor_set_resident(fsrl_ctx* ctx, int32_t on, double idle_timeout_us);
/* out3 = {resident kernels launched, actor calls served through the doorbell, 1 if a resident kernel is live}                  */
int fsrl_actor_resident_stats(fsrl_ctx* ctx, int64_t* out3);
/* end the resident kernel now (the caller's collect is over); a no-op when none is live.  Never needed for correctness.           */
int fsrl_actor_release(fsrl_ctx* ctx);

/* Fill level of the first n sub-buffers (len(buffer.buffers[e]); ReplayBufferManager.sample_indices weighs by it). */
int fsrl_store_sizes(const fsrl_ctx* ctx, int64_t* sizes_out, int32_t n);

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
/* stopped_out == NULL above only enqueues the pass: the caller may draw the next permutation while the device works
 * (rolling its RNG back if the pass turns out to be the last) and collects the verdict here before the next call.       */
int fsrl_ppo_pass_result(fsrl_ctx* ctx, int32_t* stopped_out);
/* end: drains per-minibatch stats ([n_steps][FSRL_PPO_NSTATS] float32, row-major).       */
int fsrl_ppo_end(fsrl_ctx* ctx, float* stats_out, int64_t cap_steps, int64_t* n_steps_out);
/* abandon an update after fsrl_ppo_begin when fsrl_ppo_end cannot be reached (an exception between the calls on the
 * caller's side, base_policy.py:345-351 has no such state): drains the stream and clears the begin/end state.       */
int fsrl_ppo_abort(fsrl_ctx* ctx);
/* convenience: begin + `repeat` passes (perms = [repeat][n] or NULL) + end.              */
int fsrl_ppo_update(fsrl_ctx* ctx, const double* lagrangians, double rescaling,
                    int32_t batch_size, int32_t repeat, const int64_t* perms, uint64_t seed,
                    float* stats_out, int64_t cap_steps, int64_t* n_steps_out,
                    int32_t* stopped_pass_out);

/* Launch plan of the minibatch step's forward / backward launch (A/B and tests; every plan gives the same bits): how many leading tiles of a
 * network are 32 rows tall.  -1 automatic (default: all of them once the 16-row tiles exceed the CU count, i.e. minibatches above
 * ~1 360 rows with three networks), 0 none, n > 0 min(n, tiles / 2).  Needs hidden >= 128 and obs_dim <= 64.                        */
int fsrl_ppo_set_plan(fsrl_ctx* ctx, int32_t tall_tiles);

/* ---- grouped updates: k independent PPO-Lagrangian agents of ONE network shape on one GPU, stepped in lock step
 *      (multi-seed runs; SURVEY 8e "within-GPU batching of k seeds").  The reference runs seeds as separate jobs; one
 *      agent's update is a chain of small dependent launches that leaves most of an MI355X idle, so k agents share every
 *      launch of the minibatch step (grid.y = member).  Per member the result is fsrl_ppo_update's on that member
 *      alone within the golden tests' tolerances, and bit-identical only when the launch shapes agree (a group of one,
 *      minibatches of at most 512 rows): the group picks its tile height from the number of active members and keeps the
 *      in-kernel weight-gradient reduction above 512 rows.  Members keep their own store, parameters,
 *      Adam state and random streams; shape and PPO hyper-parameters must agree, batch sizes N_i may differ.
 *      While grouped, a member's own update calls still work (they run on the group's stream).                         */
typedef struct fsrl_group fsrl_group;
int fsrl_group_create(fsrl_ctx** ctxs, int32_t k, fsrl_group** out);       /* 1 <= k <= 16; members are not owned   */
int fsrl_group_destroy(fsrl_group* group);
/* Launch plan of the group's forward / backward launch (A/B and tests; every plan gives the same bits): how many leading tiles of a
 * (member, network) are 32 rows tall.  -1 automatic (default: all of them once the 16-row tiles of the group exceed the CU count, e.g.
 * 8 members x 3 networks x 256 rows: 192 workgroups of 32 rows instead of 384 of 16), 0 none, n > 0 min(n, tiles / 2).
 * Needs hidden >= 128 and obs_dim <= 64 (else 16-row tiles whatever the plan).                                                      */
int fsrl_group_set_plan(fsrl_group* group, int32_t tall_tiles);
/* k x BasePolicy.update (base_policy.py:332-355).  lagrangians [k][n_critics - 1], rescaling [k]; perms: NULL (library
 * shuffle, member i seeded by seed + 1000003 i + pass) or k pointers to [repeat][N_i]; stats_out: NULL or k pointers to
 * [cap_steps][FSRL_PPO_NSTATS]; n_steps_out [k]; stopped_pass_out [k] (-1 = ran every pass).                          */
int fsrl_group_ppo_update(fsrl_group* group, const double* lagrangians, const double* rescaling, int32_t batch_size,
                          int32_t repeat, const int64_t* const* perms, uint64_t seed, float* const* stats_out,
                          int64_t cap_steps, int64_t* n_steps_out, int32_t* stopped_pass_out);

/* read back process_fn products of the current batch: which = "values" | "rets" | "advs"
 * ([n][n_critics] float32, like torch.stack(.., -1)) | "logp_old" ([n]).                 */
int fsrl_batch_get(fsrl_ctx* ctx, const char* which, float* out, int64_t cap);

/* ---- stand-alone float64 GAE scan on the device (gae_return, base_policy.py:524-540);
 *      v_next already masked.  Bit-exact with the sequential reference.                  */
int fsrl_gae_return(fsrl_ctx* ctx, const float* v, const float* v_next, const double* rew,
                    const uint8_t* end_flag, int64_t n, double gamma, double gae_lambda,
                    double* adv_out);

/* ---- stand-alone float64 n-step return on the device (nstep_return, base_policy.py:543-567; called from
 *      compute_nstep_returns :453-512 with metric = rew / info.cost of the WHOLE buffer :481, end_flag = done | unfinished,
 *      target_q already value-masked, indices = the buffer.next chain [n_step][bsz]).  out[bsz][q] float64.
 *      Bit-exact with the sequential reference.  n_step < 1 -> FSRL_EINVAL (the reference asserts, :472).   */
int fsrl_nstep_return(fsrl_ctx* ctx, const double* metric, const uint8_t* end_flag, int64_t len,
                      const float* target_q, const int64_t* indices, int64_t bsz, int64_t q, double gamma,
                      int32_t n_step, double* out);

/* ---- trust-region policy updates: CPO (fsrl/policy/cpo.py) and TRPO-Lagrangian
 *      (fsrl/policy/trpo_lag.py).  Same context / store / parameter layout as PPO-Lag;
 *      full batch (the reference runs them with batch_size = 99999).                       */
typedef struct fsrl_tr_config {
    float target_kl;            /* delta: 0.01 (CPO), 0.001 (TRPO-Lag)                       */
    float backtrack_coeff;      /* 0.8                                                       */
    float damping;              /* 0.1  (cpo.py:182 damping_coeff, trpo_lag.py:115)          */
    float l2_reg;               /* CPO critics: + l2_reg * sum(theta^2)   (cpo.py:155-156)   */
    float critic_lr;            /* Adam over the critic parameters                           */
    int32_t max_backtracks;     /* 10 (agent default) / 100 (cpo_cfg.py:17)                  */
    int32_t optim_critic_iters; /* critic Adam steps per repeat                              */
    int32_t cg_iters;           /* 10                                                        */
    int32_t norm_adv;           /* full-batch advantage normalisation                        */
    double cost_limit;          /* CPO                                                       */
} fsrl_tr_config;

#define FSRL_CPO_NSTATS 17  /* kl, entropy, rew_loss, cost_loss, optim_A, optim_B, optim_C, optim_Q,
                               optim_R, optim_S, optim_lam, optim_nu, optim_case, step_size
                               (cpo.py:335-350) then vf0, vf1, vf_total (cpo.py:160-161)     */
#define FSRL_TRPO_NSTATS 11 /* rescaling, lagrangian, actor_safety, actor_rew, actor_total, vf0,
                               vf1, vf_total, kl, step_size, entropy (trpo_lag.py:140-171,241-249) */

/* begin: buffer.sample(0) + process_fn (cpo.py:123-145 / trpo_lag.py:117-134): GAE, full-batch
 * advantage normalisation, logp_old, mean_old / std_old.                                      */
int fsrl_tr_begin(fsrl_ctx* ctx, const fsrl_tr_config* cfg, int64_t* n_out);
/* CPO.learn (cpo.py:353-370): `repeat` x { optim_critic_iters critic steps ; policy_loss }.
 * ave_cost_return = stats_train["cost"] (cpo.py:112-113).  stats_out: [repeat][FSRL_CPO_NSTATS]. */
int fsrl_cpo_learn(fsrl_ctx* ctx, double ave_cost_return, int32_t repeat, float* stats_out);
/* TRPOLagrangian.learn (trpo_lag.py:173-251).  stats_out: [repeat][FSRL_TRPO_NSTATS].          */
int fsrl_trpo_learn(fsrl_ctx* ctx, const double* lagrangians, double rescaling, int32_t repeat,
                    float* stats_out);
/* The same two with Batch.split(batch_size, shuffle=True, merge_last=True) INSIDE learn (cpo.py:357-358, trpo_lag.py:178):
 * every repeat walks the minibatches of one permutation (a remainder is merged into the last one), critic steps and the
 * policy step per minibatch, one stats row per minibatch.  perms: [repeat][N] (np.random.permutation draws of the caller)
 * or NULL = the library shuffles (seed).  batch_size >= N: one minibatch, the whole batch in store order (perms ignored:
 * the reference's shuffle of a full batch only reorders sums) -- fsrl_cpo_learn / fsrl_trpo_learn are that case.
 * stats_out: [cap_rows][FSRL_*_NSTATS]; *n_rows_out = rows written = repeat x minibatches.                              */
int fsrl_cpo_learn_mb(fsrl_ctx* ctx, double ave_cost_return, int32_t repeat, int32_t batch_size, const int64_t* perms,
                      uint64_t seed, float* stats_out, int64_t cap_rows, int64_t* n_rows_out);
int fsrl_trpo_learn_mb(fsrl_ctx* ctx, const double* lagrangians, double rescaling, int32_t repeat, int32_t batch_size,
                       const int64_t* perms, uint64_t seed, float* stats_out, int64_t cap_rows, int64_t* n_rows_out);
/* policy evaluations the line search of each repeat of the last learn call made (cpo.py:306-333,
 * trpo_lag.py:205-231); returns the number written.  The reference samples an action in every one of them --
 * callers that follow its random streams need the count.                                        */
int32_t fsrl_tr_linesearch_evals(fsrl_ctx* ctx, int32_t* out, int32_t cap);
/* building blocks, exposed for the parity tests (all on the batch of the last fsrl_tr_begin):
 * which = 0: grad of mean(ratio*A_r)   1: grad of -mean(ratio*A_c)   2: grad of mean KL(old||new)
 * out: flat ACTOR parameter vector (sigma_param, W1, b1, W2, b2, W3, b3).                      */
int64_t fsrl_actor_param_count(const fsrl_ctx* ctx);
int fsrl_tr_grad(fsrl_ctx* ctx, int32_t which, float* out, int64_t n);
/* out = H v (no damping), H = Hessian of mean KL(N(mean_old,std_old) || pi_theta) at the current theta */
int fsrl_tr_hvp(fsrl_ctx* ctx, const float* v, float* out, int64_t n);
/* the same product for a caller that promises that theta, the batch and mean_old / std_old are those of its previous
 * fsrl_tr_hvp / fsrl_tr_hvp_cached call: the theta-only activations (h1, h2, dout, dz2) are read back instead of recomputed --
 * the kernel the conjugate-gradient solves inside fsrl_cpo_learn / fsrl_trpo_learn run for their 2nd .. 11th product
 * (cpo.py:184-204).  Bit-identical to fsrl_tr_hvp.                                                                     */
int fsrl_tr_hvp_cached(fsrl_ctx* ctx, const float* v, float* out, int64_t n);
/* stats8: mean(ratio*A_r), mean(ratio*A_c), mean KL, mean(logp_old - logp), mean A_r, mean A_c, 0, 0 */
int fsrl_tr_eval(fsrl_ctx* ctx, double* stats8);
/* Kernel plan of the full-batch path, for A/B timing and the bit-identity tests (no reference counterpart: the reference
 * has one code path).  tile_rows: 0 = automatic (mixed 32- / 16-row tiles; at 256-wide layers, obs_dim <= 64 and act_dim <= 4
 * as TWO co-resident 512-thread workgroups per CU: fb_tile_co_kernel), 16 = 16-row tiles only, 32 = mixed tiles with one
 * 1024-thread workgroup per CU (round 4's kernel); adding 64 runs the critics' regression steps on the compute stream, one
 * launch after the other, instead of on a stream of their own beside the actor's step (r5; same launches, same order inside
 * each chain: identical results).  hvp: 0 = automatic (mixed tiles; the theta-only activations of the KL
 * Hessian product are computed by the first product of a conjugate-gradient solve and read back by the others: cpo.py:184-204
 * calls _MVP 10 + 1 times per right-hand side at one theta; the cached products run as two co-resident workgroups per CU at
 * 256-wide layers: fb_hvp_co_kernel), 1 = the 16-row kernel that recomputes everything, 2 = mixed tiles without the cache,
 * 3 = mixed tiles + cache with one 1024-thread workgroup per CU (round 4's kernel); adding 4 to any of them switches the
 * Gauss-Newton form of the product off (A/B): by default, whenever mean_old / std_old of the batch were computed at the present
 * theta -- every TRPO-Lag product (trpo_lag.py:189-190), CPO's products until its first line-search step -- the KL gradient
 * is identically zero and the heads use mu - mean_old = 0, std_old = sigma exactly (as the reference's autograd does on its
 * detached copy of the same forward), so the dz2 / dout terms vanish and are not computed.  wgrad (the weight-side products):
 * 0 = automatic -- round 6's tile jobs (fb_wgrad3_kernel: every workgroup a 64 x 64 MFMA tile job, 512 threads, two per CU,
 * XCD-aware block order) at 256-wide layers over >= 4096 rows, else round 5's split-K kernel;
 * 1 = the split-K kernel with XCD-aware placement of its blocks, 2 = the split-K kernel; 3 = the one-pass streaming kernel
 * (one workgroup per network, output quarter and row slice; operands by LDS-DMA); 4 = the tile jobs in
 * plain block order; 5 = the same in XCD-aware order with half the row splits; 6 = both.  The tile_rows / hvp plans give
 * bit-identical results under EVERY wgrad plan; wgrad 1 and 2 agree to the bit, 0 and 4 where the tile jobs run; the kernel
 * families add the rows up in different orders (fp32 MFMA chains of different lengths, partials in float64): their results
 * agree to rounding (tests/test_gpu_fullsize.py states the tolerance).  tile_rows + 64: the critics' steps on the compute
 * stream instead of beside the actor's step; tile_rows + 128: the host-side read-backs of an update (line-search statistics,
 * dot products of the dual solve) by hipMemcpyAsync + hipStreamSynchronize instead of completion words the reducing kernels
 * write to pinned memory (r6: CPO configs[2] 28.3 -> 27.65 ms per update); same results.                                   */
int fsrl_tr_set_plan(fsrl_ctx* ctx, int32_t tile_rows, int32_t hvp, int32_t wgrad);
/* A/B only: force how many 32-row tiles (per network) the co-resident launches of the tile kernel / of the cached Hessian
 * product start with, the remaining rows going to 16-row tiles behind them; -1 = automatic.  A PERSISTENT form exists:
 * 2 x (number of CUs) workgroups draw their tiles from a device counter, so every CU keeps a pair of tiles in
 * flight until the batch runs out -- that is the A/B form, selected by -2 in either argument; the default is one workgroup per
 * tile (the hardware dispatcher already hands tiles out dynamically, and 1 250 draws on one counter cost ~50 us per launch).  Neither the split nor the scheduling changes a result (a row's arithmetic does not depend on its tile's
 * height or on the workgroup that computes it).                                                                        */
int fsrl_tr_set_tile_split(fsrl_ctx* ctx, int32_t n32_tile, int32_t n32_hvp);
/* A/B only: how late the SECOND co-resident workgroup of every CU starts its first tile, in periods of 8 128 shader cycles
 * (s_sleep 127), for the tile kernel / the cached Hessian product; -1 = the built-in default.  Two workgroups that start in
 * the same cycle run their phases in lockstep and never overlap; the offset is created once per launch.  No effect on a result. */
int fsrl_tr_set_co_delay(fsrl_ctx* ctx, int32_t tile_periods, int32_t hvp_periods);

/* ---- FOCOPS (fsrl/policy/focops.py:126-251), on the PPO entry points: create the context
 *      with algo = FSRL_ALGO_FOCOPS (same networks and parameter vector as PPO-Lag), call fsrl_focops_init once,
 *      then per update: fsrl_focops_set_nu (the host-side nu step, focops.py:154-159) -> fsrl_ppo_begin
 *      (lagrangians / rescaling ignored) -> fsrl_ppo_pass x repeat (pass-level KL early stop against `delta`)
 *      -> fsrl_ppo_end.  Stats rows keep FSRL_PPO_NSTATS floats; the first FSRL_FOCOPS_NSTATS are
 *      nu_loss, nu_value, actor_loss, kl, entropy, vf0, vf1, vf_total (focops.py:158,208-213,166-176).      */
typedef struct fsrl_focops_config {
    float actor_lr, critic_lr;   /* two Adam optimisers: actor / both critics                              */
    float l2_reg;                /* critics: + l2_reg * sum(theta^2)                                       */
    float delta;                 /* pass-level mean-KL early stop                                          */
    float eta;                   /* rows with KL(new||old) > eta leave the actor loss                      */
    float tem_lambda;            /* temperature lambda                                                     */
    float max_grad_norm;         /* clip_grad_norm_ over the ACTOR parameters; 0 = off                     */
} fsrl_focops_config;
#define FSRL_FOCOPS_NSTATS 8
int fsrl_focops_init(fsrl_ctx* ctx, const fsrl_focops_config* cfg);
/* A/B and tests only: 1 = every FOCOPS minibatch step as the four-launch sequence (split-K weight gradients + their sum); 0 =
 * three launches whenever the minibatch has at most 512 rows (the default).  Same arithmetic per element, different
 * summation order of the weight gradients. */
int fsrl_focops_set_plan(fsrl_ctx* ctx, int32_t four_launch);
int fsrl_focops_set_nu(fsrl_ctx* ctx, double nu, double nu_loss);

/* ---- SAC-Lagrangian (fsrl/policy/sac_lag.py), off-policy on the HIP-resident replay store.
 *      Create the context with algo = FSRL_ALGO_SAC_LAG (obs_dim, act_dim <= 8, hidden, env_num,
 *      buffer_size, gamma are read from fsrl_config), then fsrl_sac_init.
 *      Parameter vectors in torch parameters() order:
 *        actor   : W1[H,Do] b1 W2[H,H] b2 Wmu[Da,H] bmu Wsig[Da,H] bsig   (ActorProb, conditioned sigma,
 *                  unbounded mean; sac_lag_agent.py:126-134)
 *        critics : for (reward, cost): pre1(W1[H,Do+Da] b1 W2 b2) pre2(..) last1(W[1,H] b) last2(W b)
 *                  (DoubleCritic, fsrl/utils/net/continuous.py:13-101)                          */
typedef struct fsrl_sac_config {
    float actor_lr, critic_lr, alpha_lr;   /* 5e-4, 1e-3, 3e-4                                   */
    float tau;                             /* Polyak, 0.05                                       */
    float alpha;                           /* fixed temperature when auto_alpha == 0             */
    float target_entropy;                  /* -act_dim                                           */
    int32_t n_step;                        /* 2 (sacl_cfg.py:21)                                 */
    int32_t auto_alpha;
    int32_t use_lagrangian;
    /* DDPG-Lagrangian (fsrl/policy/ddpg_lag.py:125-223) on the same entry points: deterministic actor
     * a = max_action * tanh(MLP(s)) with a target copy, ONE critic per metric (tianshou Critic on
     * concat(s, a)) with target copies, no entropy term.  Parameter vectors then are
     *   actor: W1 b1 W2 b2 W3[Da,H] b3      critics: for (reward, cost): W1[H,Do+Da] b1 W2 b2 W3[1,H] b3
     * fsrl_sac_params_get which = 3 returns the target actor.  exploration_sigma: std of the Gaussian
     * exploration noise fsrl_actor_sample adds (GaussianNoise, ddpg_lag_agent.py:80,160).              */
    int32_t deterministic;
    float exploration_sigma;
} fsrl_sac_config;
#define FSRL_SAC_NSTATS 10  /* rescaling, lagrangian, actor_safety, alpha_loss, alpha_value, actor_rew,
                               actor_total (sac_lag.py:231-257) then q0, q1, q_total (:203-208) */
int fsrl_sac_init(fsrl_ctx* ctx, const fsrl_sac_config* cfg);
/* A/B and tests only; plan is a 7-bit mask, 0..127 (default 0):
 *   bit 0 (1): split-K weight gradients (fb_wgrad_kernel) at every batch size (default: batches of up to 512 rows use the
 *              PPO step's one-workgroup-per-tile kernel; same products, different summation order);
 *   bit 1 (2): the sampler and the row gather as two launches (default: one launch);
 *   bit 2 (4): the float64 n-step targets (base_policy.py:453-512) as a launch of their own (default: the critics' tile
 *              launch computes its targets itself).  Layered contexts (hidden_sizes outside the fused kernels) ALWAYS run the
 *              stand-alone n-step launch, whatever the caller set;
 *   bit 3 (8): prefetch the next update's sample + gather on the side stream (double-buffered batch);
 *   bit 4 (16): sample + gather as ONE launch of their own (round 4: 10 launches per update).  Default (r5): the forward launch of
 *              both actors draws and gathers its own rows first (9 launches per update; fused two-layer networks only);
 *   bit 5 (32): no rider blocks.  Default (r6) where the actor's weight-gradient launch is fb_wgrad_kernel (batches above 512 rows):
 *              that launch leaves 150 CUs idle, and extra blocks of it draw + gather the NEXT update's batch into a second set of
 *              batch arrays, in stream order; the next update uses them if nothing that determines the sample changed (store
 *              contents, batch size, key, update count) and draws its own otherwise.  The reference's trainer runs its updates
 *              back to back between collects (offpolicy trainer: round(update_per_step * steps) calls of policy.update), so
 *              every update but the first after a collect finds its rows in place.
 *   bit 6 (64): the critics' split-K weight-gradient launch in plain block order.  Default (r6): XCD-aware order -- the blocks of one
 *              (network, split), which stream the same rows, behind one L2: 46 -> 20 MB of memory-side traffic per launch, same time.
 * Bits 1 .. 6 keep the same Philox counters, the same float64 operations and the same sums: bit-identical results. */
int fsrl_sac_set_plan(fsrl_ctx* ctx, int32_t plan);
int64_t fsrl_sac_param_count(const fsrl_ctx* ctx, int32_t which);    /* 0 actor, 1 critics         */
int fsrl_sac_params_set(fsrl_ctx* ctx, const float* actor, int64_t na, const float* critics, int64_t nc,
                        float log_alpha);   /* also copies critics -> critics_old, resets Adam    */
/* which: 0 actor, 1 critics, 2 critics_old (targets)                                            */
int fsrl_sac_params_get(fsrl_ctx* ctx, int32_t which, float* out, int64_t n, float* alpha_out);
/* Checkpoint load (policy.load_state_dict): overwrite ONE parameter set, same `which` as above; targets,
 * Adam moments and step counts are left alone.                                                        */
int fsrl_sac_params_put(fsrl_ctx* ctx, int32_t which, const float* in, int64_t n);
/* One SACLagrangian.update(batch_size, buffer) = sample + n-step targets + critic step + actor
 * step + alpha step + Polyak (sac_lag.py:185-269, base_policy.py:356-395,453-512).
 * Sampling: indices / eps_target / eps_pi are given TOGETHER (the caller's numpy / torch RNG
 * streams -- parity mode) or are all NULL (library RNG on the device: Philox4x32-10 keyed by
 * `seed` (0 = keep the current key), uniform over the stored rows -- nothing is staged from the host).
 * stats_out: FSRL_SAC_NSTATS floats, synchronous; NULL = the call only enqueues work and the row is
 * kept in a device ring for fsrl_sac_stats_drain.                                                  */
int fsrl_sac_update(fsrl_ctx* ctx, int32_t batch_size, const int64_t* indices, const float* eps_target,
                    const float* eps_pi, uint64_t seed, const double* lagrangians, double rescaling,
                    float* stats_out);
/* Statistics rows of the updates issued with stats_out == NULL since the last drain, oldest first
 * (the ring keeps the newest 4096).  Returns the number of rows written (>= 0) or an error code.   */
int64_t fsrl_sac_stats_drain(fsrl_ctx* ctx, float* out, int64_t max_rows);
/* The sample the last fsrl_sac_update used, either mode (tests: library-RNG updates can be replayed
 * through the caller-RNG arguments).                                                              */
int fsrl_sac_last_sample(fsrl_ctx* ctx, int64_t* indices, float* eps_target, float* eps_pi, int32_t batch_size);
/* the actor for the collector: mu and sigma = exp(clamp(log sigma)) of the tanh-Gaussian policy   */
int fsrl_sac_actor_forward(fsrl_ctx* ctx, const float* obs, int32_t k, float* mu_out, float* sigma_out);

/* ---- CVPO (fsrl/policy/cvpo.py:71-430), on the replay context of SAC-Lagrangian: create the
 *      context with algo = FSRL_ALGO_SAC_LAG, then fsrl_cvpo_init INSTEAD of fsrl_sac_init.
 *      Networks (cvpo_agent.py:143-186): Gaussian actor, mu = max_action * tanh(head), sigma = exp(clamp(head,
 *      -20, 2)), NOT squashed after sampling, with a hard-copied actor_old; per metric one SingleCritic
 *      (double_critic == 0; vector layout as DDPG-Lag's critics) or one DoubleCritic (layout as SAC's) with
 *      Polyak targets.  Parameters move through fsrl_sac_params_set / _get / _put (which = 3 is actor_old);
 *      the collector's actor through fsrl_sac_actor_forward / fsrl_actor_sample; the statistics ring through
 *      fsrl_sac_stats_drain with rows of FSRL_CVPO_NSTATS floats.                                            */
typedef struct fsrl_cvpo_config {
    float actor_lr, critic_lr;             /* 5e-4, 1e-3                           (cvpo_agent.py:101-102) */
    float tau;                             /* Polyak of critics_old, 0.05                                  */
    int32_t n_step;                        /* 2                                                            */
    int32_t double_critic;                 /* 0: SingleCritic (default), 1: DoubleCritic, predict = min    */
    int32_t sample_act_num;                /* K particles per state, 16                                    */
    int32_t estep_iter_num, mstep_iter_num;/* 1, 1                                                         */
    float estep_kl, estep_dual_max, estep_dual_lr;                 /* 0.02, 20, 0.02                       */
    float mstep_kl_mu, mstep_kl_std, mstep_dual_max, mstep_dual_lr;/* 0.005, 0.0005, 0.5, 0.1              */
    double qc_thres;                       /* cost_limit * (1 - gamma^T) / (1 - gamma) / T   (cvpo.py:138-141) */
} fsrl_cvpo_config;
#define FSRL_CVPO_NSTATS 17 /* estep_loss, dual0 (eta), dual1 (lambda)                      (cvpo.py:346-354)
                               kl_mu, kl_std, loss_kl, loss_mle, loss_total, dual_mu, dual_std, entropy (:405-415)
                               loss_q0, val_q0, loss_q1, val_q1, thres_q1, q_total           (:262-275)    */
int fsrl_cvpo_init(fsrl_ctx* ctx, const fsrl_cvpo_config* cfg);
/* CVPO.pre_update_fn (cvpo.py:178-188): zero the two M-step multipliers and their Adam state.             */
int fsrl_cvpo_pre_update(fsrl_ctx* ctx);
/* CVPO.post_update_fn (cvpo.py:190-193): actor_old <- actor.                                              */
int fsrl_cvpo_post_update(fsrl_ctx* ctx);
/* CVPO.update_cost_limit (cvpo.py:165-176): the new E-step threshold on Qc.                               */
int fsrl_cvpo_set_thres(fsrl_ctx* ctx, double qc_thres);
/* One CVPO.update(batch_size, buffer): sample + n-step targets (a' ~ actor, critics_old) + critic step +
 * E-step (K particles of actor_old through the UPDATED critics, Adam on (eta, lambda), softmax weights) +
 * M-step (weighted decoupled-Gaussian likelihood + KL multipliers, actor Adam) + Polyak (cvpo.py:206-430).
 * Sampling as fsrl_sac_update: indices [B], eps_target [B][Da] and eps_particles [K][B][Da] are given
 * together (caller RNG) or all NULL (Philox on the device).  stats_out: FSRL_CVPO_NSTATS floats or NULL.    */
int fsrl_cvpo_update(fsrl_ctx* ctx, int32_t batch_size, const int64_t* indices, const float* eps_target,
                     const float* eps_particles, uint64_t seed, float* stats_out);
/* out[0..3] = eta, lambda (estep_dual, clamped) and mstep_dual_mu, mstep_dual_std (stored unclipped).      */
int fsrl_cvpo_duals_get(fsrl_ctx* ctx, float* out4);
/* The K particles' N(0,1) block of the last update ([K][B][Da]); indices / eps_target: fsrl_sac_last_sample. */
int fsrl_cvpo_last_particles(fsrl_ctx* ctx, float* eps_particles, int64_t n);

/* ---- timing of the last update, measured with hipEvents on the compute stream --------- */
/* out[0] = process_fn ms, out[1] = learn ms (all passes), out[2] = fused fwd/bwd kernel
 * total ms over the update (sum of per-launch event pairs when profiling is enabled),
 * out[3] = number of fwd/bwd launches, out[4] (if n >= 5) = the raw event-bracket sum (no
 * overhead correction).                                                                   */
int fsrl_set_profiling(fsrl_ctx* ctx, int enable);
int fsrl_last_timing(fsrl_ctx* ctx, double* out, int32_t n);

/* The part of a PPO optimiser step (ppo_lag.py:224-243: forward/backward, clip, Adam = three dependent launches here) that no
 * kernel work can remove, measured on this device: empty kernels with the step's grids, block sizes and LDS footprints for a
 * minibatch of mb_rows rows, `iters` times back to back on the compute stream.  out_us[0..2] = one launch of the forward/backward,
 * weight-gradient and Adam grid behind itself, out_us[3] = the three behind each other (microseconds per launch / per triple). */
int fsrl_launch_floors(fsrl_ctx* ctx, int32_t mb_rows, int32_t iters, double* out_us);

/* ---- the one collective of the multi-GPU layout: independent agents, one process per GPU; once per epoch the
 *      ranks sum a short float64 metric vector [n_st, n_ep, sum rew, sum cost, ...] (fsrl_amd/parallel.py EPOCH_KEYS).  The
 *      reference has no multi-GPU code.  RCCL (ncclAllReduce, sum, float64, on the context's compute stream) is reached through
 *      dlopen -- the copy of librccl the process already uses (torch's, when torch.distributed runs "nccl"), else /opt/rocm's --
 *      so the library links against nothing and loads on boxes without RCCL.  The Python host may equally use torch.distributed
 *      (fsrl_amd.parallel.reduce_epoch does when no communicator was initialised here).
 *      fsrl_comm_unique_id: rank 0, 128 bytes (ncclUniqueId), handed to every rank by the caller (file / TCP / launcher store);
 *      fsrl_comm_init: every rank, collective; fsrl_metrics_allreduce: in place, 1 <= n <= 64, the identity without a communicator. */
int fsrl_comm_unique_id(uint8_t* id_out, int32_t cap);
int fsrl_comm_init(fsrl_ctx* ctx, int32_t rank, int32_t world, const uint8_t* id, int32_t id_bytes);
int fsrl_metrics_allreduce(fsrl_ctx* ctx, double* v, int32_t n);
int fsrl_comm_info(const fsrl_ctx* ctx, int32_t* rank_out, int32_t* world_out);
int fsrl_comm_destroy(fsrl_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* FSRL_HIP_H */
