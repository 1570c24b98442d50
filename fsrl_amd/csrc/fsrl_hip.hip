// fsrl_hip.hip -- C ABI of libfsrl_hip.so (see include/fsrl_hip.h).  gfx950 only.
//
// Host side: context, parameter plumbing, the HIP-resident transition store (pinned staging +
// hipMemcpyAsync on a side stream), and the launch sequences of the policy update.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstddef>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <queue>
#include <string>
#include <vector>

#include "kernels_misc.hpp"
#include "kernels_mlp.hpp"
#include "kernels_fb.hpp"
#include "kernels_fbco.hpp"
#include "kernels_wgrad2.hpp"
#include "kernels_wgrad3.hpp"
#include "kernels_layered.hpp"

// ------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return fail(FSRL_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                        __FILE__, __LINE__);                                                \
    } while (0)
#define CHECK_ARG(cond, ...) \
    do { if (!(cond)) return fail(FSRL_EINVAL, __VA_ARGS__); } while (0)

extern "C" const char* fsrl_last_error(void) { return g_err.c_str(); }

// ------------------------------------------------------------------------------ context
static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// One parameter tensor: `rows x cols` values, dense at the flat API offset, rows `dev_ld` apart on the device.  The device
// layout is that of two hidden layers of the kernels' width H (64 / 128 / 256); the API layout is the caller's network with
// its own widths h1, h2 <= H (fsrl_config.hidden1 / hidden2).  Padded units have zero weights and biases: their
// pre-activation is 0, relu' is taken off h > 0, so every gradient entry of a padded weight is exactly 0 and Adam (also
// with an L2 term), Polyak averaging and the flat-vector algebra of the trust-region path leave it 0 for ever.
struct TensorMap {
    int api_off, dev_off, rows, cols, dev_ld;
    int n() const { return rows * cols; }
    void to_dev(float* dev, const float* api) const {
        for (int r = 0; r < rows; ++r) memcpy(dev + dev_off + (size_t)r * dev_ld, api + api_off + (size_t)r * cols, (size_t)cols * 4);
    }
    void to_api(float* api, const float* dev) const {
        for (int r = 0; r < rows; ++r) memcpy(api + api_off + (size_t)r * cols, dev + dev_off + (size_t)r * dev_ld, (size_t)cols * 4);
    }
};

struct EnvBook {           // tianshou ReplayBuffer bookkeeping of one sub-buffer (host side)
    int64_t index = 0, size = 0, last_index = 0;
    int64_t staged = 0;    // rows of this sub-buffer in the current staging window
    double ep_rew = 0.0;
    int32_t ep_len = 0;
    int64_t ep_idx = 0;
};

// Pinned host staging window + its device mirror (one of two).  Rows are PACKED records -- rew f64 | cost f64 | slot i32 |
// flags u32 | obs[Do] | obs_next[Do] | act[Da] -- so that a flush is ONE hipMemcpyAsync of count x rec bytes on the side
// stream (it was seven, one per column: at one flush per vector step the copies' submission cost more than the bytes).
struct Staging {
    uint8_t* h = nullptr;      // pinned, STAGE_CAP x rec bytes
    uint8_t* d = nullptr;      // device mirror
    int count = 0;
    hipEvent_t done = nullptr;
    bool in_flight = false;
};
static inline size_t stage_rec_bytes(int Do, int Da) { return ((size_t)24 + 4 * (size_t)(2 * Do + Da) + 7) / 8 * 8; }

struct fsrl_group;
struct fsrl_ctx {
    fsrl_config cfg{};
    fsrl_group* group = nullptr;   // set while the context is a member of a grouped-update set (host_group.inc)
    struct CommState* comm = nullptr;   // RCCL communicator of the metric exchange (host_comm.inc), owned
    int device = 0;
    hipStream_t compute = nullptr, side = nullptr;
    ModelDesc md{};
    std::vector<TensorMap> tmap;
    int h1 = 0, h2 = 0;    // widths of the caller's two hidden layers (<= cfg.hidden, the padded width the kernels run at)
    int64_t n_api = 0;     // flat parameter count (API)
    int n_dev = 0;         // padded device parameter count (main vector: what Adam / clip / copies see)
    int n_alloc = 0;       // n_dev + the forward-fragment mirrors of every W2 (P only)
    float *P = nullptr, *M = nullptr, *V = nullptr, *G = nullptr;
    int64_t adam_t = 0;
    CtrlBlock* ctrl = nullptr;
    CtrlBlock* h_ctrl = nullptr;  // pinned
    bool verdict_pending = false; // a pass's stop flag is on its way to h_ctrl (fsrl_ppo_pass with stopped_out == NULL)

    // store
    int64_t sub_size = 0, maxsize = 0;
    float* mbstat = nullptr; size_t mbstat_cap = 0;      // per-minibatch advantage statistics of the current pass
    double* d_rms = nullptr;           // reward_normalization: [n_critics][3] running (mean, var, count) of the returns
    double* ret64 = nullptr; int64_t ret64_cap = 0;   // float64 normalised returns of the current batch (rms update input)
    int64_t alloc_rows = 0;            // rows every store / batch array was allocated for (fsrl_store_configure stays inside)
    int active_envs = 0;               // sub-buffers in use (<= cfg.env_num)
    std::vector<EnvBook> env;
    std::vector<uint8_t> h_flags;      // host mirror of (terminated|truncated<<1) per slot
    StorePtrs st{};
    static constexpr int STAGE_CAP = 4096;
    Staging stage[2];
    int cur_stage = 0;
    hipEvent_t store_ready = nullptr;

    // batch (sample(0) order)
    BatchPtrs b{};
    int* d_indices = nullptr; uint8_t* d_end = nullptr; int* d_seg = nullptr;
    int* h_indices = nullptr; uint8_t* h_end = nullptr; int* h_seg = nullptr;  // pinned
    float *values = nullptr, *vnext = nullptr, *advs = nullptr, *rets = nullptr, *logp_old = nullptr;
    int64_t N = 0;
    int n_seg = 0;                 // episode segments of the current batch (GAE scan)
    bool batch_ready = false;

    // ppo working set
    int batch_size = 0, mbp_max = 0, n_tiles_max = 0;
    float *A1 = nullptr, *A2 = nullptr, *D1 = nullptr, *D2 = nullptr, *DO = nullptr;
    float *obs_p = nullptr, *rd_p = nullptr;   // pass-ordered batch (+32 pad rows)
    float *statp = nullptr, *gsq_part = nullptr;
    int *d_perm = nullptr, *h_perm = nullptr;            // [N] (pinned host)
    int *d_mbstart = nullptr, *d_mbsize = nullptr, *h_mbplan = nullptr;  // h_mbplan pinned [2*cap]
    size_t mb_cap = 0;
    std::vector<int> mb_start, mb_size;
    float* d_stats = nullptr; int64_t stats_cap = 0;     // [steps][11]
    int64_t n_steps = 0;
    int pass_index = 0;
    double lagr[FSRL_MAX_CRITICS] = {0, 0, 0, 0};
    double rescaling = 1.0;
    bool in_update = false;

    // scratch for the small host-pointer APIs
    void* scratch = nullptr; size_t scratch_bytes = 0;

    // timing
    bool profiling = false;
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr;
    std::vector<hipEvent_t> k_ev;     // pairs around the fwd/bwd kernel (profiling mode)
    size_t k_ev_used = 0;
    double t_process_ms = 0, t_learn_ms = 0, t_fwdbwd_ms = 0, t_fwdbwd_raw_ms = 0;
    int64_t n_fwdbwd = 0;
    uint64_t rng[4] = {0x9E3779B97F4A7C15ull, 0xBF58476D1CE4E5B9ull, 0x94D049BB133111EBull, 1};          // the collector's action noise
    uint64_t shuffle_rng[4] = {0xD6E8FEB86659FD93ull, 0xA0761D6478BD642Full, 0xE7037ED1A0B428DBull, 2};  // the library's own minibatch shuffles (perm == NULL): a stream of its own
    // split-K partial gradients of fb_wgrad_kernel, one buffer per parameter layout (keyed by its
    // padded size) so that the never-written inter-tensor padding stays zero
    struct Parts { int stride = 0; float* p = nullptr; size_t floats = 0; } parts[3];
    float* wg_parts = nullptr;      // the buffer the last wgrad_launch wrote
    int n_cus = 256;                // compute units of the device (tile-shape heuristic)
    bool wgrad_xcd = false;         // fb_wgrad_kernel: XCD-aware placement of the splits (fsrl_tr_set_plan)
    bool wgrad_stream = false;      // fsrl_tr_set_plan(wgrad = 3): fb_wgrad2_kernel (one streaming pass per workgroup) where it applies
    int wgrad_tiles = 1;            // r6: fb_wgrad3_kernel (every workgroup a 64 x 64 tile job, two per CU) where it applies: 1 = default
                                    // (XCD-aware order), 0 = off (fsrl_tr_set_plan wgrad 1 / 2 / 3), bit 1 = plain block order, bit 2 = half the splits
    struct FocState* foc = nullptr; // FOCOPS working set, owned
    float* mu_old = nullptr;        // [maxsize][Da] actor means at process time (FOCOPS)
    float* sigma_old = nullptr;     // [FSRL_MAX_ACT] sigma_param at process time (FOCOPS)
    void* h_actor = nullptr; size_t h_actor_bytes = 0;   // pinned staging of fsrl_actor_forward
    int actor_k = 0; size_t actor_ob = 0, actor_mb = 0;  // geometry of the actor evaluation in flight
    unsigned* h_done = nullptr; int done_cap = 0;        // pinned per-block completion words of that evaluation
    unsigned actor_seq = 0; int actor_blocks = 0;
    // r6: the collector's actor as a RESIDENT workgroup (actor_resident_kernel: one launch per collect instead of one per vector step)
    bool pa_on = true;              // fsrl_actor_set_resident
    bool pa_live = false;           // a kernel of generation pa_gen was launched and not told to end
    bool pa_req = false;            // the evaluation in flight went to the resident kernel (actor_eval_finish)
    unsigned pa_gen = 0, pa_seq = 0; int pa_cap = 0, pa_blocks = 1; double pa_idle_us = 2000.0;
    void* h_pa = nullptr;           // pinned: [bell 8 B | pad | done[4] at 16 | state[4] at 32 | pad to 64 B | obs cap x Do | mu cap x Da | sigma_param]
    long long pa_launches = 0, pa_requests = 0;          // fsrl_actor_resident_stats
    // timing-probe switches: always 0 / false in the shipped library; a -DFSRL_PROBES build reads them ONCE, at
    // fsrl_ctx_create, from FSRL_DBG_PHASE / FSRL_TILE16 / FSRL_WGRAD_SKIP / FSRL_NO_SPIN (tools/phase_probe.sh)
    int probe_phase = 0, probe_wgrad_skip = 0;
    unsigned long long* probe_ts = nullptr;   // probe builds: [1024][16] phase stamps of the last fused-kernel launch
    bool probe_tile16 = false;
    int tall_tiles = -1;               // fsrl_ppo_set_plan: 32-row tiles of the minibatch step's forward / backward launch (-1 automatic)
    bool no_fuse_adam = false;      // probe builds: FSRL_NO_FUSE_ADAM keeps the separate Adam launch without a clip (A/B, bit-compare)
    bool no_xcd_pair = false;       // probe builds: FSRL_NO_XCD_PAIR keeps the tile-major block order of the fused forward/backward launch (A/B)
    bool no_spin = false;           // wait for the collector's actor with hipStreamSynchronize instead of the completion words
    std::vector<float> act_mu, act_sg;                   // mean / std of the last actor evaluation (host)
    std::vector<int> perm_tmp;      // this pass's permutation before it goes to the pinned buffer
    hipEvent_t perm_copied = nullptr; bool perm_in_flight = false;
    uint64_t store_version = 1;     // bumped by every push / reset (device copies of the bookkeeping)
    uint64_t joined_version = 0;    // store_version at the last side -> compute stream join (join_store)
    double t_collect_env = 0.0, t_collect_act = 0.0;   // fsrl_collect_timing
    float* snap = nullptr;          // fsrl_state_snapshot: P (with mirrors) | M | V
    int64_t snap_adam_t = 0, snap_critic_t = 0, snap_foc_a = 0, snap_foc_c = 0; bool snap_valid = false;
    struct LayState* lay = nullptr; // layered PPO-Lag context (hidden_sizes of other depths / widths, host_layered.inc), owned
    struct TrState* tr = nullptr;   // trust-region (CPO / TRPO-Lag) working set, owned
    uint64_t theta_version = 1;     // bumped by everything that may write the actor's parameters (uploads, restores, optimiser
                                    // steps, line-search steps): TrState::rd_version == theta_version <=> mean_old / std_old of the
                                    // trust-region batch were computed at the CURRENT theta (the Gauss-Newton form of the KL product)
    void* sac = nullptr;            // SacState, owned
};
static bool ctx_is_replay(const fsrl_ctx* c);
static void sac_free(fsrl_ctx* c);
static void tr_free(fsrl_ctx* c);
static void foc_free(fsrl_ctx* c);
static void tr_reset_optim(fsrl_ctx* c);
static int64_t tr_critic_steps_taken(fsrl_ctx* c);
static void tr_set_critic_steps(fsrl_ctx* c, int64_t t);
static void foc_steps(fsrl_ctx* c, int64_t* t_actor, int64_t* t_critic, bool set);
static void foc_reset_optim(fsrl_ctx* c);
static void group_detach(fsrl_ctx* c);
static void lay_free(fsrl_ctx* c);
static void lay_build_layout(fsrl_ctx* c, const int* widths, int L);
static int lay_ensure_mb(fsrl_ctx* c, int B);
static int lay_ppo_steps(fsrl_ctx* c);
struct InferArgs;
static int lay_infer(fsrl_ctx* c, const InferArgs& ia, int jobs_y, hipStream_t s);
static void comm_free(fsrl_ctx* c);
static int focops_pass(fsrl_ctx* c, int32_t* stopped_out);
static int pass_verdict(fsrl_ctx* c, int32_t* stopped_out);

// Make the rows pushed so far visible to the compute stream: flush the staging window and order the side stream's
// copies before whatever the compute stream runs next.  Off-policy trainers call update() many times between collects;
// the join (an event record + wait, ~6 us of GPU idle per update) is skipped while the store has not changed.
static int flush_stage(fsrl_ctx* c);
// Every entry point that may enqueue work on the compute stream tells the resident actor (if one is live) to end first: the work is
// then behind a kernel that is on its way out, and the next actor call launches a new one BEHIND that work -- stream order as with
// one launch per call.  The actor / collector entry points themselves keep it (plain hipSetDevice).
static void pactor_release(fsrl_ctx* c);
#define ENTER_DEV(c) do { HIPCHK(hipSetDevice((c)->device)); pactor_release(c); } while (0)

static int join_store(fsrl_ctx* c) {
    if (c->joined_version == c->store_version) return 0;
    int rc = flush_stage(c);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->store_ready, c->side));
    HIPCHK(hipStreamWaitEvent(c->compute, c->store_ready, 0));
    c->joined_version = c->store_version;
    return 0;
}

static int ensure_scratch(fsrl_ctx* c, size_t bytes) {
    if (c->scratch_bytes >= bytes) return 0;
    if (c->scratch) HIPCHK(hipFree(c->scratch));
    c->scratch = nullptr; c->scratch_bytes = 0;
    HIPCHK(hipMalloc(&c->scratch, bytes));
    c->scratch_bytes = bytes;
    return 0;
}

// ---- split-K launch of fb_wgrad_kernel.  Rows are cut into <= 24 splits of >= 256 rows; every split
//      writes a partial gradient at parts + z * stride, summed later in z order.
struct WgradPlan { int nsplit, ks_per_split; };
static WgradPlan wgrad_plan(int rows, int blocks_per_split, int n_cus) {
    const int KS = rows >> 2;
    int n = std::max(1, std::min((rows + 255) / 256, 24));
    // One 1024-thread workgroup per CU: more than n_cus blocks means several rounds, and a last round that is mostly empty
    // costs as much as a full one (CPO's HVP weight products at N = 20 000: 25 x 23 = 575 blocks = 3 rounds of 40 us where
    // 250 blocks of 2.3x the rows take one round of 83 us).  Past one round, split so that the grid is exactly one round.
    if (blocks_per_split * n > n_cus) n = std::max(1, n_cus / std::max(blocks_per_split, 1));
    const int per = round_up((KS + n - 1) / n, 16);
    n = (KS + per - 1) / per;
    return WgradPlan{std::max(n, 1), per};
}
static int ensure_parts(fsrl_ctx* c, int stride, int nsplit) {
    fsrl_ctx::Parts* slot = nullptr;
    for (auto& pp : c->parts)
        if (pp.stride == stride || pp.stride == 0) { slot = &pp; break; }
    if (!slot) return fail(FSRL_ESTATE, "no free split-K buffer slot");
    const size_t floats = (size_t)stride * nsplit;
    if (slot->floats < floats) {
        HIPCHK(hipStreamSynchronize(c->compute));
        if (slot->p) HIPCHK(hipFree(slot->p));
        slot->p = nullptr; slot->floats = 0; slot->stride = stride;
        HIPCHK(hipMalloc(&slot->p, floats * 4));
        HIPCHK(hipMemsetAsync(slot->p, 0, floats * 4, c->compute));
        slot->floats = floats;
    }
    c->wg_parts = slot->p;
    return 0;
}
// rider (replay agents, the actor's launch): blocks appended along x draw + gather the next update's batch (kernels_sample.hpp);
// *rider_done tells the caller whether this launch carried them (only the 3-D grid of round 5's split-K kernel does)
template <bool PAIR2>
static int wgrad_launch(fsrl_ctx* c, const ModelDesc& md, FbWgradArgs& wa, int ny, int stride, int* nsplit,
                        SgRider* rider = nullptr, bool* rider_done = nullptr, int xcd_order = -1) {
    if (rider_done) *rider_done = false;
    const int H_ = c->cfg.hidden;
    // r6 default at 256 wide over a few thousand rows or more, for the callers that hand over the re-laid observations: every
    // workgroup a 64 x 64 tile job (kernels_wgrad3.hpp), 512 threads, two per CU, ONE round of at most 2 x CUs workgroups, in
    // XCD-aware block order.  Same box, against round 5's kernel: obs 60 (CPO configs[2]) 32.3 -> 28.9 ms, 362 -> 163 MB per launch;
    // obs 8 (TRPO-Lag, one dW1 job per split) 21.93 -> 21.69 ms -- in plain block order it lost there (22.3), which is why narrow
    // observations kept round 5's kernel until the XCD-aware order existed.
    const bool wg3_auto = c->wgrad_tiles != 0;
    if (H_ == 256 && wa.rows >= 4096 && wa.obs_pad && wg3_auto) {
        for (int y = 0; y < ny; ++y)
            CHECK_ARG(wa.nets[y].b1_src == wa.nets[y].w1_y && wa.nets[y].b2_src == wa.nets[y].w2_ya && wa.nets[y].do_src == wa.nets[y].w3_ya,
                      "fb_wgrad3_kernel takes the bias sums off the operands of the matrix products");
        constexpr int NP = PAIR2 ? 2 : 1;
        int NB = 16 * NP + 2 * NP;                          // dW2 tile jobs + dW3 jobs, + the dW1 jobs of every observation group
        for (int ko = 0; ko < wa.obs_ko; ++ko) NB += wg3_ujobs(md.Do, ko);
        const int KS = wa.rows >> 2;
        int ns = std::max(1, std::min(24 / NP, 2 * c->n_cus / (NB * ny)));
        if (c->wgrad_tiles & 4) ns = std::max(1, ns / 2);
        const int per = round_up((KS + ns - 1) / ns, 8);
        ns = (KS + per - 1) / per;
        int rc = ensure_parts(c, stride, NP * ns);
        if (rc) return rc;
        wa.out = c->wg_parts; wa.split_stride = stride; wa.ks_per_split = per;
        wa.remap_total = NB * ny * ns; wa.remap_ny = ny; wa.wg3_flags = (c->wgrad_tiles & 2) ? 0 : 1;
        *nsplit = NP * ns;                                  // the consumers add NP x ns partials (pair b's slots behind pair a's)
        wa.dbg_skip = c->probe_wgrad_skip;                  // 0 outside probe builds
        hipLaunchKernelGGL((fb_wgrad3_kernel<256>), dim3(round_up(wa.remap_total, 8)), dim3(512), 0, c->compute, md, wa, NP, ns, NB);
        HIPCHK(hipGetLastError());
        return 0;
    }
    // fsrl_tr_set_plan(wgrad = 3), 256-wide layers over a few thousand rows or more: the one-pass streaming form
    // (kernels_wgrad2.hpp) -- a workgroup per (network, output quarter, row slice), ONE round of workgroups, slices of a multiple
    // of 32 rows.  It moves 2.6x fewer bytes (160 vs 413-430 MB per launch at N = 20 000) in the same time (80 vs 82 us for the
    // R-op product, 80 vs 78 us plain), and its 32-64 partials cost the consumers 3 us more per launch than the <= 24 of the
    // split-K kernel: CPO 36.5 vs 36.0 ms, TRPO-Lag 29.7 vs 29.0 ms same box -- so it is NOT the default.
    if (H_ == 256 && wa.rows >= 4096 && wa.rows % 16 == 0 && c->wgrad_stream) {
        for (int y = 0; y < ny; ++y)
            CHECK_ARG(wa.nets[y].b1_src == wa.nets[y].w1_y && wa.nets[y].b2_src == wa.nets[y].w2_ya && wa.nets[y].do_src == wa.nets[y].w3_ya,
                      "fb_wgrad2_kernel takes the bias sums off the operands of the matrix products");
        int ns = std::max(1, c->n_cus / (WG2_Q * ny));
        ns = std::min(ns, std::max(1, wa.rows / 128));
        const int rps = round_up((wa.rows + ns - 1) / ns, 32);
        ns = (wa.rows + rps - 1) / rps;
        int rc = ensure_parts(c, stride, ns);
        if (rc) return rc;
        wa.out = c->wg_parts; wa.split_stride = stride; wa.ks_per_split = rps / 4;
        *nsplit = ns;
        hipLaunchKernelGGL((fb_wgrad2_kernel<256, PAIR2>), dim3(8 * WG2_Q * ((ns + 7) / 8), ny), dim3(1024), 0, c->compute, md, wa, ns, rps);
        HIPCHK(hipGetLastError());
        return 0;
    }
    // blocks of one split and network: 64x64 dW2 tiles + one block per (64-column group, pass over the rows) + the db3 block.
    // Passes: the first carries NCH0 16-column chunks of dW1 (+ dW3, db1, db2), every further one four chunks.
    const int nch0 = PAIR2 ? 1 : 2;
    const int passes = 1 + std::max(0, (md.Do - 16 * nch0 + 63) / 64);
    const int NB = (H_ / 64) * (H_ / 64) + (H_ / FB_AUX_COLS) * passes + 1;
    const WgradPlan pl = wgrad_plan(wa.rows, NB * ny, c->n_cus);
    int rc = ensure_parts(c, stride, pl.nsplit);
    if (rc) return rc;
    wa.out = c->wg_parts; wa.ks_per_split = pl.ks_per_split; wa.split_stride = stride;
    wa.dbg_skip = c->probe_wgrad_skip;
    wa.aux_passes = passes;
    *nsplit = pl.nsplit;
    return dispatch_H(c->cfg.hidden, [&](auto hc) {
        constexpr int HH = decltype(hc)::value;
        if (xcd_order < 0 ? c->wgrad_xcd : xcd_order != 0) {      // xcd_order: a caller's own choice (the replay agents' critics)
            wa.remap_total = NB * ny * pl.nsplit; wa.remap_ny = ny;
            hipLaunchKernelGGL((fb_wgrad_kernel<HH, PAIR2>), dim3(round_up(wa.remap_total, 8)), dim3(1024), 0, c->compute, md, wa, NoRider{});
        } else if (rider && !PAIR2) {
            const int blocks = (rider->sa.B + SG_ROWS - 1) / SG_ROWS;
            rider->x0 = NB; rider->nx = (blocks + ny * pl.nsplit - 1) / (ny * pl.nsplit);
            hipLaunchKernelGGL((fb_wgrad_kernel<HH, false, SgRider>), dim3(NB + rider->nx, ny, pl.nsplit), dim3(1024), 0, c->compute, md, wa, *rider);
            if (rider_done) *rider_done = true;
        } else
        hipLaunchKernelGGL((fb_wgrad_kernel<HH, PAIR2>), dim3(NB, ny, pl.nsplit), dim3(1024), 0, c->compute, md, wa, NoRider{});
        HIPCHK(hipGetLastError());
        return 0;
    });
}

extern "C" void fsrl_config_default(fsrl_config* c) {
    // defaults of PPOLagAgent.__init__ (fsrl/agent/ppo_lag_agent.py:82-116)
    memset(c, 0, sizeof(*c));
    c->algo = FSRL_ALGO_PPO_LAG;
    c->obs_dim = 8; c->act_dim = 2; c->hidden = 128; c->n_critics = 2;
    c->env_num = 20; c->buffer_size = 100000; c->max_action = 1.0f;
    c->gamma = 0.99; c->gae_lambda = 0.95; c->eps_clip = 0.2f; c->dual_clip = 0.0f;
    c->vf_coef = 0.25f; c->max_grad_norm = 0.0f; c->target_kl = 0.02f;
    c->norm_adv = 1; c->use_lagrangian = 1;
    c->lr = 5e-4f; c->beta1 = 0.9f; c->beta2 = 0.999f; c->adam_eps = 1e-8f;
}

static void build_layout(fsrl_ctx* c) {
    const int Do = c->cfg.obs_dim, Da = c->cfg.act_dim, H = c->cfg.hidden;
    ModelDesc& md = c->md;
    md.Do = Do; md.Da = Da; md.H = H; md.n_nets = 1 + c->cfg.n_critics;
    md.unbounded = (c->cfg.unbounded && c->cfg.algo != FSRL_ALGO_SAC_LAG) ? 1 : 0;     // replay actors have their own (raw mu | log sigma) head
    const int h1 = c->h1, h2 = c->h2;
    int api = 0, dev = 0;
    // rows x cols API values inside a device tensor of `dev_n` floats whose rows are dev_ld apart
    auto add = [&](int rows, int cols, int dev_ld, int dev_n) {
        TensorMap t{api, dev, rows, cols, dev_ld};
        c->tmap.push_back(t);
        api += rows * cols;
        dev = round_up(dev + dev_n, 64);  // every tensor starts 256-byte aligned on the device
        return t.dev_off;
    };
    for (int net = 0; net < md.n_nets; ++net) {
        NetOff& no = md.net[net];
        no.begin = dev;
        const int out = (net == 0) ? Da : 1;
        no.out = out;
        no.sigma = (net == 0) ? add(1, Da, Da, Da) : -1;
        no.W1 = add(h1, Do, Do, H * Do); no.b1 = add(1, h1, H, H);
        no.W2 = add(h2, h1, H, H * H);   no.b2 = add(1, h2, H, H);
        no.W3 = add(out, h2, H, out * H); no.b3 = add(1, out, out, out);
        no.end = dev;
    }
    c->n_api = api;
    c->n_dev = round_up(dev, 1024);
    for (int net = 0; net < md.n_nets; ++net) md.net[net].W2f = c->n_dev + net * H * H;
    c->n_alloc = c->n_dev + md.n_nets * H * H;
}

// host copy of a parameter vector in device layout: fill the W2 mirrors behind the main part
static void fill_mirrors(const ModelDesc& md, std::vector<float>& v) {
    const int H = md.H;
    for (int net = 0; net < md.n_nets; ++net) {
        const NetOff& no = md.net[net];
        for (int n = 0; n < H; ++n)
            for (int k = 0; k < H; ++k) v[(size_t)no.W2f + w2f_index(H, n, k)] = v[(size_t)no.W2 + (size_t)n * H + k];
    }
}

extern "C" int fsrl_ctx_destroy(fsrl_ctx* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    pactor_release(c);
    (void)hipDeviceSynchronize();
    if (c->group) group_detach(c);          // a member destroyed before its group: take its own stream back
    comm_free(c);
    tr_free(c);
    sac_free(c);
    foc_free(c);
    lay_free(c);
    if (c->h_actor) (void)hipHostFree(c->h_actor);
    if (c->h_done) (void)hipHostFree(c->h_done);
    if (c->h_pa) (void)hipHostFree(c->h_pa);
    if (c->mbstat) (void)hipFree(c->mbstat);
    if (c->d_rms) (void)hipFree(c->d_rms);
    if (c->ret64) (void)hipFree(c->ret64);
    if (c->mu_old) (void)hipFree(c->mu_old);
    if (c->sigma_old) (void)hipFree(c->sigma_old);
    if (c->snap) (void)hipFree(c->snap);
    void* dptrs[] = {c->P, c->M, c->V, c->G, c->ctrl, c->st.obs, c->st.obs_next, c->st.act, c->st.rew,
                     c->st.cost, c->st.flags, c->b.obs, c->b.obs_next, c->b.act, c->b.rew, c->b.cost,
                     c->b.flags, c->d_indices, c->d_end, c->d_seg, c->values, c->vnext, c->advs,
                     c->rets, c->logp_old, c->A1, c->A2, c->D1, c->D2, c->DO, c->obs_p, c->rd_p, c->statp,
                     c->gsq_part, c->d_perm, c->d_mbstart, c->d_mbsize, c->d_stats,
                     c->scratch};
    for (void* p : dptrs) if (p) (void)hipFree(p);
    for (auto& pp : c->parts) if (pp.p) (void)hipFree(pp.p);
    void* hptrs[] = {c->h_ctrl, c->h_indices, c->h_end, c->h_seg, c->h_perm, c->h_mbplan};
    for (void* p : hptrs) if (p) (void)hipHostFree(p);
    for (auto& s : c->stage) {
        if (s.h) (void)hipHostFree(s.h);
        if (s.d) (void)hipFree(s.d);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (hipEvent_t e : {c->store_ready, c->ev_a, c->ev_b, c->ev_c, c->perm_copied}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->k_ev) (void)hipEventDestroy(e);
    if (c->compute) (void)hipStreamDestroy(c->compute);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    return 0;
}

extern "C" int fsrl_ctx_create(int device_id, const fsrl_config* cfg, fsrl_ctx** out) {
    CHECK_ARG(cfg && out, "null argument");
    CHECK_ARG(cfg->algo == FSRL_ALGO_PPO_LAG || cfg->algo == FSRL_ALGO_SAC_LAG || cfg->algo == FSRL_ALGO_CPO ||
                  cfg->algo == FSRL_ALGO_TRPO_LAG || cfg->algo == FSRL_ALGO_FOCOPS, "unknown algo %d", cfg->algo);
    CHECK_ARG(cfg->obs_dim >= 1 && cfg->obs_dim <= FSRL_MAX_OBS, "obs_dim must be in [1,%d]", FSRL_MAX_OBS);
    CHECK_ARG(cfg->act_dim >= 1 && cfg->act_dim <= FSRL_MAX_ACT, "act_dim must be in [1,%d]", FSRL_MAX_ACT);
    // hidden_sizes (fsrl/agent/ppo_lag_agent.py:91,136): n_hidden > 0 names them.  Two layers of at most 256 units run on the
    // fused kernels (H = 64 / 128 / 256, narrower layers zero-padded); any other depth / width is a LAYERED context
    // (host_layered.inc): every algorithm
    CHECK_ARG(cfg->n_hidden >= 0 && cfg->n_hidden <= FSRL_MAX_HIDDEN, "n_hidden must be in [0, %d]", FSRL_MAX_HIDDEN);
    bool layered = false;
    int n_hid1 = cfg->hidden1, n_hid2 = cfg->hidden2;
    if (cfg->n_hidden > 0) {
        CHECK_ARG(cfg->hidden1 == 0 && cfg->hidden2 == 0, "give hidden_sizes[n_hidden] or hidden1 / hidden2, not both");
        for (int l = 0; l < cfg->n_hidden; ++l)
            CHECK_ARG(cfg->hidden_sizes[l] >= 1 && cfg->hidden_sizes[l] <= FSRL_MAX_WIDTH, "hidden_sizes[%d] = %d outside [1, %d]", l,
                      cfg->hidden_sizes[l], FSRL_MAX_WIDTH);
        if (cfg->n_hidden == 2 && cfg->hidden_sizes[0] <= 256 && cfg->hidden_sizes[1] <= 256 && !cfg->force_layered) {
            n_hid1 = cfg->hidden_sizes[0]; n_hid2 = cfg->hidden_sizes[1];
        } else {
            layered = true;
        }
    } else {
        CHECK_ARG(!cfg->force_layered, "force_layered needs hidden_sizes[n_hidden]");
    }
    const int h1_ = layered ? 0 : (n_hid1 > 0 ? n_hid1 : cfg->hidden), h2_ = layered ? 0 : (n_hid2 > 0 ? n_hid2 : cfg->hidden);
    if (!layered) {
        CHECK_ARG((n_hid1 > 0) == (n_hid2 > 0), "hidden1 and hidden2 are given together (0, 0 = two layers of `hidden`)");
        CHECK_ARG(h1_ >= 1 && h1_ <= 256 && h2_ >= 1 && h2_ <= 256, "hidden layer widths must be in [1, 256] (two hidden layers)");
    }
    const int Hpad_ = layered ? 0 : std::max(h1_, h2_) <= 64 ? 64 : std::max(h1_, h2_) <= 128 ? 128 : 256;
    CHECK_ARG(layered || n_hid1 == 0 || cfg->hidden == 0 || cfg->hidden == Hpad_ ,
              "hidden = %d does not fit hidden1 / hidden2 = %d / %d (leave hidden 0 or give %d)", cfg->hidden, h1_, h2_, Hpad_);
    CHECK_ARG(cfg->n_critics >= 1 && cfg->n_critics <= 2,
              "n_critics must be 1 or 2 (reward [+ one cost], get_metrics base_policy.py:377-382)");
    CHECK_ARG(cfg->env_num >= 1 && cfg->buffer_size >= cfg->env_num, "bad env_num/buffer_size");
    CHECK_ARG(cfg->gamma >= 0.0 && cfg->gamma <= 1.0, "discount factor should be in [0, 1].");
    CHECK_ARG(cfg->gae_lambda >= 0.0 && cfg->gae_lambda <= 1.0, "GAE lambda should be in [0, 1].");
    CHECK_ARG(cfg->dual_clip == 0.0f || cfg->dual_clip > 1.0f,
              "Dual-clip PPO parameter should greater than 1.0.");
    CHECK_ARG(cfg->buffer_size + cfg->env_num < (int64_t)INT_MAX / 2, "buffer too large for 32-bit slot ids");
    CHECK_ARG(cfg->rew_norm || !cfg->value_clip, "value clip is available only when `reward_normalization` is True");
    CHECK_ARG(!cfg->value_clip || cfg->algo == FSRL_ALGO_PPO_LAG, "value_clip is a PPO-Lagrangian option (ppo_lag.py:158-164)");
    CHECK_ARG(!cfg->rew_norm || cfg->algo != FSRL_ALGO_SAC_LAG, "reward_normalization acts on GAE returns: on-policy contexts only");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    CHECK_ARG(device_id >= 0 && device_id < ndev, "device %d not present (%d devices)", device_id, ndev);
    HIPCHK(hipSetDevice(device_id));
    int n_cus_probe = 256;
    (void)hipDeviceGetAttribute(&n_cus_probe, hipDeviceAttributeMultiprocessorCount, device_id);
    fsrl_ctx* c = new fsrl_ctx();
    c->cfg = *cfg;
    c->cfg.hidden = Hpad_;                 // from here on `hidden` is the kernels' width; h1 / h2 are the caller's layers
    c->h1 = h1_; c->h2 = h2_;
    c->device = device_id;
    c->n_cus = n_cus_probe > 0 ? n_cus_probe : 256;
#ifdef FSRL_PROBES
    { const char* e = getenv("FSRL_DBG_PHASE"); c->probe_phase = e ? atoi(e) : 0; }
    { const char* e = getenv("FSRL_WGRAD_SKIP"); c->probe_wgrad_skip = e ? atoi(e) : 0; }
    c->probe_tile16 = getenv("FSRL_TILE16") != nullptr;
    c->no_spin = getenv("FSRL_NO_SPIN") != nullptr;
    c->no_xcd_pair = getenv("FSRL_NO_XCD_PAIR") != nullptr;
    c->no_fuse_adam = getenv("FSRL_NO_FUSE_ADAM") != nullptr;
    if (getenv("FSRL_TSTAMP")) {
        (void)hipMalloc(&c->probe_ts, 1024 * 16 * sizeof(unsigned long long));
        (void)hipMemset(c->probe_ts, 0, 1024 * 16 * sizeof(unsigned long long));
    }
#endif
    if (layered) lay_build_layout(c, cfg->hidden_sizes, cfg->n_hidden);
    else build_layout(c);
#define TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { fail(FSRL_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); fsrl_ctx_destroy(c); return FSRL_EHIP; } } while (0)
    TRY(hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking));
    TRY(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    const size_t pb = (size_t)c->n_dev * sizeof(float);
    TRY(hipMalloc(&c->P, (size_t)c->n_alloc * sizeof(float))); TRY(hipMemsetAsync(c->P, 0, (size_t)c->n_alloc * sizeof(float), c->compute));
    TRY(hipMalloc(&c->M, pb)); TRY(hipMalloc(&c->V, pb)); TRY(hipMalloc(&c->G, pb));
    TRY(hipMemsetAsync(c->M, 0, pb, c->compute)); TRY(hipMemsetAsync(c->V, 0, pb, c->compute)); TRY(hipMemsetAsync(c->G, 0, pb, c->compute));
    TRY(hipMalloc(&c->ctrl, sizeof(CtrlBlock)));
    if (cfg->rew_norm) {      // RunningMeanStd(): mean 0, var 1, count 0 per critic
        TRY(hipMalloc(&c->d_rms, 3 * FSRL_MAX_CRITICS * sizeof(double)));
        double init[3 * FSRL_MAX_CRITICS];
        for (int i = 0; i < FSRL_MAX_CRITICS; ++i) { init[3 * i] = 0.0; init[3 * i + 1] = 1.0; init[3 * i + 2] = 0.0; }
        TRY(hipMemcpy(c->d_rms, init, sizeof(init), hipMemcpyHostToDevice));
    }
    TRY(hipHostMalloc(&c->h_ctrl, sizeof(CtrlBlock)));
    // store: n sub-buffers of ceil(total/n) rows (tianshou VectorReplayBuffer)
    c->sub_size = (cfg->buffer_size + cfg->env_num - 1) / cfg->env_num;
    c->maxsize = c->sub_size * cfg->env_num;
    // + env_num rows of slack: fsrl_store_configure with fewer sub-buffers rounds ceil(total / n) * n up past `total`
    c->alloc_rows = c->maxsize + cfg->env_num; c->active_envs = cfg->env_num;
    c->env.resize(cfg->env_num);
    c->h_flags.assign((size_t)c->alloc_rows, 0);
    const size_t ms = (size_t)c->alloc_rows;
    const int Do = cfg->obs_dim, Da = cfg->act_dim;
    TRY(hipMalloc(&c->st.obs, ms * Do * 4)); TRY(hipMalloc(&c->st.obs_next, ms * Do * 4));
    TRY(hipMalloc(&c->st.act, ms * Da * 4)); TRY(hipMalloc(&c->st.rew, ms * 8));
    TRY(hipMalloc(&c->st.cost, ms * 8)); TRY(hipMalloc(&c->st.flags, ms));
    TRY(hipMalloc(&c->b.obs, ms * Do * 4)); TRY(hipMalloc(&c->b.obs_next, ms * Do * 4));
    TRY(hipMalloc(&c->b.act, ms * Da * 4)); TRY(hipMalloc(&c->b.rew, ms * 8));
    TRY(hipMalloc(&c->b.cost, ms * 8)); TRY(hipMalloc(&c->b.flags, ms));
    TRY(hipMalloc(&c->d_indices, ms * 4)); TRY(hipMalloc(&c->d_end, ms)); TRY(hipMalloc(&c->d_seg, (ms + 2) * 4));
    TRY(hipHostMalloc(&c->h_indices, ms * 4)); TRY(hipHostMalloc(&c->h_end, ms)); TRY(hipHostMalloc(&c->h_seg, (ms + 2) * 4));
    const int C = cfg->n_critics;
    TRY(hipMalloc(&c->values, ms * C * 4)); TRY(hipMalloc(&c->vnext, ms * C * 4));
    TRY(hipMalloc(&c->advs, ms * C * 4)); TRY(hipMalloc(&c->rets, ms * C * 4));
    TRY(hipMalloc(&c->logp_old, ms * 4));
    TRY(hipMalloc(&c->d_perm, ms * 4)); TRY(hipHostMalloc(&c->h_perm, ms * 4));
    TRY(hipMalloc(&c->obs_p, (ms + 32) * Do * 4)); TRY(hipMemsetAsync(c->obs_p, 0, (ms + 32) * Do * 4, c->compute));
    TRY(hipMalloc(&c->rd_p, (ms + 32) * FSRL_RD * 4)); TRY(hipMemsetAsync(c->rd_p, 0, (ms + 32) * FSRL_RD * 4, c->compute));
    TRY(hipStreamSynchronize(c->compute));     // zero fills have landed before any other stream touches them
    for (auto& s : c->stage) {
        const size_t k = fsrl_ctx::STAGE_CAP;
        TRY(hipHostMalloc(&s.h, k * stage_rec_bytes(Do, Da)));
        TRY(hipMalloc(&s.d, k * stage_rec_bytes(Do, Da)));
        TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    TRY(hipEventCreateWithFlags(&c->store_ready, hipEventDisableTiming));
    TRY(hipEventCreate(&c->ev_a)); TRY(hipEventCreate(&c->ev_b)); TRY(hipEventCreate(&c->ev_c));
    TRY(hipEventCreateWithFlags(&c->perm_copied, hipEventDisableTiming));
    CtrlBlock init{INT_MAX, 0, 0.0, 0.0f, 0.0f};
    TRY(hipMemcpy(c->ctrl, &init, sizeof(init), hipMemcpyHostToDevice));
#undef TRY
    *out = c;
    return 0;
}

extern "C" int fsrl_sync(fsrl_ctx* c) {
    CHECK_ARG(c, "null ctx");
    ENTER_DEV(c);
    HIPCHK(hipStreamSynchronize(c->side));
    HIPCHK(hipStreamSynchronize(c->compute));
    return 0;
}

// ------------------------------------------------------------------------------ parameters
extern "C" int64_t fsrl_param_count(const fsrl_ctx* c) { return c ? c->n_api : 0; }

static int copy_flat(fsrl_ctx* c, float* dev, float* host_out, const float* host_in, int64_t n) {
    CHECK_ARG(n == c->n_api, "expected %lld parameters, got %lld", (long long)c->n_api, (long long)n);
    ENTER_DEV(c);
    std::vector<float> tmp((size_t)c->n_dev, 0.0f);
    if (host_in) {                              // only P is ever written from the host
        for (const TensorMap& t : c->tmap) t.to_dev(tmp.data(), host_in);
        HIPCHK(hipStreamSynchronize(c->compute));
        HIPCHK(hipMemcpy(dev, tmp.data(), (size_t)c->n_dev * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(w2f_sync_kernel, dim3(192), dim3(256), 0, c->compute, dev, c->md);   // W2 mirrors
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipStreamSynchronize(c->compute));
        HIPCHK(hipMemcpy(tmp.data(), dev, (size_t)c->n_dev * 4, hipMemcpyDeviceToHost));
        for (const TensorMap& t : c->tmap) t.to_api(host_out, tmp.data());
    }
    return 0;
}

extern "C" int fsrl_params_set(fsrl_ctx* c, const float* flat, int64_t n) {
    CHECK_ARG(c && flat, "null argument");
    c->theta_version += 1;
    return copy_flat(c, c->P, nullptr, flat, n);
}
extern "C" int fsrl_params_get(fsrl_ctx* c, float* flat, int64_t n) {
    CHECK_ARG(c && flat, "null argument");
    return copy_flat(c, c->P, flat, nullptr, n);
}
extern "C" int fsrl_grads_get(fsrl_ctx* c, float* flat, int64_t n) {
    CHECK_ARG(c && flat, "null argument");
    return copy_flat(c, c->G, flat, nullptr, n);
}
extern "C" int fsrl_optim_reset(fsrl_ctx* c) {
    CHECK_ARG(c, "null ctx");
    ENTER_DEV(c);
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipMemsetAsync(c->M, 0, (size_t)c->n_dev * 4, c->compute));
    HIPCHK(hipMemsetAsync(c->V, 0, (size_t)c->n_dev * 4, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    c->adam_t = 0;
    tr_reset_optim(c);                                          // the critics' optimiser of CPO / TRPO-Lag shares M / V
    foc_reset_optim(c);
    return 0;
}

// Device-resident checkpoint of the on-policy training state: parameters (with their W2 mirrors), Adam moments and step
// counts, kept in HBM.  snapshot / restore are device-to-device copies on the compute stream -- no host round trip, no
// synchronisation: restoring between updates costs 3 x 0.8 MB of HBM traffic instead of a host repack + H2D copy + two stream
// drains (what fsrl_params_set + fsrl_optim_reset cost).  Uses: repeated measurements of ONE update from the same state
// (bench.py), roll-backs after a rejected update.  The reference has no counterpart (it would deepcopy state_dict()).
extern "C" int fsrl_state_snapshot(fsrl_ctx* c) {
    CHECK_ARG(c, "null ctx");
    CHECK_ARG(!c->in_update, "fsrl_state_snapshot inside an update");
    // replay agents keep their training state elsewhere (SacState: actor / Q / target stores, log alpha, three Adam states): a
    // snapshot of c->P / M / V would restore nothing of it
    CHECK_ARG(!ctx_is_replay(c), "fsrl_state_snapshot covers on-policy contexts (PPO-Lag, FOCOPS, CPO, TRPO-Lag) only");
    ENTER_DEV(c);
    if (!c->snap) {
        HIPCHK(hipMalloc(&c->snap, ((size_t)c->n_alloc + 2 * (size_t)c->n_dev) * 4 + 3 * FSRL_MAX_CRITICS * sizeof(double)));
    }
    hipStream_t st = c->compute;
    HIPCHK(hipMemcpyAsync(c->snap, c->P, (size_t)c->n_alloc * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(c->snap + c->n_alloc, c->M, (size_t)c->n_dev * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(c->snap + c->n_alloc + c->n_dev, c->V, (size_t)c->n_dev * 4, hipMemcpyDeviceToDevice, st));
    if (c->d_rms)       // reward_normalization: the running return statistics are training state too
        HIPCHK(hipMemcpyAsync(c->snap + c->n_alloc + 2 * (size_t)c->n_dev, c->d_rms, 3 * FSRL_MAX_CRITICS * sizeof(double),
                              hipMemcpyDeviceToDevice, st));
    c->snap_adam_t = c->adam_t;
    c->snap_critic_t = tr_critic_steps_taken(c);
    foc_steps(c, &c->snap_foc_a, &c->snap_foc_c, false);
    c->snap_valid = true;
    return 0;
}
extern "C" int fsrl_state_restore(fsrl_ctx* c) {
    CHECK_ARG(c, "null ctx");
    CHECK_ARG(!c->in_update, "fsrl_state_restore inside an update");
    if (!c->snap_valid) return fail(FSRL_ESTATE, "fsrl_state_restore before fsrl_state_snapshot");
    ENTER_DEV(c);
    hipStream_t st = c->compute;
    c->theta_version += 1;
    HIPCHK(hipMemcpyAsync(c->P, c->snap, (size_t)c->n_alloc * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(c->M, c->snap + c->n_alloc, (size_t)c->n_dev * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(c->V, c->snap + c->n_alloc + c->n_dev, (size_t)c->n_dev * 4, hipMemcpyDeviceToDevice, st));
    if (c->d_rms)
        HIPCHK(hipMemcpyAsync(c->d_rms, c->snap + c->n_alloc + 2 * (size_t)c->n_dev, 3 * FSRL_MAX_CRITICS * sizeof(double),
                              hipMemcpyDeviceToDevice, st));
    c->adam_t = c->snap_adam_t;
    tr_set_critic_steps(c, c->snap_critic_t);
    foc_steps(c, &c->snap_foc_a, &c->snap_foc_c, true);
    return 0;
}

// ------------------------------------------------------------------------------ store
static int flush_stage(fsrl_ctx* c) {
    Staging& s = c->stage[c->cur_stage];
    if (s.count == 0) return 0;
    const int k = s.count, Do = c->cfg.obs_dim, Da = c->cfg.act_dim;
    const size_t rec = stage_rec_bytes(Do, Da);
    HIPCHK(hipMemcpyAsync(s.d, s.h, (size_t)k * rec, hipMemcpyHostToDevice, c->side));      // the pinned side-stream ingest: ONE copy
    const int per = 2 * Do + Da + 1;
    const int blocks = std::min(1024, (k * per + 255) / 256);
    hipLaunchKernelGGL(store_scatter_kernel, dim3(blocks), dim3(256), 0, c->side, c->st, s.d, (int)rec, k, Do, Da);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s.done, c->side));
    s.in_flight = true;
    s.count = 0;
    for (EnvBook& e : c->env) e.staged = 0;
    c->cur_stage ^= 1;
    Staging& nx = c->stage[c->cur_stage];
    if (nx.in_flight) {  // the other buffer's copies must have left pinned memory before reuse
        HIPCHK(hipEventSynchronize(nx.done));
        nx.in_flight = false;
    }
    return 0;
}

extern "C" int fsrl_store_push(fsrl_ctx* c, const int32_t* env_ids, int32_t k, const float* obs,
                               const float* act, const double* rew, const double* cost,
                               const uint8_t* terminated, const uint8_t* truncated,
                               const float* obs_next, int64_t* ptr_out, double* ep_rew_out,
                               int32_t* ep_len_out, int64_t* ep_idx_out) {
    CHECK_ARG(c && env_ids && obs && act && rew && terminated && truncated && obs_next, "null argument");
    CHECK_ARG(k >= 0 && k <= c->active_envs, "k=%d rows but %d sub-buffers", k, c->active_envs);
    HIPCHK(hipSetDevice(c->device));      // keeps a resident actor alive (ENTER_DEV would end it)
    c->store_version += 1;
    const int Do = c->cfg.obs_dim, Da = c->cfg.act_dim;
    const size_t rec = stage_rec_bytes(Do, Da);
    for (int j = 0; j < k; ++j) {
        const int e = env_ids[j];
        CHECK_ARG(e >= 0 && e < c->active_envs, "buffer id %d out of range", e);
        // flush when the window is full -- or when this sub-buffer would wrap onto a slot that is
        // already in the window: the scatter kernel writes a window's rows in parallel, so one slot
        // must not appear twice in it (windows themselves are ordered on the side stream)
        if (c->stage[c->cur_stage].count == fsrl_ctx::STAGE_CAP || c->env[e].staged == c->sub_size) {
            int rc = flush_stage(c);
            if (rc) return rc;
        }
        Staging& s = c->stage[c->cur_stage];
        EnvBook& eb = c->env[e];
        eb.staged += 1;
        const int64_t ptr = eb.index;
        const int64_t gptr = ptr + (int64_t)e * c->sub_size;
        const int i = s.count++;
        const uint8_t fl = (uint8_t)((terminated[j] ? 1 : 0) | (truncated[j] ? 2 : 0));
        {
            uint8_t* r = s.h + (size_t)i * rec;
            const double rw = rew[j], cs = cost ? cost[j] : 0.0;
            const int32_t sl = (int32_t)gptr; const uint32_t f32 = fl;
            memcpy(r, &rw, 8); memcpy(r + 8, &cs, 8); memcpy(r + 16, &sl, 4); memcpy(r + 20, &f32, 4);
            memcpy(r + 24, obs + (size_t)j * Do, (size_t)Do * 4);
            memcpy(r + 24 + (size_t)Do * 4, obs_next + (size_t)j * Do, (size_t)Do * 4);
            memcpy(r + 24 + (size_t)Do * 8, act + (size_t)j * Da, (size_t)Da * 4);
        }
        c->h_flags[(size_t)gptr] = fl;
        // ReplayBuffer.add bookkeeping (ptr, ep_rew, ep_len, ep_idx)
        eb.last_index = ptr;
        eb.size = std::min(eb.size + 1, c->sub_size);
        eb.index = (eb.index + 1) % c->sub_size;
        eb.ep_rew += rew[j];
        eb.ep_len += 1;
        const bool done = fl != 0;
        if (ptr_out) ptr_out[j] = gptr;
        if (done) {
            if (ep_rew_out) ep_rew_out[j] = eb.ep_rew;
            if (ep_len_out) ep_len_out[j] = eb.ep_len;
            if (ep_idx_out) ep_idx_out[j] = eb.ep_idx + (int64_t)e * c->sub_size;
            eb.ep_rew = 0.0; eb.ep_len = 0; eb.ep_idx = eb.index;
        } else {
            if (ep_rew_out) ep_rew_out[j] = 0.0;
            if (ep_len_out) ep_len_out[j] = 0;
            if (ep_idx_out) ep_idx_out[j] = eb.ep_idx + (int64_t)e * c->sub_size;
        }
    }
    return 0;
}

extern "C" int fsrl_store_reset(fsrl_ctx* c, int keep_statistics) {
    CHECK_ARG(c, "null ctx");
    ENTER_DEV(c);
    int rc = flush_stage(c);   // rows already staged still land (then become unreachable)
    if (rc) return rc;
    // tianshou-0.5 ReplayBuffer.reset(keep_statistics): index / size / last_index go back to zero; the running episode's
    // reward, length and start index survive when keep_statistics is set (OnpolicyTrainer resets the buffer after every
    // update, fsrl/trainer/onpolicy.py:109, while episodes are still in flight)
    for (EnvBook& e : c->env) {
        EnvBook fresh;
        if (keep_statistics) { fresh.ep_rew = e.ep_rew; fresh.ep_len = e.ep_len; fresh.ep_idx = e.ep_idx; }
        e = fresh;
    }
    c->batch_ready = false;
    c->store_version += 1;
    return 0;
}

// VectorReplayBuffer(total_size, buffer_num) as the agents' learn() builds it (fsrl/agent/base_agent.py:279): re-cut the
// allocated store into buffer_num sub-buffers of ceil(total_size / buffer_num) rows.  The new geometry must fit the
// allocation made at fsrl_ctx_create (cfg.buffer_size rows, cfg.env_num sub-buffers); the store is emptied.
extern "C" int fsrl_store_configure(fsrl_ctx* c, int64_t total_size, int32_t buffer_num) {
    CHECK_ARG(c, "null ctx");
    CHECK_ARG(buffer_num >= 1 && buffer_num <= c->cfg.env_num, "buffer_num %d: the context has %d sub-buffers", buffer_num,
              c->cfg.env_num);
    CHECK_ARG(total_size >= buffer_num, "total_size must be >= buffer_num");
    const int64_t sub = (total_size + buffer_num - 1) / buffer_num;
    CHECK_ARG(sub * buffer_num <= c->alloc_rows,
              "a store of %lld x %d rows does not fit the %lld rows allocated at fsrl_ctx_create (buffer_size)",
              (long long)sub, buffer_num, (long long)c->alloc_rows);
    CHECK_ARG(!c->in_update, "fsrl_store_configure inside an update");
    int rc = fsrl_store_reset(c, 0);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->side));
    c->sub_size = sub;
    c->maxsize = sub * buffer_num;
    c->active_envs = buffer_num;
    std::fill(c->h_flags.begin(), c->h_flags.end(), (uint8_t)0);
    return 0;
}
extern "C" int fsrl_store_geometry(const fsrl_ctx* c, int64_t* sub_size_out, int32_t* buffer_num_out) {
    CHECK_ARG(c && sub_size_out && buffer_num_out, "null argument");
    *sub_size_out = c->sub_size; *buffer_num_out = c->active_envs;
    return 0;
}

extern "C" int64_t fsrl_store_len(const fsrl_ctx* c) {
    int64_t n = 0;
    if (c) for (const EnvBook& e : c->env) n += e.size;
    return n;
}

// sample_indices(0): concat over sub-buffers of [index, size) ++ [0, index)  (+offset)
static int64_t sample0(const fsrl_ctx* c, int* idx, uint8_t* endf) {
    int64_t n = 0;
    for (int e = 0; e < c->cfg.env_num; ++e) {
        const EnvBook& eb = c->env[e];
        const int64_t off = (int64_t)e * c->sub_size;
        const int64_t first = n;
        for (int64_t i = eb.index; i < eb.size; ++i) idx[n++] = (int)(off + i);
        for (int64_t i = 0; i < eb.index; ++i) idx[n++] = (int)(off + i);
        if (endf) {
            for (int64_t r = first; r < n; ++r) endf[r] = c->h_flags[(size_t)idx[r]] != 0;
            // unfinished_index(): the last written row of a sub-buffer is always the final row
            // of its sample(0) range; if it is not done the GAE scan must still stop there
            if (n > first) endf[n - 1] = 1;
        }
    }
    return n;
}

extern "C" int fsrl_store_sample0(fsrl_ctx* c, int64_t* out, int64_t cap, int64_t* n_out) {
    CHECK_ARG(c && n_out, "null argument");
    const int64_t n = sample0(c, c->h_indices, nullptr);
    *n_out = n;
    if (out) {
        CHECK_ARG(cap >= n, "indices_out too small (%lld < %lld)", (long long)cap, (long long)n);
        for (int64_t i = 0; i < n; ++i) out[i] = c->h_indices[i];
    }
    return 0;
}

// buffer[indices] (tianshou ReplayBufferManager.__getitem__): the stored rows at the given slots, copied to the host.
// Not on the training path (the update gathers on the device); evaluation tooling and the parity tests read rows back.
extern "C" int fsrl_store_read(fsrl_ctx* c, const int64_t* indices, int64_t n, float* obs_out, float* act_out,
                               double* rew_out, double* cost_out, uint8_t* terminated_out, uint8_t* truncated_out,
                               float* obs_next_out) {
    CHECK_ARG(c && (indices || n == 0), "null argument");
    CHECK_ARG(n >= 0 && n <= c->maxsize, "bad row count");
    if (n == 0) return 0;
    ENTER_DEV(c);
    int rc = join_store(c);
    if (rc) return rc;
    const size_t N = (size_t)n, Do = (size_t)c->cfg.obs_dim, Da = (size_t)c->cfg.act_dim;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_idx = 0, o_end = o_idx + al(N * 4), o_obs = o_end + al(N), o_nxt = o_obs + al(N * Do * 4),
                 o_act = o_nxt + al(N * Do * 4), o_rew = o_act + al(N * Da * 4), o_cost = o_rew + al(N * 8),
                 o_fl = o_cost + al(N * 8), total = o_fl + al(N);
    rc = ensure_scratch(c, total);
    if (rc) return rc;
    std::vector<int> idx(N);
    for (size_t i = 0; i < N; ++i) {
        CHECK_ARG(indices[i] >= 0 && indices[i] < c->maxsize, "index %lld out of range", (long long)indices[i]);
        idx[i] = (int)indices[i];
    }
    char* base = (char*)c->scratch;
    hipStream_t s = c->compute;
    HIPCHK(hipMemcpyAsync(base + o_idx, idx.data(), N * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(base + o_end, 0, N, s));
    BatchPtrs b{(float*)(base + o_obs), (float*)(base + o_nxt), (float*)(base + o_act), (double*)(base + o_rew),
                (double*)(base + o_cost), (uint8_t*)(base + o_fl)};
    const size_t work = N * (2 * Do + Da + 1);
    hipLaunchKernelGGL(batch_gather_kernel, dim3((int)std::min<size_t>(2048, (work + 255) / 256)), dim3(256), 0, s, c->st, b,
                       (const int*)(base + o_idx), (const uint8_t*)(base + o_end), (int)n, (int)Do, (int)Da);
    HIPCHK(hipGetLastError());
    std::vector<uint8_t> fl(N);
    if (obs_out) HIPCHK(hipMemcpyAsync(obs_out, b.obs, N * Do * 4, hipMemcpyDeviceToHost, s));
    if (obs_next_out) HIPCHK(hipMemcpyAsync(obs_next_out, b.obs_next, N * Do * 4, hipMemcpyDeviceToHost, s));
    if (act_out) HIPCHK(hipMemcpyAsync(act_out, b.act, N * Da * 4, hipMemcpyDeviceToHost, s));
    if (rew_out) HIPCHK(hipMemcpyAsync(rew_out, b.rew, N * 8, hipMemcpyDeviceToHost, s));
    if (cost_out) HIPCHK(hipMemcpyAsync(cost_out, b.cost, N * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(fl.data(), b.flags, N, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (size_t i = 0; i < N; ++i) {
        if (terminated_out) terminated_out[i] = fl[i] & 1;
        if (truncated_out) truncated_out[i] = (fl[i] >> 1) & 1;
    }
    return 0;
}

// ------------------------------------------------------------------------------ launch helpers
template <typename F>
static int dispatch_H(int H, F&& f) {
    switch (H) {
        case 64: return f(std::integral_constant<int, 64>());
        case 128: return f(std::integral_constant<int, 128>());
        case 256: return f(std::integral_constant<int, 256>());
    }
    return fail(FSRL_EINVAL, "unsupported hidden width %d", H);
}

static int launch_infer(fsrl_ctx* c, const InferArgs& ia, int jobs_y, hipStream_t s) {
    const int tiles = (ia.N + 15) / 16;
    if (tiles == 0) return 0;
    if (c->lay) return lay_infer(c, ia, jobs_y, s);
    return dispatch_H(c->cfg.hidden, [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        // persistent over tiles: about one workgroup per CU in total (their LDS footprint allows no more)
        const int gx = std::min(tiles, std::max(1, c->n_cus / jobs_y));
        hipLaunchKernelGGL(mlp_infer_kernel<H>, dim3(gx, jobs_y), dim3(4 * H), 0, s, c->P, c->md, ia);
        HIPCHK(hipGetLastError());
        return 0;
    });
}

// ------------------------------------------------------------------------------ actor forward
// Actor evaluation for the collector, in two halves so that host work can overlap the round trip:
//   actor_eval_launch: copy the k observations into pinned host memory and launch the actor on them -- the kernel
//     reads the observations from, and writes its head outputs to, pinned host memory (zero-copy: a few hundred
//     bytes over PCIe), nothing else is staged;
//   actor_eval_finish: wait, then turn the head outputs into (mean, std) of the policy's Gaussian.
// PPO-family contexts: mu = max_action * tanh(head) (mlp_infer_kernel applies it), sigma = exp(sigma_param).
// Replay contexts (SAC / DDPG / CVPO): raw head outputs [mu | log sigma], see actor_eval_finish.
struct SacState;
static bool ctx_is_replay(const fsrl_ctx* c) { return c->cfg.algo == FSRL_ALGO_SAC_LAG; }
static int sac_actor_launch(fsrl_ctx* c, const float* h_obs, float* h_raw, int k);                 // defined with the SAC code
static void sac_actor_finish(fsrl_ctx* c, const float* h_raw, int k, float* mu_out, float* sigma_out);
static bool sac_squashes(fsrl_ctx* c);
static bool sac_actor_resident_args(fsrl_ctx* c, const float** P, const ModelDesc** md);   // false: no fused actor (layered / not initialised)

// ---- the resident actor (actor_resident_kernel, kernels_mlp.hpp).  Protocol, host side:
//   post:    [wait until every workgroup of every earlier generation has reported its end] -> launch generation g if none is live ->
//            write the k observations -> ring the doorbell (k << 32 | seq, one release store);
//   finish:  spin on `done[b] == seq` of the request's tiles; if a `state[b] == g` shows up instead (a workgroup ended by its idle
//            timeout just before the doorbell), tell the rest to end, wait for them, launch generation g + 1 and ring again;
//   release: doorbell = EXIT; nothing is waited for (the stream orders what follows behind the kernel).
// A doorbell is only ever rung when generation pa_gen is the one kernel that can hear it.
struct PaLayout { unsigned long long* bell; unsigned* done; unsigned* state; float* obs; float* mu; float* sp; };
static PaLayout pa_layout(const fsrl_ctx* c) {
    char* b = (char*)c->h_pa;
    PaLayout l;
    l.bell = (unsigned long long*)b; l.done = (unsigned*)(b + 16); l.state = (unsigned*)(b + 32);
    l.obs = (float*)(b + 64);
    l.mu = l.obs + (size_t)c->pa_cap * c->cfg.obs_dim;
    l.sp = l.mu + (size_t)c->pa_cap * 2 * c->cfg.act_dim;        // replay contexts: [mu | log sigma] per row
    return l;
}

static bool pactor_ok(const fsrl_ctx* c, int k) {
    const int blocks = std::min(PACTOR_BLOCKS, std::max(1, (c->cfg.env_num + 15) / 16));
    if (!(c->pa_on && !c->no_spin && !c->lay && !c->group && k >= 1 && k <= 16 * blocks)) return false;
    if (c->cfg.algo != FSRL_ALGO_SAC_LAG) return true;
    const float* P; const ModelDesc* md;
    return sac_actor_resident_args(const_cast<fsrl_ctx*>(c), &P, &md);      // replay contexts: their fused actor network
}

static void pactor_release(fsrl_ctx* c) {
    if (!c->pa_live) return;
    const PaLayout l = pa_layout(c);
    c->pa_seq += 1;
    __atomic_store_n(l.bell, ((unsigned long long)PACTOR_EXIT << 32) | c->pa_seq, __ATOMIC_RELEASE);
    c->pa_live = false;
}

// how many workgroups of generation pa_gen have ended
static int pactor_ended_count(const fsrl_ctx* c) {
    const PaLayout l = pa_layout(c);
    int n = 0;
    for (int b = 0; b < c->pa_blocks; ++b) n += __atomic_load_n(l.state + b, __ATOMIC_ACQUIRE) == c->pa_gen;
    return n;
}

// generation pa_gen has ended (true at once if none was ever launched); waits for a kernel that was told to end, never for a live one
static bool pactor_ended(fsrl_ctx* c, bool wait) {
    if (c->pa_gen == 0) return true;
    for (long spins = 0;; ++spins) {
        if (pactor_ended_count(c) == c->pa_blocks) return true;
        if (!wait) return false;
        if (spins > 4000000) { (void)hipStreamSynchronize(c->compute); return pactor_ended_count(c) == c->pa_blocks; }
        __builtin_ia32_pause();
    }
}

static int pactor_launch(fsrl_ctx* c, unsigned last_seq) {
    const PaLayout l = pa_layout(c);
    PActorArgs a{};
    a.obs = l.obs; a.mu_out = l.mu; a.sigma_param_out = l.sp; a.bell = l.bell; a.done = l.done; a.state = l.state;
    c->pa_gen += 1;
    if (c->pa_gen == 0) c->pa_gen = 1;
    a.gen = c->pa_gen; a.last_seq = last_seq; a.max_action = c->cfg.max_action;
    a.timeout_ticks = (unsigned long long)(c->pa_idle_us * 100.0);            // wall_clock64: 100 MHz
    c->pa_blocks = std::min(PACTOR_BLOCKS, std::max(1, (c->cfg.env_num + 15) / 16));
    const float* P = c->P; const ModelDesc* md = &c->md;
    if (c->cfg.algo == FSRL_ALGO_SAC_LAG) {                                    // the replay agents' actor: raw head outputs, as sac_actor_launch
        if (!sac_actor_resident_args(c, &P, &md)) return fail(FSRL_ESTATE, "no fused actor network");
        a.raw_cols = 2 * c->cfg.act_dim; a.max_action = 1.0f;
    }
    const ModelDesc mdv = *md;
    const int rc = dispatch_H(c->cfg.hidden, [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        if (a.raw_cols > 0) hipLaunchKernelGGL((actor_resident_kernel<H, true>), dim3(c->pa_blocks), dim3(4 * H), 0, c->compute, P, mdv, a);
        else hipLaunchKernelGGL((actor_resident_kernel<H, false>), dim3(c->pa_blocks), dim3(4 * H), 0, c->compute, P, mdv, a);
        HIPCHK(hipGetLastError());
        return 0;
    });
    if (rc) return rc;
    c->pa_live = true; c->pa_launches += 1;
    return 0;
}

// ring the doorbell for the k rows already in place (launching a kernel first if none can hear it)
static int pactor_ring(fsrl_ctx* c, int k) {
    const PaLayout l = pa_layout(c);
    if (c->pa_live && pactor_ended_count(c) > 0) pactor_release(c);          // (some of) it ended by its idle timeout: the rest follows
    c->pa_seq += 1;
    if (!c->pa_live) {
        if (!pactor_ended(c, true)) return fail(FSRL_EHIP, "the resident actor did not end");
        const int rc = pactor_launch(c, c->pa_seq - 1);
        if (rc) return rc;
    }
    __atomic_store_n(l.bell, ((unsigned long long)(unsigned)k << 32) | c->pa_seq, __ATOMIC_RELEASE);
    return 0;
}

static int pactor_post(fsrl_ctx* c, const float* obs, int k) {
    const int Do = c->cfg.obs_dim, Da = c->cfg.act_dim;
    if (!c->h_pa) {
        c->pa_cap = 16 * PACTOR_BLOCKS;
        const size_t bytes = 64 + ((size_t)c->pa_cap * (Do + 2 * Da) + FSRL_MAX_ACT) * 4;
        HIPCHK(hipHostMalloc(&c->h_pa, bytes));
        memset(c->h_pa, 0, bytes);
    }
    memcpy(pa_layout(c).obs, obs, (size_t)k * Do * 4);
    const int rc = pactor_ring(c, k);
    if (rc) return rc;
    c->actor_k = k; c->pa_req = true; c->pa_requests += 1;
    return 0;
}

static int pactor_wait(fsrl_ctx* c) {
    const PaLayout l = pa_layout(c);
    const int tiles = (c->actor_k + 15) / 16;
    auto served = [&]() {
        for (int b = 0; b < tiles; ++b)
            if (__atomic_load_n(l.done + b, __ATOMIC_ACQUIRE) != c->pa_seq) return false;
        return true;
    };
    for (long spins = 0;; ++spins) {
        if (served()) return 0;
        if ((spins & 255) == 255 && pactor_ended_count(c) > 0) {
            // a workgroup is gone (idle timeout just before the doorbell) -- unless it served the request first
            if (served()) return 0;
            const int rc = pactor_ring(c, c->actor_k);
            if (rc) return rc;
        }
        if (spins > 500000000L) return fail(FSRL_EHIP, "the resident actor does not answer");
        __builtin_ia32_pause();
    }
}

static int actor_eval_launch(fsrl_ctx* c, const float* obs, int32_t k, bool want_sigma) {
    const int Do = c->cfg.obs_dim, Da = c->cfg.act_dim;
    c->pa_req = false;
    if (want_sigma && pactor_ok(c, k)) return pactor_post(c, obs, k);
    pactor_release(c);                       // a launch behind a live resident kernel would wait for its idle timeout
    // pinned staging [obs | head outputs (2*Da per row) | sigma_param]
    const size_t ob = (size_t)k * Do * 4, mb = (size_t)k * 2 * Da * 4;
    const size_t need = ob + mb + FSRL_MAX_ACT * 4;
    if (c->h_actor_bytes < need) {
        HIPCHK(hipStreamSynchronize(c->compute));
        if (c->h_actor) HIPCHK(hipHostFree(c->h_actor));
        c->h_actor = nullptr; c->h_actor_bytes = 0;
        HIPCHK(hipHostMalloc(&c->h_actor, need * 2));
        c->h_actor_bytes = need * 2;
    }
    float* h_obs = (float*)c->h_actor;
    float* h_mu = (float*)((char*)c->h_actor + ob);
    float* h_sp = (float*)((char*)c->h_actor + ob + mb);
    c->actor_k = k; c->actor_ob = ob; c->actor_mb = mb;
    // completion words: one per 16-row tile, written by the kernel after its outputs; actor_eval_finish spins on them
    // (a stream synchronisation costs several microseconds more than the kernel itself at these sizes)
    c->actor_blocks = (k + 15) / 16;
    if (c->actor_blocks > c->done_cap) {
        if (c->h_done) HIPCHK(hipHostFree(c->h_done));
        c->h_done = nullptr;
        c->done_cap = std::max(2 * c->actor_blocks, 64);
        HIPCHK(hipHostMalloc(&c->h_done, (size_t)c->done_cap * 4));
        memset(c->h_done, 0, (size_t)c->done_cap * 4);
    }
    c->actor_seq += 1;
    if (c->actor_seq == 0) c->actor_seq = 1;
    if (k > 0) memcpy(h_obs, obs, ob);
    if (ctx_is_replay(c)) return k > 0 ? sac_actor_launch(c, h_obs, h_mu, k) : 0;
    if (k > 0) {
        InferArgs ia{};
        ia.obs = h_obs; ia.obs_next = h_obs; ia.act = nullptr; ia.flags = nullptr; ia.values = nullptr;
        ia.vnext = nullptr; ia.logp_old = nullptr; ia.mu_out = h_mu; ia.N = k; ia.C = 0;
        ia.max_action = c->cfg.max_action; ia.sigma_param_out = want_sigma ? h_sp : nullptr;
        ia.done = c->h_done; ia.seq = c->actor_seq;
        return launch_infer(c, ia, 1, c->compute);   // job 0 == 2*C == actor
    }
    if (want_sigma)
        HIPCHK(hipMemcpyAsync(h_sp, c->P + c->md.net[0].sigma, (size_t)Da * 4, hipMemcpyDeviceToHost, c->compute));
    return 0;
}

static int actor_eval_finish(fsrl_ctx* c, float* mu_out, float* sigma_out) {
    const int Da = c->cfg.act_dim, k = c->actor_k;
    if (c->pa_req) {                                    // served by the resident kernel
        c->pa_req = false;
        const int rc = pactor_wait(c);
        if (rc) return rc;
        const PaLayout l = pa_layout(c);
        if (ctx_is_replay(c)) { sac_actor_finish(c, l.mu, k, mu_out, sigma_out); return 0; }
        memcpy(mu_out, l.mu, (size_t)k * Da * 4);
        if (sigma_out)
            for (int r = 0; r < k; ++r)
                for (int d = 0; d < Da; ++d) sigma_out[(size_t)r * Da + d] = expf(l.sp[d]);
        return 0;
    }
    bool landed = false;
    if (k > 0 && !c->no_spin) {                        // spin on the kernel's completion words (bounded), else synchronise
        landed = true;
        for (int b = 0; b < c->actor_blocks && landed; ++b) {
            int spins = 0;
            while (__atomic_load_n(&c->h_done[b], __ATOMIC_ACQUIRE) != c->actor_seq) {
                if (++spins > 2000000) { landed = false; break; }
                __builtin_ia32_pause();
            }
        }
    }
    if (!landed) HIPCHK(hipStreamSynchronize(c->compute));
    const float* h_mu = (const float*)((char*)c->h_actor + c->actor_ob);
    const float* h_sp = (const float*)((char*)c->h_actor + c->actor_ob + c->actor_mb);
    if (ctx_is_replay(c)) { if (k > 0) sac_actor_finish(c, h_mu, k, mu_out, sigma_out); return 0; }
    if (k > 0) memcpy(mu_out, h_mu, (size_t)k * Da * 4);
    if (sigma_out) {
        for (int r = 0; r < k; ++r)
            for (int d = 0; d < Da; ++d) sigma_out[(size_t)r * Da + d] = expf(h_sp[d]);
    }
    return 0;
}

extern "C" int fsrl_actor_forward(fsrl_ctx* c, const float* obs, int32_t k, float* mu_out,
                                  float* sigma_out) {
    CHECK_ARG(c && obs && mu_out, "null argument");
    CHECK_ARG(k >= 0, "negative row count");
    CHECK_ARG(!ctx_is_replay(c), "replay contexts: fsrl_sac_actor_forward");
    HIPCHK(hipSetDevice(c->device));      // keeps a resident actor alive (ENTER_DEV would end it)
    int rc = actor_eval_launch(c, obs, k, sigma_out != nullptr);
    if (rc) return rc;
    return actor_eval_finish(c, mu_out, sigma_out);
}

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t xoshiro_next(uint64_t* s) {
    const uint64_t result = rotl64(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl64(s[3], 45);
    return result;
}

// Collector-time action sampling (fsrl/data/fast_collector.py:283-300: policy.forward -> dist.sample()):
// the actor runs on the device, the k x Da Gaussian draws come from the library's xoshiro256** stream.
//   PPO / CPO / TRPO contexts: a = mu + exp(sigma_param) * eps          (ppo_lag / cpo forward: Independent(Normal))
//   SAC contexts:              a = tanh(mu + sigma(s) * eps)             (sac_lag.py:155-183)
// deterministic != 0 returns the mean (tanh(mean) for SAC).  Noise is NOT torch's stream: callers that
// need the reference's random numbers keep the host mirror of the actor (fsrl_amd/policy).
// second half of fsrl_actor_sample / fsrl_collect_step: wait for the actor, then a = mean (+ std * N(0,1))
static int actor_sample_finish(fsrl_ctx* c, int32_t deterministic, float* act_out) {
    const int Da = c->cfg.act_dim, k = c->actor_k;
    c->act_mu.resize((size_t)k * Da); c->act_sg.resize((size_t)k * Da);
    int rc = actor_eval_finish(c, c->act_mu.data(), c->act_sg.data());
    if (rc) return rc;
    const bool squash = ctx_is_replay(c) && sac_squashes(c);
    for (size_t i = 0; i < c->act_mu.size(); ++i) {
        float u = c->act_mu[i];
        if (!deterministic) {
            // Box-Muller on two 53-bit uniforms of the context's stream
            double u1 = (double)(xoshiro_next(c->rng) >> 11) * (1.0 / 9007199254740992.0);
            const double u2 = (double)(xoshiro_next(c->rng) >> 11) * (1.0 / 9007199254740992.0);
            if (u1 < 1e-300) u1 = 1e-300;
            u += c->act_sg[i] * (float)(std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2));
        }
        act_out[i] = squash ? std::tanh(u) : u;     // DDPG-Lag / CVPO: mu is already max_action * tanh
    }
    return 0;
}

extern "C" int fsrl_actor_sample(fsrl_ctx* c, const float* obs, int32_t k, int32_t deterministic, uint64_t seed,
                                 float* act_out) {
    CHECK_ARG(c && obs && act_out, "null argument");
    CHECK_ARG(k >= 0, "negative row count");
    if (k == 0) return 0;
    HIPCHK(hipSetDevice(c->device));      // keeps a resident actor alive (ENTER_DEV would end it)
    if (seed) { c->rng[2] ^= seed; c->rng[3] += seed * 0x9E3779B97F4A7C15ull; }
    int rc = actor_eval_launch(c, obs, k, true);
    if (rc) return rc;
    return actor_sample_finish(c, deterministic, act_out);
}

// One vector step of FastCollector.collect with the actor on the device (fsrl/data/fast_collector.py:283-368) in ONE call:
// launch the actor on the observations the next actions are for; while it runs store the transitions that just
// finished (fsrl_store_push, bookkeeping included); wait; draw the noise; map the action to the env's range
// (BasePolicy.map_action, base_policy.py:226-256).  Same random stream and the same results as fsrl_actor_sample +
// fsrl_store_push called one after the other.
extern "C" int fsrl_collect_step(fsrl_ctx* c, const int32_t* env_ids, int32_t k, const float* obs, const float* act,
                                 const double* rew, const double* cost, const uint8_t* terminated,
                                 const uint8_t* truncated, const float* obs_next, int64_t* ptr_out, double* ep_rew_out,
                                 int32_t* ep_len_out, int64_t* ep_idx_out, const float* obs_act, int32_t k_act,
                                 int32_t deterministic, int32_t bound_method, const float* act_low,
                                 const float* act_high, float* act_out, float* env_act_out) {
    CHECK_ARG(c, "null ctx");
    CHECK_ARG(k >= 0 && k_act >= 0, "negative row count");
    CHECK_ARG(k_act == 0 || (obs_act && act_out), "obs_act / act_out missing");
    CHECK_ARG(bound_method >= 0 && bound_method <= 2, "bound_method: 0 none, 1 clip, 2 tanh");
    CHECK_ARG((act_low == nullptr) == (act_high == nullptr), "act_low and act_high are given together");
    HIPCHK(hipSetDevice(c->device));      // keeps a resident actor alive (ENTER_DEV would end it)
    int rc = 0;
    if (k_act > 0) {
        rc = actor_eval_launch(c, obs_act, k_act, true);
        if (rc) return rc;
    }
    if (k > 0) {
        rc = fsrl_store_push(c, env_ids, k, obs, act, rew, cost, terminated, truncated, obs_next, ptr_out, ep_rew_out,
                             ep_len_out, ep_idx_out);
        if (rc) { if (k_act > 0) (void)hipStreamSynchronize(c->compute); return rc; }
    }
    if (k_act == 0) return 0;
    rc = actor_sample_finish(c, deterministic, act_out);
    if (rc) return rc;
    if (env_act_out) {
        const int Da = c->cfg.act_dim;
        for (int r = 0; r < k_act; ++r)
            for (int d = 0; d < Da; ++d) {
                float a = act_out[(size_t)r * Da + d];
                if (bound_method == 1) a = std::min(std::max(a, -1.0f), 1.0f);
                else if (bound_method == 2) a = std::tanh(a);
                if (act_low) a = act_low[d] + (act_high[d] - act_low[d]) * (a + 1.0f) / 2.0f;
                env_act_out[(size_t)r * Da + d] = a;
            }
    }
    return 0;
}

extern "C" int fsrl_store_sizes(const fsrl_ctx* c, int64_t* sizes_out, int32_t n) {
    CHECK_ARG(c && sizes_out && n >= 0 && n <= c->cfg.env_num, "bad argument");
    for (int e = 0; e < n; ++e) sizes_out[e] = c->env[(size_t)e].size;
    return 0;
}

// ------------------------------------------------------------------------------ GAE (standalone)
extern "C" int fsrl_gae_return(fsrl_ctx* c, const float* v, const float* v_next, const double* rew,
                               const uint8_t* end_flag, int64_t n, double gamma, double gae_lambda,
                               double* adv_out) {
    CHECK_ARG(c && adv_out, "null argument");
    CHECK_ARG(n >= 0 && n < INT_MAX / 4, "bad n");
    CHECK_ARG(gae_lambda >= 0.0 && gae_lambda <= 1.0, "GAE lambda should be in [0, 1].");
    if (n == 0) return 0;
    CHECK_ARG(v && v_next && rew && end_flag, "null argument");
    ENTER_DEV(c);
    // segments: cut after every end_flag (disc = 0 there, so the scan restarts)
    std::vector<int> seg{0};
    std::vector<uint8_t> fl((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        fl[(size_t)i] = end_flag[i] ? 4 : 0;
        if (end_flag[i] && i + 1 < n) seg.push_back((int)i + 1);
    }
    seg.push_back((int)n);
    const int nseg = (int)seg.size() - 1;
    const size_t N = (size_t)n;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_v = 0, o_vn = o_v + al(N * 4), o_rew = o_vn + al(N * 4), o_fl = o_rew + al(N * 8),
                 o_seg = o_fl + al(N), o_adv = o_seg + al(seg.size() * 4), o_ret = o_adv + al(N * 4),
                 o_a64 = o_ret + al(N * 4), total = o_a64 + al(N * 8);
    int rc = ensure_scratch(c, total);
    if (rc) return rc;
    char* base = (char*)c->scratch;
    hipStream_t s = c->compute;
    HIPCHK(hipMemcpyAsync(base + o_v, v, N * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(base + o_vn, v_next, N * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(base + o_rew, rew, N * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(base + o_fl, fl.data(), N, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(base + o_seg, seg.data(), seg.size() * 4, hipMemcpyHostToDevice, s));
    GaeArgs ga{};
    ga.values = (float*)(base + o_v); ga.vnext = (float*)(base + o_vn); ga.rew = (double*)(base + o_rew);
    ga.cost = ga.rew; ga.flags = (uint8_t*)(base + o_fl); ga.seg_start = (int*)(base + o_seg);
    ga.advs = (float*)(base + o_adv); ga.rets = (float*)(base + o_ret); ga.adv64 = (double*)(base + o_a64);
    ga.N = (int)n; ga.gamma = gamma; ga.gl = gamma * gae_lambda;
    hipLaunchKernelGGL(gae_kernel, dim3(nseg, 1), dim3(64), 0, s, ga);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(adv_out, base + o_a64, N * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// ------------------------------------------------------------------------------ n-step return (standalone)
// nstep_return (base_policy.py:543-567) on the device: the twin of fsrl_gae_return.  Host arrays in, host array out.
extern "C" int fsrl_nstep_return(fsrl_ctx* c, const double* metric, const uint8_t* end_flag, int64_t len,
                                 const float* target_q, const int64_t* indices, int64_t bsz, int64_t q, double gamma,
                                 int32_t n_step, double* out) {
    CHECK_ARG(c && out, "null argument");
    CHECK_ARG(n_step >= 1, "n_step should be greater than 0");          /* base_policy.py:472 */
    CHECK_ARG(bsz >= 0 && q >= 1 && len >= 0 && bsz < INT_MAX / 8 && q < 65536, "bad sizes");
    if (bsz == 0) return 0;
    CHECK_ARG(metric && end_flag && target_q && indices && len > 0, "null argument");
    for (int64_t i = 0; i < (int64_t)n_step * bsz; ++i)
        CHECK_ARG(indices[i] >= 0 && indices[i] < len, "index out of range");
    ENTER_DEV(c);
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t L = (size_t)len, B = (size_t)bsz, Q = (size_t)q;
    const size_t o_m = 0, o_e = o_m + al(L * 8), o_t = o_e + al(L), o_i = o_t + al(B * Q * 4),
                 o_o = o_i + al((size_t)n_step * B * 8), total = o_o + al(B * Q * 8);
    int rc = ensure_scratch(c, total);
    if (rc) return rc;
    char* base = (char*)c->scratch;
    hipStream_t s = c->compute;
    HIPCHK(hipMemcpyAsync(base + o_m, metric, L * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(base + o_e, end_flag, L, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(base + o_t, target_q, B * Q * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(base + o_i, indices, (size_t)n_step * B * 8, hipMemcpyHostToDevice, s));
    NstepArgs na{};
    na.metric = (const double*)(base + o_m); na.end_flag = (const uint8_t*)(base + o_e);
    na.target_q = (const float*)(base + o_t); na.indices = (const int64_t*)(base + o_i);
    na.out = (double*)(base + o_o); na.len = len; na.bsz = (int)bsz; na.q = (int)q; na.n_step = n_step; na.gamma = gamma;
    hipLaunchKernelGGL(nstep_return_kernel, dim3((unsigned)((bsz + 255) / 256)), dim3(256), 0, s, na);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, base + o_o, B * Q * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

#include "host_collect.inc"

#include "host_ppo.inc"

#include "host_layered.inc"

#include "host_group.inc"

// abandon an update that began with fsrl_ppo_begin and cannot reach fsrl_ppo_end (an exception between the calls on the
// caller's side): drain the stream, clear the state machine.  No-op outside an update.
extern "C" int fsrl_ppo_abort(fsrl_ctx* c) {
    CHECK_ARG(c, "null ctx");
    ENTER_DEV(c);
    if (c->in_update) (void)hipStreamSynchronize(c->compute);
    c->in_update = false; c->verdict_pending = false;
    return 0;
}

// ------------------------------------------------------------------------------ timing
extern "C" int fsrl_set_profiling(fsrl_ctx* c, int enable) {
    CHECK_ARG(c, "null ctx");
    c->profiling = enable != 0;
    return 0;
}
extern "C" int fsrl_last_timing(fsrl_ctx* c, double* out, int32_t n) {
    CHECK_ARG(c && out && n >= 4, "need room for 4 doubles");
    out[0] = c->t_process_ms; out[1] = c->t_learn_ms; out[2] = c->t_fwdbwd_ms; out[3] = (double)c->n_fwdbwd;
    if (n >= 5) out[4] = c->t_fwdbwd_raw_ms;
    return 0;
}

// ------------------------------------------------------------------------------ launch floors (measured, bench.py)
// What no kernel work can remove from one PPO optimiser step: its three dependent launches.  Three EMPTY kernels with the grids,
// block sizes and LDS footprints of the step's kernels (fused forward/backward, weight gradients, clip + Adam) for a minibatch
// of `mb_rows` rows run `iters` times back to back on the compute stream, bracketed by events: out_us[0..2] = one launch of each
// grid behind itself, out_us[3] = the triple behind itself (what a step's launches cost with nothing in them).
template <int NT>
__global__ __launch_bounds__(NT) void floor_kernel(int* sink) {
    extern __shared__ float floor_lds[];
    if (sink && threadIdx.x == 4095) sink[0] = (int)floor_lds[0];          // never true: keeps the LDS allocation alive
}
extern "C" int fsrl_launch_floors(fsrl_ctx* c, int32_t mb_rows, int32_t iters, double* out_us) {
    CHECK_ARG(c && out_us && mb_rows >= 1 && iters >= 1 && iters <= 100000, "bad argument");
    CHECK_ARG(!c->lay, "launch floors describe the fused three-launch step, not a layered context");
    ENTER_DEV(c);
    const int H = c->cfg.hidden, nn = c->md.n_nets;
    const int tiles = (mb_rows + 15) / 16;
    const bool rows4 = tiles * 4 * nn <= c->n_cus, rows8 = !rows4 && tiles * 2 * nn <= c->n_cus;
    const int nt_ = rows4 ? tiles * 4 : rows8 ? tiles * 2 : tiles;
    const int g_fb = (nn <= 4 && !c->no_xcd_pair) ? 8 * ((nt_ + 1) / 2) : nt_ * nn;      // the step's own grid (one network per XCD pair)
    const int g_wg = wg_grid(H, nn);
    const int g_ad = (c->n_dev + 4 * ADAM_NT - 1) / (4 * ADAM_NT);
    size_t lds_fb = 0, lds_wg = 0;
    int rc = dispatch_H(H, [&](auto hc) {
        constexpr int HH = decltype(hc)::value;
        hipFuncAttributes fa{};
        const void* f = rows4 ? (const void*)ppo_fwd_bwd_kernel<HH, 4> : rows8 ? (const void*)ppo_fwd_bwd_kernel<HH, 8>
                                                                              : (const void*)ppo_fwd_bwd_kernel<HH, 16>;
        HIPCHK(hipFuncGetAttributes(&fa, f));
        lds_fb = fa.sharedSizeBytes;
        HIPCHK(hipFuncGetAttributes(&fa, (const void*)ppo_wgrad_kernel<HH, false, false>));
        lds_wg = fa.sharedSizeBytes;
        return 0;
    });
    if (rc) return rc;
    const int nt_fb = 4 * H;
    auto launch = [&](int which) {
        hipStream_t s = c->compute;
        if (which == 0) {
            if (nt_fb == 1024) hipLaunchKernelGGL(floor_kernel<1024>, dim3(g_fb), dim3(1024), lds_fb, s, (int*)nullptr);
            else if (nt_fb == 512) hipLaunchKernelGGL(floor_kernel<512>, dim3(g_fb), dim3(512), lds_fb, s, (int*)nullptr);
            else hipLaunchKernelGGL(floor_kernel<256>, dim3(g_fb), dim3(256), lds_fb, s, (int*)nullptr);
        } else if (which == 1) hipLaunchKernelGGL(floor_kernel<1024>, dim3(g_wg), dim3(1024), lds_wg, s, (int*)nullptr);
        else hipLaunchKernelGGL(floor_kernel<ADAM_NT>, dim3(g_ad), dim3(ADAM_NT), 0, s, (int*)nullptr);
    };
    const size_t lds_max = std::max(lds_fb, lds_wg);
    HIPCHK(hipFuncSetAttribute((const void*)floor_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    HIPCHK(hipFuncSetAttribute((const void*)floor_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    HIPCHK(hipFuncSetAttribute((const void*)floor_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int m = 0; m < 4; ++m) {
        for (int w = 0; w < 8; ++w) launch(m < 3 ? m : w % 3);                       // warm the code objects
        HIPCHK(hipEventRecord(e0, c->compute));
        for (int i = 0; i < iters; ++i) {
            if (m < 3) launch(m);
            else { launch(0); launch(1); launch(2); }
        }
        HIPCHK(hipEventRecord(e1, c->compute));
        HIPCHK(hipEventSynchronize(e1));
        HIPCHK(hipGetLastError());
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        out_us[m] = (double)ms * 1e3 / iters;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

#include "host_trust.inc"

#include "host_focops.inc"

#include "host_sac.inc"

#include "host_cvpo.inc"

#include "host_comm.inc"

#ifdef FSRL_PROBES
// probe builds only (not in include/fsrl_hip.h): the phase stamps of the last ppo_fwd_bwd_kernel launch
extern "C" int fsrl_probe_tstamps(fsrl_ctx* c, unsigned long long* out, int64_t n) {
    if (!c || !c->probe_ts) return FSRL_ESTATE;
    (void)hipStreamSynchronize(c->compute);
    return hipMemcpy(out, c->probe_ts, (size_t)std::min<int64_t>(n, 1024 * 16) * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : FSRL_EHIP;
}
#endif

// ------------------------------------------------------------------------------ return statistics
// BasePolicy.ret_rms (base_policy.py:111): one RunningMeanStd per critic, rows of (mean, var, count).  The reference keeps
// them as plain attributes (not in state_dict); a host that wants them across a restart reads / writes them here.
extern "C" int fsrl_ret_rms_get(fsrl_ctx* c, double* out, int32_t n) {
    CHECK_ARG(c && out, "null argument");
    CHECK_ARG(n == 3 * c->cfg.n_critics, "expected 3 * n_critics = %d doubles", 3 * c->cfg.n_critics);
    if (!c->d_rms) return fail(FSRL_ESTATE, "reward_normalization is off in this context");
    ENTER_DEV(c);
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipMemcpy(out, c->d_rms, (size_t)n * 8, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int fsrl_ret_rms_set(fsrl_ctx* c, const double* in, int32_t n) {
    CHECK_ARG(c && in, "null argument");
    CHECK_ARG(n == 3 * c->cfg.n_critics, "expected 3 * n_critics = %d doubles", 3 * c->cfg.n_critics);
    if (!c->d_rms) return fail(FSRL_ESTATE, "reward_normalization is off in this context");
    for (int i = 0; i < c->cfg.n_critics; ++i)
        CHECK_ARG(std::isfinite(in[3 * i]) && in[3 * i + 1] >= 0.0 && in[3 * i + 2] >= 0.0, "row %d: var and count must be >= 0", i);
    ENTER_DEV(c);
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipMemcpy(c->d_rms, in, (size_t)n * 8, hipMemcpyHostToDevice));
    return 0;
}

// ------------------------------------------------------------------------------ learning rates
// lr_scheduler.step() of BasePolicy.update (fsrl/policy/base_policy.py:352-354): the caller's scheduler owns the
// schedule, this call moves the result into the engine.  Takes effect from the next optimiser step.
extern "C" int fsrl_set_lr(fsrl_ctx* c, int32_t group, float lr) {
    CHECK_ARG(c, "null ctx");
    CHECK_ARG(lr >= 0.0f && std::isfinite(lr), "learning rate must be finite and >= 0");
    if (ctx_is_replay(c)) {
        SacState* s = sac_of(c);
        if (!s) return fail(FSRL_ESTATE, "fsrl_set_lr before fsrl_sac_init / fsrl_cvpo_init");
        CHECK_ARG(group >= 0 && group <= 2, "replay contexts: group 0 actor, 1 critics, 2 alpha");
        if (group == 0) { s->cfg.actor_lr = lr; if (s->cvpo) s->ccfg.actor_lr = lr; }
        else if (group == 1) { s->cfg.critic_lr = lr; if (s->cvpo) s->ccfg.critic_lr = lr; }
        else s->cfg.alpha_lr = lr;
        return 0;
    }
    if (c->cfg.algo == FSRL_ALGO_FOCOPS) {
        if (!c->foc) return fail(FSRL_ESTATE, "fsrl_set_lr before fsrl_focops_init");
        CHECK_ARG(group == 0 || group == 1, "FOCOPS: group 0 actor, 1 critics");
        if (group == 0) c->foc->cfg.actor_lr = lr; else c->foc->cfg.critic_lr = lr;
        return 0;
    }
    CHECK_ARG(group == 0, "on-policy contexts have one optimiser (group 0)");
    c->cfg.lr = lr;
    if (c->tr) c->tr->cfg.critic_lr = lr;      // CPO / TRPO-Lag: the optimiser steps the critics (fsrl_tr_begin re-reads it too)
    return 0;
}
extern "C" float fsrl_get_lr(const fsrl_ctx* c, int32_t group) {
    if (!c) return -1.0f;
    if (c->cfg.algo == FSRL_ALGO_SAC_LAG) {
        const SacState* s = reinterpret_cast<const SacState*>(c->sac);
        if (!s) return -1.0f;
        return group == 0 ? s->cfg.actor_lr : group == 1 ? s->cfg.critic_lr : group == 2 ? s->cfg.alpha_lr : -1.0f;
    }
    if (c->cfg.algo == FSRL_ALGO_FOCOPS) return !c->foc ? -1.0f : group == 0 ? c->foc->cfg.actor_lr : group == 1 ? c->foc->cfg.critic_lr : -1.0f;
    return group == 0 ? c->cfg.lr : -1.0f;
}
