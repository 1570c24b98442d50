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
#include <string>
#include <vector>

#include "kernels_misc.hpp"
#include "kernels_mlp.hpp"
#include "kernels_fb.hpp"

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

struct TensorMap { int api_off, dev_off, n; };   // one parameter tensor: flat API offset -> device

struct EnvBook {           // tianshou ReplayBuffer bookkeeping of one sub-buffer (host side)
    int64_t index = 0, size = 0, last_index = 0;
    int64_t staged = 0;    // rows of this sub-buffer in the current staging window
    double ep_rew = 0.0;
    int32_t ep_len = 0;
    int64_t ep_idx = 0;
};

struct Staging {           // pinned host staging + its device mirror (one of two)
    int* slot = nullptr; float* obs = nullptr; float* obs_next = nullptr; float* act = nullptr;
    double* rew = nullptr; double* cost = nullptr; uint8_t* flags = nullptr;
    int* d_slot = nullptr; float* d_obs = nullptr; float* d_obs_next = nullptr; float* d_act = nullptr;
    double* d_rew = nullptr; double* d_cost = nullptr; uint8_t* d_flags = nullptr;
    int count = 0;
    hipEvent_t done = nullptr;
    bool in_flight = false;
};

struct fsrl_ctx {
    fsrl_config cfg{};
    int device = 0;
    hipStream_t compute = nullptr, side = nullptr;
    ModelDesc md{};
    std::vector<TensorMap> tmap;
    int64_t n_api = 0;     // flat parameter count (API)
    int n_dev = 0;         // padded device parameter count (main vector: what Adam / clip / copies see)
    int n_alloc = 0;       // n_dev + the forward-fragment mirrors of every W2 (P only)
    float *P = nullptr, *M = nullptr, *V = nullptr, *G = nullptr;
    int64_t adam_t = 0;
    CtrlBlock* ctrl = nullptr;
    CtrlBlock* h_ctrl = nullptr;  // pinned

    // store
    int64_t sub_size = 0, maxsize = 0;
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
    uint64_t rng[4] = {0x9E3779B97F4A7C15ull, 0xBF58476D1CE4E5B9ull, 0x94D049BB133111EBull, 1};
    // split-K partial gradients of fb_wgrad_kernel, one buffer per parameter layout (keyed by its
    // padded size) so that the never-written inter-tensor padding stays zero
    struct Parts { int stride = 0; float* p = nullptr; size_t floats = 0; } parts[3];
    float* wg_parts = nullptr;      // the buffer the last wgrad_launch wrote
    int n_cus = 256;                // compute units of the device (tile-shape heuristic)
    struct FocState* foc = nullptr; // FOCOPS working set, owned
    float* mu_old = nullptr;        // [maxsize][Da] actor means at process time (FOCOPS)
    float* sigma_old = nullptr;     // [FSRL_MAX_ACT] sigma_param at process time (FOCOPS)
    void* h_actor = nullptr; size_t h_actor_bytes = 0;   // pinned staging of fsrl_actor_forward
    int actor_k = 0; size_t actor_ob = 0, actor_mb = 0;  // geometry of the actor evaluation in flight
    unsigned* h_done = nullptr; int done_cap = 0;        // pinned per-block completion words of that evaluation
    unsigned actor_seq = 0; int actor_blocks = 0;
    bool no_spin = getenv("FSRL_NO_SPIN") != nullptr;   // A/B switch: wait with hipStreamSynchronize instead
    std::vector<float> act_mu, act_sg;                   // mean / std of the last actor evaluation (host)
    std::vector<int> perm_tmp;      // this pass's permutation before it goes to the pinned buffer
    hipEvent_t perm_copied = nullptr; bool perm_in_flight = false;
    uint64_t store_version = 1;     // bumped by every push / reset (device copies of the bookkeeping)
    uint64_t joined_version = 0;    // store_version at the last side -> compute stream join (join_store)
    struct TrState* tr = nullptr;   // trust-region (CPO / TRPO-Lag) working set, owned
    void* sac = nullptr;            // SacState, owned
};
static void sac_free(fsrl_ctx* c);
static void tr_free(fsrl_ctx* c);
static void foc_free(fsrl_ctx* c);
static int focops_pass(fsrl_ctx* c, int32_t* stopped_out);

// Make the rows pushed so far visible to the compute stream: flush the staging window and order the side stream's
// copies before whatever the compute stream runs next.  Off-policy trainers call update() many times between collects;
// the join (an event record + wait, ~6 us of GPU idle per update) is skipped while the store has not changed.
static int flush_stage(fsrl_ctx* c);
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
static WgradPlan wgrad_plan(int rows) {
    const int KS = rows >> 2;
    int n = std::max(1, std::min((rows + 255) / 256, 24));
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
template <bool PAIR2>
static int wgrad_launch(fsrl_ctx* c, const ModelDesc& md, FbWgradArgs& wa, int ny, int stride, int* nsplit) {
    const WgradPlan pl = wgrad_plan(wa.rows);
    int rc = ensure_parts(c, stride, pl.nsplit);
    if (rc) return rc;
    wa.out = c->wg_parts; wa.ks_per_split = pl.ks_per_split; wa.split_stride = stride;
    { const char* e = getenv("FSRL_WGRAD_SKIP"); wa.dbg_skip = e ? atoi(e) : 0; }
    *nsplit = pl.nsplit;
    return dispatch_H(c->cfg.hidden, [&](auto hc) {
        constexpr int HH = decltype(hc)::value;
        constexpr int NB = (HH / 64) * (HH / 64) + HH / 32 + 1;   // 64x64 dW2 tiles + 32-column aux blocks + db3 block
        hipLaunchKernelGGL((fb_wgrad_kernel<HH, PAIR2>), dim3(NB, ny, pl.nsplit), dim3(1024), 0, c->compute, md, wa);
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
    int api = 0, dev = 0;
    auto add = [&](int n) {
        TensorMap t{api, dev, n};
        c->tmap.push_back(t);
        api += n;
        dev = round_up(dev + n, 64);  // every tensor starts 256-byte aligned on the device
        return t.dev_off;
    };
    for (int net = 0; net < md.n_nets; ++net) {
        NetOff& no = md.net[net];
        no.begin = dev;
        const int out = (net == 0) ? Da : 1;
        no.out = out;
        no.sigma = (net == 0) ? add(Da) : -1;
        no.W1 = add(H * Do); no.b1 = add(H);
        no.W2 = add(H * H);  no.b2 = add(H);
        no.W3 = add(out * H); no.b3 = add(out);
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
    (void)hipDeviceSynchronize();
    tr_free(c);
    sac_free(c);
    foc_free(c);
    if (c->h_actor) (void)hipHostFree(c->h_actor);
    if (c->h_done) (void)hipHostFree(c->h_done);
    if (c->mu_old) (void)hipFree(c->mu_old);
    if (c->sigma_old) (void)hipFree(c->sigma_old);
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
        void* hp[] = {s.slot, s.obs, s.obs_next, s.act, s.rew, s.cost, s.flags};
        for (void* p : hp) if (p) (void)hipHostFree(p);
        void* dp[] = {s.d_slot, s.d_obs, s.d_obs_next, s.d_act, s.d_rew, s.d_cost, s.d_flags};
        for (void* p : dp) if (p) (void)hipFree(p);
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
    CHECK_ARG(cfg->hidden == 64 || cfg->hidden == 128 || cfg->hidden == 256,
              "hidden must be 64, 128 or 256 (two equal hidden layers)");
    CHECK_ARG(cfg->n_critics >= 1 && cfg->n_critics <= 2,
              "n_critics must be 1 or 2 (reward [+ one cost], get_metrics base_policy.py:377-382)");
    CHECK_ARG(cfg->env_num >= 1 && cfg->buffer_size >= cfg->env_num, "bad env_num/buffer_size");
    CHECK_ARG(cfg->gamma >= 0.0 && cfg->gamma <= 1.0, "discount factor should be in [0, 1].");
    CHECK_ARG(cfg->gae_lambda >= 0.0 && cfg->gae_lambda <= 1.0, "GAE lambda should be in [0, 1].");
    CHECK_ARG(cfg->dual_clip == 0.0f || cfg->dual_clip > 1.0f,
              "Dual-clip PPO parameter should greater than 1.0.");
    CHECK_ARG(cfg->buffer_size + cfg->env_num < (int64_t)INT_MAX / 2, "buffer too large for 32-bit slot ids");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    CHECK_ARG(device_id >= 0 && device_id < ndev, "device %d not present (%d devices)", device_id, ndev);
    HIPCHK(hipSetDevice(device_id));
    int n_cus_probe = 256;
    (void)hipDeviceGetAttribute(&n_cus_probe, hipDeviceAttributeMultiprocessorCount, device_id);
    fsrl_ctx* c = new fsrl_ctx();
    c->cfg = *cfg;
    c->device = device_id;
    c->n_cus = n_cus_probe > 0 ? n_cus_probe : 256;
    build_layout(c);
#define TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { fail(FSRL_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); fsrl_ctx_destroy(c); return FSRL_EHIP; } } while (0)
    TRY(hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking));
    TRY(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    const size_t pb = (size_t)c->n_dev * sizeof(float);
    TRY(hipMalloc(&c->P, (size_t)c->n_alloc * sizeof(float))); TRY(hipMemsetAsync(c->P, 0, (size_t)c->n_alloc * sizeof(float), c->compute));
    TRY(hipMalloc(&c->M, pb)); TRY(hipMalloc(&c->V, pb)); TRY(hipMalloc(&c->G, pb));
    TRY(hipMemsetAsync(c->M, 0, pb, c->compute)); TRY(hipMemsetAsync(c->V, 0, pb, c->compute)); TRY(hipMemsetAsync(c->G, 0, pb, c->compute));
    TRY(hipMalloc(&c->ctrl, sizeof(CtrlBlock)));
    TRY(hipHostMalloc(&c->h_ctrl, sizeof(CtrlBlock)));
    // store: n sub-buffers of ceil(total/n) rows (tianshou VectorReplayBuffer)
    c->sub_size = (cfg->buffer_size + cfg->env_num - 1) / cfg->env_num;
    c->maxsize = c->sub_size * cfg->env_num;
    c->env.resize(cfg->env_num);
    c->h_flags.assign((size_t)c->maxsize, 0);
    const size_t ms = (size_t)c->maxsize;
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
        TRY(hipHostMalloc(&s.slot, k * 4)); TRY(hipHostMalloc(&s.obs, k * Do * 4));
        TRY(hipHostMalloc(&s.obs_next, k * Do * 4)); TRY(hipHostMalloc(&s.act, k * Da * 4));
        TRY(hipHostMalloc(&s.rew, k * 8)); TRY(hipHostMalloc(&s.cost, k * 8)); TRY(hipHostMalloc(&s.flags, k));
        TRY(hipMalloc(&s.d_slot, k * 4)); TRY(hipMalloc(&s.d_obs, k * Do * 4));
        TRY(hipMalloc(&s.d_obs_next, k * Do * 4)); TRY(hipMalloc(&s.d_act, k * Da * 4));
        TRY(hipMalloc(&s.d_rew, k * 8)); TRY(hipMalloc(&s.d_cost, k * 8)); TRY(hipMalloc(&s.d_flags, k));
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
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->side));
    HIPCHK(hipStreamSynchronize(c->compute));
    return 0;
}

// ------------------------------------------------------------------------------ parameters
extern "C" int64_t fsrl_param_count(const fsrl_ctx* c) { return c ? c->n_api : 0; }

static int copy_flat(fsrl_ctx* c, float* dev, float* host_out, const float* host_in, int64_t n) {
    CHECK_ARG(n == c->n_api, "expected %lld parameters, got %lld", (long long)c->n_api, (long long)n);
    HIPCHK(hipSetDevice(c->device));
    std::vector<float> tmp((size_t)c->n_dev, 0.0f);
    if (host_in) {                              // only P is ever written from the host
        for (const TensorMap& t : c->tmap) memcpy(&tmp[t.dev_off], host_in + t.api_off, (size_t)t.n * 4);
        HIPCHK(hipStreamSynchronize(c->compute));
        HIPCHK(hipMemcpy(dev, tmp.data(), (size_t)c->n_dev * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(w2f_sync_kernel, dim3(192), dim3(256), 0, c->compute, dev, c->md);   // W2 mirrors
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipStreamSynchronize(c->compute));
        HIPCHK(hipMemcpy(tmp.data(), dev, (size_t)c->n_dev * 4, hipMemcpyDeviceToHost));
        for (const TensorMap& t : c->tmap) memcpy(host_out + t.api_off, &tmp[t.dev_off], (size_t)t.n * 4);
    }
    return 0;
}

extern "C" int fsrl_params_set(fsrl_ctx* c, const float* flat, int64_t n) {
    CHECK_ARG(c && flat, "null argument");
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
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipMemsetAsync(c->M, 0, (size_t)c->n_dev * 4, c->compute));
    HIPCHK(hipMemsetAsync(c->V, 0, (size_t)c->n_dev * 4, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    c->adam_t = 0;
    return 0;
}

// ------------------------------------------------------------------------------ store
static int flush_stage(fsrl_ctx* c) {
    Staging& s = c->stage[c->cur_stage];
    if (s.count == 0) return 0;
    const int k = s.count, Do = c->cfg.obs_dim, Da = c->cfg.act_dim;
    HIPCHK(hipMemcpyAsync(s.d_slot, s.slot, (size_t)k * 4, hipMemcpyHostToDevice, c->side));
    HIPCHK(hipMemcpyAsync(s.d_obs, s.obs, (size_t)k * Do * 4, hipMemcpyHostToDevice, c->side));
    HIPCHK(hipMemcpyAsync(s.d_obs_next, s.obs_next, (size_t)k * Do * 4, hipMemcpyHostToDevice, c->side));
    HIPCHK(hipMemcpyAsync(s.d_act, s.act, (size_t)k * Da * 4, hipMemcpyHostToDevice, c->side));
    HIPCHK(hipMemcpyAsync(s.d_rew, s.rew, (size_t)k * 8, hipMemcpyHostToDevice, c->side));
    HIPCHK(hipMemcpyAsync(s.d_cost, s.cost, (size_t)k * 8, hipMemcpyHostToDevice, c->side));
    HIPCHK(hipMemcpyAsync(s.d_flags, s.flags, (size_t)k, hipMemcpyHostToDevice, c->side));
    const int per = 2 * Do + Da + 1;
    const int blocks = std::min(1024, (k * per + 255) / 256);
    hipLaunchKernelGGL(store_scatter_kernel, dim3(blocks), dim3(256), 0, c->side, c->st, s.d_slot, s.d_obs,
                       s.d_obs_next, s.d_act, s.d_rew, s.d_cost, s.d_flags, k, Do, Da);
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
    CHECK_ARG(k >= 0 && k <= c->cfg.env_num, "k=%d rows but %d sub-buffers", k, c->cfg.env_num);
    HIPCHK(hipSetDevice(c->device));
    c->store_version += 1;
    const int Do = c->cfg.obs_dim, Da = c->cfg.act_dim;
    for (int j = 0; j < k; ++j) {
        const int e = env_ids[j];
        CHECK_ARG(e >= 0 && e < c->cfg.env_num, "buffer id %d out of range", e);
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
        s.slot[i] = (int)gptr;
        memcpy(s.obs + (size_t)i * Do, obs + (size_t)j * Do, (size_t)Do * 4);
        memcpy(s.obs_next + (size_t)i * Do, obs_next + (size_t)j * Do, (size_t)Do * 4);
        memcpy(s.act + (size_t)i * Da, act + (size_t)j * Da, (size_t)Da * 4);
        s.rew[i] = rew[j];
        s.cost[i] = cost ? cost[j] : 0.0;
        const uint8_t fl = (uint8_t)((terminated[j] ? 1 : 0) | (truncated[j] ? 2 : 0));
        s.flags[i] = fl;
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
    (void)keep_statistics;
    HIPCHK(hipSetDevice(c->device));
    int rc = flush_stage(c);   // rows already staged still land (then become unreachable)
    if (rc) return rc;
    for (EnvBook& e : c->env) e = EnvBook();
    c->batch_ready = false;
    c->store_version += 1;
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
    return dispatch_H(c->cfg.hidden, [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        hipLaunchKernelGGL(mlp_infer_kernel<H>, dim3(tiles, jobs_y), dim3(4 * H), 0, s, c->P, c->md, ia);
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

static int actor_eval_launch(fsrl_ctx* c, const float* obs, int32_t k, bool want_sigma) {
    const int Do = c->cfg.obs_dim, Da = c->cfg.act_dim;
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
    HIPCHK(hipSetDevice(c->device));
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
    HIPCHK(hipSetDevice(c->device));
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
    HIPCHK(hipSetDevice(c->device));
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
    HIPCHK(hipSetDevice(c->device));
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

// ------------------------------------------------------------------------------ PPO-Lagrangian
static void split_plan(int n, int B, std::vector<int>& st, std::vector<int>& sz) {
    // tianshou Batch.split(size, merge_last=True): the remainder is merged into the last chunk
    st.clear(); sz.clear();
    const bool merge = (n % B) > 0;
    for (int i = 0; i < n; i += B) {
        if (merge && i + 2 * B >= n) { st.push_back(i); sz.push_back(n - i); break; }
        st.push_back(i); sz.push_back(std::min(B, n - i));
    }
}

static int ensure_ppo_buffers(fsrl_ctx* c, int B) {
    const int mbp = round_up(2 * B, 16);
    if (mbp <= c->mbp_max) return 0;
    HIPCHK(hipStreamSynchronize(c->compute));
    for (float** p : {&c->A1, &c->A2, &c->D1, &c->D2, &c->DO, &c->statp, &c->gsq_part})
        if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; }
    const int H = c->cfg.hidden, nn = c->md.n_nets;
    const size_t act = (size_t)nn * mbp * H * 4;
    HIPCHK(hipMalloc(&c->A1, act)); HIPCHK(hipMalloc(&c->A2, act));
    HIPCHK(hipMalloc(&c->D1, act)); HIPCHK(hipMalloc(&c->D2, act));
    HIPCHK(hipMalloc(&c->DO, (size_t)nn * mbp * FSRL_DOW * 4));
    HIPCHK(hipMalloc(&c->statp, (size_t)(mbp / 4) * nn * 4 * 4));      // one slot per 4-row tile
    const int pb = (H / 32) * (H / 32) + H / 32;
    HIPCHK(hipMalloc(&c->gsq_part, (size_t)(nn * pb + 1) * 4));
    c->mbp_max = mbp;
    c->n_tiles_max = mbp / 16;
    return 0;
}

extern "C" int fsrl_ppo_begin(fsrl_ctx* c, const double* lagrangians, double rescaling,
                              int32_t batch_size, int64_t* n_out) {
    CHECK_ARG(c, "null ctx");
    CHECK_ARG(batch_size >= 1, "batch_size must be >= 1");
    CHECK_ARG(!c->in_update, "fsrl_ppo_begin called twice without fsrl_ppo_end");
    HIPCHK(hipSetDevice(c->device));
    int rc = flush_stage(c);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->store_ready, c->side));
    HIPCHK(hipStreamWaitEvent(c->compute, c->store_ready, 0));   // join side stream -> compute
    const int C = c->cfg.n_critics;
    for (int i = 0; i < FSRL_MAX_CRITICS; ++i) c->lagr[i] = 0.0;
    if (c->cfg.use_lagrangian && C > 1) {
        CHECK_ARG(lagrangians, "lags and values length must be equal");
        for (int i = 0; i < C - 1; ++i) c->lagr[i] = lagrangians[i];
    }
    c->rescaling = rescaling;
    c->batch_size = batch_size;
    // ---- batch, indices = buffer.sample(0)
    const int64_t n = sample0(c, c->h_indices, c->h_end);
    c->N = n;
    if (n_out) *n_out = n;
    c->n_steps = 0;
    c->pass_index = 0;
    c->in_update = true;
    c->t_fwdbwd_ms = 0; c->n_fwdbwd = 0; c->k_ev_used = 0;
    CtrlBlock init{INT_MAX, 0, 0.0, 0.0f, 0.0f};
    *c->h_ctrl = init;
    hipStream_t s = c->compute;
    HIPCHK(hipMemcpyAsync(c->ctrl, c->h_ctrl, sizeof(CtrlBlock), hipMemcpyHostToDevice, s));
    if (n == 0) { c->batch_ready = true; return 0; }
    // episode segments for the GAE scan
    int nseg = 0;
    c->h_seg[nseg++] = 0;
    for (int64_t i = 0; i + 1 < n; ++i) if (c->h_end[i]) c->h_seg[nseg++] = (int)i + 1;
    c->h_seg[nseg] = (int)n;
    HIPCHK(hipEventRecord(c->ev_a, s));
    HIPCHK(hipMemcpyAsync(c->d_indices, c->h_indices, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_end, c->h_end, (size_t)n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_seg, c->h_seg, (size_t)(nseg + 1) * 4, hipMemcpyHostToDevice, s));
    const int Do = c->cfg.obs_dim, Da = c->cfg.act_dim;
    {
        const size_t work = (size_t)n * (2 * Do + Da + 1);
        const int blocks = (int)std::min<size_t>(2048, (work + 255) / 256);
        hipLaunchKernelGGL(batch_gather_kernel, dim3(blocks), dim3(256), 0, s, c->st, c->b, c->d_indices,
                           c->d_end, (int)n, Do, Da);
        HIPCHK(hipGetLastError());
    }
    // ---- process_fn: V_i(obs), V_i(obs_next)*~terminated, logp_old, then float64 GAE per critic
    InferArgs ia{};
    ia.obs = c->b.obs; ia.obs_next = c->b.obs_next; ia.act = c->b.act; ia.flags = c->b.flags;
    ia.values = c->values; ia.vnext = c->vnext; ia.logp_old = c->logp_old; ia.mu_out = nullptr;
    if (c->cfg.algo == FSRL_ALGO_FOCOPS) {      // old distribution of the pass batches: means + sigma_param snapshot
        if (!c->mu_old) {
            HIPCHK(hipMalloc(&c->mu_old, (size_t)c->maxsize * Da * 4));
            HIPCHK(hipMalloc(&c->sigma_old, FSRL_MAX_ACT * 4));
        }
        ia.mu_out = c->mu_old;
        HIPCHK(hipMemcpyAsync(c->sigma_old, c->P + c->md.net[0].sigma, (size_t)Da * 4, hipMemcpyDeviceToDevice, s));
    }
    ia.N = (int)n; ia.C = C; ia.max_action = c->cfg.max_action;
    rc = launch_infer(c, ia, 2 * C + 1, s);
    if (rc) return rc;
    GaeArgs ga{};
    ga.values = c->values; ga.vnext = c->vnext; ga.rew = c->b.rew; ga.cost = c->b.cost; ga.flags = c->b.flags;
    ga.seg_start = c->d_seg; ga.advs = c->advs; ga.rets = c->rets; ga.adv64 = nullptr; ga.N = (int)n;
    ga.gamma = c->cfg.gamma; ga.gl = c->cfg.gamma * c->cfg.gae_lambda;
    hipLaunchKernelGGL(gae_kernel, dim3(nseg, C), dim3(64), 0, s, ga);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev_b, s));
    // ---- minibatch plan of every pass (same for all passes)
    split_plan((int)n, batch_size, c->mb_start, c->mb_size);
    rc = ensure_ppo_buffers(c, batch_size);
    if (rc) return rc;
    const size_t nmb = c->mb_start.size();
    if (nmb > c->mb_cap) {
        HIPCHK(hipStreamSynchronize(s));
        if (c->d_mbstart) HIPCHK(hipFree(c->d_mbstart));
        if (c->d_mbsize) HIPCHK(hipFree(c->d_mbsize));
        if (c->h_mbplan) HIPCHK(hipHostFree(c->h_mbplan));
        c->d_mbstart = c->d_mbsize = nullptr; c->h_mbplan = nullptr;
        const size_t cap = nmb * 2;
        HIPCHK(hipMalloc(&c->d_mbstart, cap * 4)); HIPCHK(hipMalloc(&c->d_mbsize, cap * 4));
        HIPCHK(hipHostMalloc(&c->h_mbplan, cap * 2 * 4));
        c->mb_cap = cap;
    }
    memcpy(c->h_mbplan, c->mb_start.data(), nmb * 4);
    memcpy(c->h_mbplan + c->mb_cap, c->mb_size.data(), nmb * 4);
    HIPCHK(hipMemcpyAsync(c->d_mbstart, c->h_mbplan, nmb * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_mbsize, c->h_mbplan + c->mb_cap, nmb * 4, hipMemcpyHostToDevice, s));
    c->batch_ready = true;
    return 0;
}


static int ensure_stats(fsrl_ctx* c, int64_t steps) {
    if (steps <= c->stats_cap) return 0;
    int64_t cap = std::max<int64_t>(steps, c->stats_cap * 2);
    cap = std::max<int64_t>(cap, 1024);
    float* nw = nullptr;
    HIPCHK(hipMalloc(&nw, (size_t)cap * FSRL_PPO_NSTATS * 4));
    if (c->d_stats) {
        HIPCHK(hipStreamSynchronize(c->compute));
        HIPCHK(hipMemcpy(nw, c->d_stats, (size_t)c->stats_cap * FSRL_PPO_NSTATS * 4, hipMemcpyDeviceToDevice));
        HIPCHK(hipFree(c->d_stats));
    }
    c->d_stats = nw;
    c->stats_cap = cap;
    return 0;
}

extern "C" int fsrl_ppo_pass(fsrl_ctx* c, const int64_t* perm, uint64_t seed, int32_t* stopped_out) {
    CHECK_ARG(c, "null ctx");
    if (!c->in_update || !c->batch_ready) return fail(FSRL_ESTATE, "fsrl_ppo_pass before fsrl_ppo_begin");
    HIPCHK(hipSetDevice(c->device));
    if (stopped_out) *stopped_out = 0;
    const int n = (int)c->N;
    if (n == 0) return 0;
    hipStream_t s = c->compute;
    const int C = c->cfg.n_critics, H = c->cfg.hidden, nn = c->md.n_nets;
    // ---- permutation of this pass (np.random.permutation on the caller side, or our own), built in
    //      pageable memory first: the GPU keeps working on the previous pass meanwhile.  Only the previous
    //      H2D copy out of the pinned buffer has to be over (an event, not a stream drain).
    c->perm_tmp.resize((size_t)n);
    if (perm) {
        std::vector<uint8_t> seen((size_t)n, 0);
        for (int i = 0; i < n; ++i) {
            CHECK_ARG(perm[i] >= 0 && perm[i] < n, "perm[%d]=%lld out of range", i, (long long)perm[i]);
            CHECK_ARG(!seen[(size_t)perm[i]], "perm is not a permutation: %lld appears twice", (long long)perm[i]);
            seen[(size_t)perm[i]] = 1;
            c->perm_tmp[(size_t)i] = (int)perm[i];
        }
    } else {
        if (seed) { c->rng[0] ^= seed; c->rng[1] += seed * 0x9E3779B97F4A7C15ull; }
        for (int i = 0; i < n; ++i) c->perm_tmp[(size_t)i] = i;
        for (int i = n - 1; i > 0; --i) {
            const int j = (int)(xoshiro_next(c->rng) % (uint64_t)(i + 1));
            std::swap(c->perm_tmp[(size_t)i], c->perm_tmp[(size_t)j]);
        }
    }
    if (c->perm_in_flight) { HIPCHK(hipEventSynchronize(c->perm_copied)); c->perm_in_flight = false; }
    memcpy(c->h_perm, c->perm_tmp.data(), (size_t)n * 4);
    const int nmb = (int)c->mb_start.size();
    int rc = ensure_stats(c, c->n_steps + nmb);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->d_perm, c->h_perm, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(c->perm_copied, s));
    c->perm_in_flight = true;
    {
        PrepArgs pa{};
        pa.obs = c->b.obs; pa.act = c->b.act; pa.advs = c->advs; pa.rets = c->rets; pa.logp_old = c->logp_old;
        pa.perm = c->d_perm; pa.mb_start = c->d_mbstart; pa.mb_size = c->d_mbsize; pa.obs_p = c->obs_p;
        pa.rd_p = c->rd_p; pa.N = n; pa.C = C; pa.Do = c->cfg.obs_dim; pa.Da = c->cfg.act_dim;
        pa.norm_adv = c->cfg.norm_adv;
        pa.mean_old = (c->cfg.algo == FSRL_ALGO_FOCOPS) ? c->mu_old : nullptr; pa.sigma_old = c->sigma_old;
        hipLaunchKernelGGL(ppo_prepare_pass_kernel, dim3(nmb), dim3(1024), 0, s, pa);
        HIPCHK(hipGetLastError());
    }
    if (c->cfg.algo == FSRL_ALGO_FOCOPS) return focops_pass(c, stopped_out);
    PpoBatchPtrs bp{};
    bp.obs_p = c->obs_p; bp.rd_p = c->rd_p; bp.A1 = c->A1; bp.A2 = c->A2; bp.D1 = c->D1; bp.D2 = c->D2;
    bp.DO = c->DO; bp.statp = c->statp; bp.mbp_max = c->mbp_max;
    WgradPtrs wp{};
    wp.A1 = c->A1; wp.A2 = c->A2; wp.D1 = c->D1; wp.D2 = c->D2; wp.DO = c->DO; wp.X = c->obs_p; wp.grad = c->G;
    wp.gsq_part = c->gsq_part; wp.ctrl = c->ctrl; wp.mbp_max = c->mbp_max;
    wp.P = c->P; wp.statp = c->statp; wp.stats = c->d_stats;
    const int pb = (H / 32) * (H / 32) + H / 32;
    const int nparts = nn * pb + 1;   // + the stats block

    PpoStepArgs sa{};
    sa.rescale = (float)c->rescaling;
    for (int i = 0; i < FSRL_MAX_CRITICS; ++i) sa.lam[i] = (float)c->lagr[i];
    sa.eps_clip = c->cfg.eps_clip; sa.dual_clip = c->cfg.dual_clip; sa.vf_coef = c->cfg.vf_coef;
    sa.max_action = c->cfg.max_action; sa.max_grad_norm = c->cfg.max_grad_norm; sa.target_kl = c->cfg.target_kl;
    sa.norm_adv = c->cfg.norm_adv; sa.use_lagrangian = c->cfg.use_lagrangian;
    sa.lr = c->cfg.lr; sa.beta1 = c->cfg.beta1; sa.beta2 = c->cfg.beta2; sa.adam_eps = c->cfg.adam_eps;
    sa.one_minus_b1 = (float)(1.0 - (double)c->cfg.beta1);
    sa.one_minus_b2 = (float)(1.0 - (double)c->cfg.beta2);
    sa.kl_thresh = 1.5 * (double)c->cfg.target_kl;
    sa.pass = c->pass_index;
    { const char* e = getenv("FSRL_DBG_PHASE"); sa.dbg_phase = e ? atoi(e) : 0; }
    sa.iters_in_pass = nmb;

    for (int mb = 0; mb < nmb; ++mb) {
        sa.mb_start = c->mb_start[mb]; sa.mb_size = c->mb_size[mb]; sa.mb_index = mb;
        sa.step = (int)c->n_steps + mb;
        sa.first_in_pass = (mb == 0); sa.last_in_pass = (mb == nmb - 1);
        c->adam_t += 1;
        const double bc1 = 1.0 - std::pow((double)c->cfg.beta1, (double)c->adam_t);
        const double bc2 = 1.0 - std::pow((double)c->cfg.beta2, (double)c->adam_t);
        sa.step_size = (float)((double)c->cfg.lr / bc1);
        sa.bc2_sqrt = (float)std::sqrt(bc2);
        const int tiles = (sa.mb_size + 15) / 16;
        wp.X = c->obs_p + (size_t)sa.mb_start * c->cfg.obs_dim;
        const bool prof = c->profiling;
        if (prof) {
            while (c->k_ev.size() < c->k_ev_used + 3) {
                hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->k_ev.push_back(e);
            }
            HIPCHK(hipEventRecord(c->k_ev[c->k_ev_used], s));
        }
        // 4-row tiles (4x4x1 MFMA) when they still fit the chip in one round: four times the CUs,
        // a quarter of the MFMA time each; 16-row tiles otherwise
        const bool rows4 = tiles * 4 * nn <= c->n_cus && !getenv("FSRL_TILE16");
        const int stat_tiles = rows4 ? tiles * 4 : tiles;
        rc = dispatch_H(H, [&](auto hc) {
            constexpr int HH = decltype(hc)::value;
            if (rows4) hipLaunchKernelGGL((ppo_fwd_bwd_kernel<HH, 4>), dim3(tiles * 4 * nn), dim3(4 * HH), 0, s, c->P, c->md, bp, sa);
            else hipLaunchKernelGGL((ppo_fwd_bwd_kernel<HH, 16>), dim3(tiles * nn), dim3(4 * HH), 0, s, c->P, c->md, bp, sa);
            return 0;
        });
        if (rc) return rc;
        if (prof) {
            // e0 | kernel | e1 | e2 : (e2 - e1) is the cost of an empty event bracket on this
            // stream.  Half of it overlaps the kernel's own dispatch, so (e1 - e0) - (e2 - e1)/2
            // is reported (calibrated against rocprofv3: 22.4 vs 22.1 us); raw sums are kept too
            HIPCHK(hipEventRecord(c->k_ev[c->k_ev_used + 1], s));
            HIPCHK(hipEventRecord(c->k_ev[c->k_ev_used + 2], s));
            c->k_ev_used += 3;
        }
        rc = dispatch_H(H, [&](auto hc) {
            constexpr int HH = decltype(hc)::value;
            if (tiles * 16 <= 512) hipLaunchKernelGGL((ppo_wgrad_kernel<HH, false>), dim3(nparts), dim3(1024), 0, s, c->md, wp, tiles * 16, sa, stat_tiles);
            else hipLaunchKernelGGL((ppo_wgrad_kernel<HH, true>), dim3(nparts), dim3(1024), 0, s, c->md, wp, tiles * 16, sa, stat_tiles);
            return 0;
        });
        if (rc) return rc;
        hipLaunchKernelGGL(adam_clip_kernel, dim3((c->n_dev + 4 * ADAM_NT - 1) / (4 * ADAM_NT)), dim3(ADAM_NT), 0, s, c->P, c->M, c->V, c->G,
                           c->gsq_part, nparts, c->n_dev, sa, c->ctrl, c->md);
        HIPCHK(hipGetLastError());
    }
    c->n_steps += nmb;
    c->pass_index += 1;
    // ---- pass-level KL early stop: one small readback per pass (the reference's `break`)
    if (c->cfg.target_kl > 0.0f) {
        HIPCHK(hipMemcpyAsync(c->h_ctrl, c->ctrl, sizeof(CtrlBlock), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (c->h_ctrl->stopped_after != INT_MAX) {
            if (stopped_out) *stopped_out = 1;
        }
    }
    return 0;
}

extern "C" int fsrl_ppo_end(fsrl_ctx* c, float* stats_out, int64_t cap_steps, int64_t* n_steps_out) {
    CHECK_ARG(c, "null ctx");
    if (!c->in_update) return fail(FSRL_ESTATE, "fsrl_ppo_end without fsrl_ppo_begin");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->compute;
    HIPCHK(hipEventRecord(c->ev_c, s));
    if (stats_out && c->n_steps > 0) {
        CHECK_ARG(cap_steps >= c->n_steps, "stats_out holds %lld steps, need %lld", (long long)cap_steps,
                  (long long)c->n_steps);
        HIPCHK(hipMemcpyAsync(stats_out, c->d_stats, (size_t)c->n_steps * FSRL_PPO_NSTATS * 4,
                              hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    if (n_steps_out) *n_steps_out = c->n_steps;
    c->in_update = false;
    if (c->N > 0) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_a, c->ev_b) == hipSuccess) c->t_process_ms = ms;
        if (hipEventElapsedTime(&ms, c->ev_b, c->ev_c) == hipSuccess) c->t_learn_ms = ms;
        double tot = 0;
        c->t_fwdbwd_raw_ms = 0;
        for (size_t i = 0; i + 2 < c->k_ev_used; i += 3) {
            float a = 0, b = 0;
            if (hipEventElapsedTime(&a, c->k_ev[i], c->k_ev[i + 1]) == hipSuccess &&
                hipEventElapsedTime(&b, c->k_ev[i + 1], c->k_ev[i + 2]) == hipSuccess)
            { tot += (double)a - 0.5 * (double)b; c->t_fwdbwd_raw_ms += (double)a; }
        }
        c->t_fwdbwd_ms = tot;
        c->n_fwdbwd = (int64_t)(c->k_ev_used / 3);
    }
    return 0;
}

extern "C" int fsrl_ppo_update(fsrl_ctx* c, const double* lagrangians, double rescaling,
                               int32_t batch_size, int32_t repeat, const int64_t* perms, uint64_t seed,
                               float* stats_out, int64_t cap_steps, int64_t* n_steps_out,
                               int32_t* stopped_pass_out) {
    CHECK_ARG(c, "null ctx");
    CHECK_ARG(repeat >= 0, "repeat must be >= 0");
    int64_t n = 0;
    int rc = fsrl_ppo_begin(c, lagrangians, rescaling, batch_size, &n);
    if (rc) return rc;
    if (stopped_pass_out) *stopped_pass_out = -1;
    for (int k = 0; k < repeat; ++k) {
        int32_t stopped = 0;
        rc = fsrl_ppo_pass(c, perms ? perms + (size_t)k * n : nullptr, seed ? seed + k : 0, &stopped);
        if (rc) { c->in_update = false; return rc; }
        if (stopped) { if (stopped_pass_out) *stopped_pass_out = k; break; }
    }
    return fsrl_ppo_end(c, stats_out, cap_steps, n_steps_out);
}

extern "C" int fsrl_batch_get(fsrl_ctx* c, const char* which, float* out, int64_t cap) {
    CHECK_ARG(c && which && out, "null argument");
    if (!c->batch_ready) return fail(FSRL_ESTATE, "no batch: call fsrl_ppo_begin first");
    HIPCHK(hipSetDevice(c->device));
    const int64_t n = c->N;
    const int C = c->cfg.n_critics;
    const float* src = nullptr;
    int cols = C;
    if (!strcmp(which, "values")) src = c->values;
    else if (!strcmp(which, "rets")) src = c->rets;
    else if (!strcmp(which, "advs")) src = c->advs;
    else if (!strcmp(which, "vnext")) src = c->vnext;
    else if (!strcmp(which, "logp_old")) { src = c->logp_old; cols = 1; }
    else return fail(FSRL_EINVAL, "unknown batch field '%s'", which);
    CHECK_ARG(cap >= n * cols, "out too small");
    HIPCHK(hipStreamSynchronize(c->compute));
    std::vector<float> tmp((size_t)n * cols);
    if (n) HIPCHK(hipMemcpy(tmp.data(), src, (size_t)n * cols * 4, hipMemcpyDeviceToHost));
    // device layout is [C][N]; the reference stacks on the last axis: [N][C]
    for (int64_t r = 0; r < n; ++r)
        for (int k = 0; k < cols; ++k) out[r * cols + k] = tmp[(size_t)k * n + r];
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

// ====================================================================================== trust region
// CPO / TRPO-Lagrangian: full-batch primitives on the device, the (tiny) flat-vector algebra of
// conjugate gradients, the dual solve and the line search on the host in float32.
struct TrState {
    fsrl_tr_config cfg{};
    bool ready = false;
    int rows_pad = 0, n_tiles = 0;
    float *A1 = nullptr, *A2 = nullptr, *D1 = nullptr, *D2 = nullptr, *DO = nullptr;      // [nets][rows_pad]
    float *RA1 = nullptr, *RA2 = nullptr, *RD1 = nullptr, *RD2 = nullptr, *RDO = nullptr;  // [rows_pad]
    float *statp = nullptr, *mu_old = nullptr, *Vdev = nullptr, *Out = nullptr, *rd = nullptr;
    float *cg_r = nullptr, *cg_x = nullptr, *cg_g = nullptr;     // device-resident CG vectors (actor layout)
    CgScal* cg_sc = nullptr;
    float* cg_part = nullptr;                                    // [3 * CG_NB]: p.z partials | r.r partials (two halves)
    double* d_scal = nullptr;
    size_t cap_rows = 0;
    int64_t critic_t = 0;     // Adam step count of the critic optimiser
    int na = 0;               // flat actor parameter count (API order)
    std::vector<int32_t> ls_iters;   // line-search evaluations of every repeat of the last learn call
};
static TrState* tr_of(fsrl_ctx* c) {
    if (!c->tr) {
        c->tr = new TrState();
        for (int i = 0; i < 7; ++i) c->tr->na += c->tmap[i].n;   // the actor's 7 tensors
    }
    return c->tr;
}

static void tr_free(fsrl_ctx* c) {
    TrState* t = c->tr;
    if (!t) return;
    for (float* p : {t->A1, t->A2, t->D1, t->D2, t->DO, t->RA1, t->RA2, t->RD1, t->RD2, t->RDO, t->statp,
                     t->mu_old, t->Vdev, t->Out, t->rd, t->cg_r, t->cg_x, t->cg_g})
        if (p) (void)hipFree(p);
    if (t->cg_sc) (void)hipFree(t->cg_sc);
    if (t->cg_part) (void)hipFree(t->cg_part);
    if (t->d_scal) (void)hipFree(t->d_scal);
    delete t;
    c->tr = nullptr;
}

extern "C" int64_t fsrl_actor_param_count(const fsrl_ctx* c) {
    if (!c) return 0;
    int64_t n = 0;
    for (int i = 0; i < 7; ++i) n += c->tmap[i].n;
    return n;
}

static int tr_alloc(fsrl_ctx* c, TrState* t, int64_t n) {
    const int tiles = (int)((n + 15) / 16);
    t->n_tiles = tiles; t->rows_pad = tiles * 16;
    if ((size_t)t->rows_pad <= t->cap_rows) return 0;
    HIPCHK(hipStreamSynchronize(c->compute));
    for (float** p : {&t->A1, &t->A2, &t->D1, &t->D2, &t->DO, &t->RA1, &t->RA2, &t->RD1, &t->RD2, &t->RDO,
                      &t->statp, &t->mu_old, &t->rd})
        if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; }
    const size_t rows = (size_t)t->rows_pad + 64, H = c->cfg.hidden, nn = c->md.n_nets;
    HIPCHK(hipMalloc(&t->A1, nn * rows * H * 4)); HIPCHK(hipMalloc(&t->A2, nn * rows * H * 4));
    HIPCHK(hipMalloc(&t->D1, nn * rows * H * 4)); HIPCHK(hipMalloc(&t->D2, nn * rows * H * 4));
    HIPCHK(hipMalloc(&t->DO, nn * rows * FSRL_DOW * 4));
    HIPCHK(hipMalloc(&t->RA1, rows * H * 4)); HIPCHK(hipMalloc(&t->RA2, rows * H * 4));
    HIPCHK(hipMalloc(&t->RD1, rows * H * 4)); HIPCHK(hipMalloc(&t->RD2, rows * H * 4));
    HIPCHK(hipMalloc(&t->RDO, rows * FSRL_DOW * 4));
    HIPCHK(hipMalloc(&t->statp, (size_t)(tiles + 4) * nn * FB_NSTAT * 4));
    HIPCHK(hipMalloc(&t->mu_old, rows * c->cfg.act_dim * 4));
    HIPCHK(hipMalloc(&t->rd, rows * FSRL_RD * 4));
    if (!t->Vdev) {
        HIPCHK(hipMalloc(&t->Vdev, (size_t)c->n_alloc * 4)); HIPCHK(hipMalloc(&t->Out, (size_t)c->n_dev * 4));
        for (float** p : {&t->cg_r, &t->cg_x, &t->cg_g}) {
            HIPCHK(hipMalloc(p, (size_t)c->n_dev * 4));
            HIPCHK(hipMemsetAsync(*p, 0, (size_t)c->n_dev * 4, c->compute));
        }
        HIPCHK(hipMalloc(&t->cg_sc, sizeof(CgScal)));
        HIPCHK(hipMalloc(&t->cg_part, 3 * CG_NB * sizeof(float)));
        HIPCHK(hipMemsetAsync(t->Vdev, 0, (size_t)c->n_alloc * 4, c->compute)); HIPCHK(hipMemsetAsync(t->Out, 0, (size_t)c->n_dev * 4, c->compute));
        HIPCHK(hipStreamSynchronize(c->compute));
        HIPCHK(hipMalloc(&t->d_scal, 64 * sizeof(double)));
    }
    t->cap_rows = (size_t)t->rows_pad;
    return 0;
}

// flat ACTOR vector (API order) <-> device-layout vector
static int actor_to_dev(fsrl_ctx* c, const float* host, float* dev) {
    std::vector<float> tmp((size_t)c->md.net[0].end, 0.0f);
    for (int i = 0; i < 7; ++i) memcpy(&tmp[c->tmap[i].dev_off], host + c->tmap[i].api_off, (size_t)c->tmap[i].n * 4);
    const int H = c->md.H;
    const NetOff& no = c->md.net[0];
    std::vector<float> mir((size_t)H * H);      // dev is P or a tangent vector: both carry the W2 mirror
    for (int n = 0; n < H; ++n)
        for (int k = 0; k < H; ++k) mir[(size_t)w2f_index(H, n, k)] = tmp[(size_t)no.W2 + (size_t)n * H + k];
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipMemcpy(dev, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dev + no.W2f, mir.data(), mir.size() * 4, hipMemcpyHostToDevice));
    return 0;
}
static int actor_from_dev(fsrl_ctx* c, const float* dev, float* host) {
    std::vector<float> tmp((size_t)c->md.net[0].end);
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipMemcpy(tmp.data(), dev, tmp.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 7; ++i) memcpy(host + c->tmap[i].api_off, &tmp[c->tmap[i].dev_off], (size_t)c->tmap[i].n * 4);
    return 0;
}

static int tr_refresh_old(fsrl_ctx* c, TrState* t) {
    // mean_old / std_old := current policy (TRPO recomputes old_dist at theta, trpo_lag.py:189-190)
    InferArgs ia{};
    ia.obs = c->b.obs; ia.obs_next = c->b.obs; ia.act = c->b.act; ia.flags = c->b.flags; ia.values = nullptr;
    ia.vnext = nullptr; ia.logp_old = nullptr; ia.mu_out = t->mu_old; ia.N = (int)c->N; ia.C = 0;
    ia.max_action = c->cfg.max_action;
    int rc = launch_infer(c, ia, 1, c->compute);
    if (rc) return rc;
    FbRowArgs ra{};
    ra.act = c->b.act; ra.advs = c->advs; ra.rets = c->rets; ra.logp_old = c->logp_old; ra.mean_old = t->mu_old;
    ra.sigma = c->P + c->md.net[0].sigma; ra.rd = t->rd; ra.N = (int)c->N; ra.C = c->cfg.n_critics; ra.Da = c->cfg.act_dim;
    hipLaunchKernelGGL(fb_rowdata_kernel, dim3(512), dim3(256), 0, c->compute, ra);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int fsrl_tr_begin(fsrl_ctx* c, const fsrl_tr_config* cfg, int64_t* n_out) {
    CHECK_ARG(c && cfg, "null argument");
    CHECK_ARG(c->cfg.n_critics == 2, "CPO / TRPO-Lag need a reward and one cost critic");
    TrState* t = tr_of(c);
    t->cfg = *cfg;
    // reuse the PPO begin for sample(0) + V(obs), V(obs_next), GAE, logp_old (batch_size irrelevant)
    double zero = 0.0;
    int64_t n = 0;
    int rc = fsrl_ppo_begin(c, &zero, 1.0, 1 << 20, &n);
    if (rc) return rc;
    c->in_update = false;
    if (n_out) *n_out = n;
    t->ready = false;
    if (n == 0) return 0;
    rc = tr_alloc(c, t, n);
    if (rc) return rc;
    if (cfg->norm_adv) {
        hipLaunchKernelGGL(fb_advnorm_kernel, dim3(c->cfg.n_critics), dim3(1024), 0, c->compute, c->advs, (int)n);
        HIPCHK(hipGetLastError());
    }
    rc = tr_refresh_old(c, t);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->compute));
    t->ready = true;
    return 0;
}

static int tr_stats(fsrl_ctx* c, TrState* t, int ny, double* out) {
    hipLaunchKernelGGL(fb_reduce_stats_kernel, dim3(ny * FB_NSTAT), dim3(256), 0, c->compute, t->statp, t->n_tiles,
                       ny, t->d_scal);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, t->d_scal, (size_t)ny * FB_NSTAT * sizeof(double), hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    return 0;
}

static int tr_tile(fsrl_ctx* c, TrState* t, int mode, int net0, int ny, float cr, float cc) {
    FbArgs a{};
    a.obs = c->b.obs; a.rd = t->rd; a.A1 = t->A1; a.A2 = t->A2; a.D1 = t->D1; a.D2 = t->D2; a.DO = t->DO;
    a.statp = t->statp; a.N = (int)c->N; a.rows_pad = t->rows_pad; a.mode = mode; a.net0 = net0; a.cr = cr; a.cc = cc;
    a.max_action = c->cfg.max_action;
    return dispatch_H(c->cfg.hidden, [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        hipLaunchKernelGGL((fb_tile_kernel<H, 16>), dim3(t->n_tiles, ny), dim3(4 * H), 0, c->compute, c->P, c->md, a);
        HIPCHK(hipGetLastError());
        return 0;
    });
}

static int tr_wgrad_plain(fsrl_ctx* c, TrState* t, int net0, int ny, float* out) {
    FbWgradArgs wa{};
    const size_t H = c->cfg.hidden;
    for (int y = 0; y < ny; ++y) {
        const size_t nb = (size_t)y * t->rows_pad;
        FbWgradNet& wn = wa.nets[y];
        wn.w2_ya = t->D2 + nb * H; wn.w2_xa = t->A1 + nb * H; wn.w2_yb = nullptr; wn.w2_xb = nullptr;
        wn.w1_y = t->D1 + nb * H;
        wn.w3_xa = t->A2 + nb * H; wn.w3_ya = t->DO + nb * FSRL_DOW; wn.w3_xb = nullptr; wn.w3_yb = nullptr;
        wn.b1_src = t->D1 + nb * H; wn.b2_src = t->D2 + nb * H; wn.do_src = t->DO + nb * FSRL_DOW;
        wn.net = net0 + y;
    }
    wa.obs = c->b.obs; wa.rows = t->rows_pad; wa.N = (int)c->N;
    int nsplit = 1;
    int rc = wgrad_launch<false>(c, c->md, wa, ny, c->n_dev, &nsplit);
    if (rc) return rc;
    const int begin = c->md.net[net0].begin, end = c->md.net[net0 + ny - 1].end;
    hipLaunchKernelGGL(fb_sum_parts_kernel, dim3((end - begin + 255) / 256), dim3(256), 0, c->compute, out, c->wg_parts,
                       begin, end, nsplit, c->n_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

// gradient of a scalar actor objective; returns the flat ACTOR gradient and the 8 batch means
static int tr_actor_grad(fsrl_ctx* c, TrState* t, int mode, float cr, float cc, float* g_out, double* means8) {
    int rc = tr_tile(c, t, mode, 0, 1, cr, cc);
    if (rc) return rc;
    rc = tr_wgrad_plain(c, t, 0, 1, t->Out);
    if (rc) return rc;
    if (means8) {
        rc = tr_stats(c, t, 1, means8);
        if (rc) return rc;
        for (int k = 0; k < FB_NSTAT; ++k) means8[k] /= (double)c->N;
    }
    return actor_from_dev(c, t->Out, g_out);
}

static int tr_eval_means(fsrl_ctx* c, TrState* t, double* means8) {
    int rc = tr_tile(c, t, FB_MODE_EVAL, 0, 1, 0.f, 0.f);
    if (rc) return rc;
    rc = tr_stats(c, t, 1, means8);
    if (rc) return rc;
    for (int k = 0; k < FB_NSTAT; ++k) means8[k] /= (double)c->N;
    return 0;
}

// H v for the tangent already in t->Vdev (main part + W2 mirror); result in t->Out (device)
static int tr_hvp_dev(fsrl_ctx* c, TrState* t) {
    int rc = 0;
    HvpArgs ha{};
    ha.obs = c->b.obs; ha.rd = t->rd; ha.V = t->Vdev; ha.A1 = t->A1; ha.RA1 = t->RA1; ha.A2 = t->A2; ha.RA2 = t->RA2;
    ha.D2 = t->D2; ha.RD2 = t->RD2; ha.RD1 = t->RD1; ha.DO = t->DO; ha.RDO = t->RDO; ha.N = (int)c->N;
    ha.rows_pad = t->rows_pad; ha.max_action = c->cfg.max_action;
    rc = dispatch_H(c->cfg.hidden, [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        hipLaunchKernelGGL(fb_hvp_tile_kernel<H>, dim3(t->n_tiles), dim3(4 * H), 0, c->compute, c->P, c->md, ha);
        HIPCHK(hipGetLastError());
        return 0;
    });
    if (rc) return rc;
    FbWgradArgs wa{};
    FbWgradNet& wn = wa.nets[0];
    wn.w2_ya = t->RD2; wn.w2_xa = t->A1; wn.w2_yb = t->D2; wn.w2_xb = t->RA1;     // R{dW2} = R{dz2}^T h1 + dz2^T R{h1}
    wn.w1_y = t->RD1;                                                             // R{dW1} = R{dz1}^T x
    wn.w3_xa = t->A2; wn.w3_ya = t->RDO; wn.w3_xb = t->RA2; wn.w3_yb = t->DO;     // R{dW3} = R{dout}^T h2 + dout^T R{h2}
    wn.b1_src = t->RD1; wn.b2_src = t->RD2; wn.do_src = t->RDO; wn.net = 0;
    wa.obs = c->b.obs; wa.rows = t->rows_pad; wa.N = (int)c->N;
    int nsplit = 1;
    rc = wgrad_launch<true>(c, c->md, wa, 1, c->n_dev, &nsplit);
    if (rc) return rc;
    hipLaunchKernelGGL(fb_sum_parts_kernel, dim3((c->md.net[0].end - c->md.net[0].begin + 255) / 256), dim3(256), 0,
                       c->compute, t->Out, c->wg_parts, c->md.net[0].begin, c->md.net[0].end, nsplit, c->n_dev);
    HIPCHK(hipGetLastError());
    return 0;
}
static int tr_hvp(fsrl_ctx* c, TrState* t, const float* v, float* out) {
    int rc = actor_to_dev(c, v, t->Vdev);
    if (rc) return rc;
    rc = tr_hvp_dev(c, t);
    if (rc) return rc;
    return actor_from_dev(c, t->Out, out);
}

// Line-search evaluations (policy forwards over the batch) of every repeat of the last fsrl_cpo_learn /
// fsrl_trpo_learn call; returns how many were written.
extern "C" int32_t fsrl_tr_linesearch_evals(fsrl_ctx* c, int32_t* out, int32_t cap) {
    CHECK_ARG(c && out && cap >= 0, "bad argument");
    TrState* t = tr_of(c);
    const int32_t n = std::min<int32_t>(cap, (int32_t)t->ls_iters.size());
    for (int32_t i = 0; i < n; ++i) out[i] = t->ls_iters[(size_t)i];
    return n;
}

extern "C" int fsrl_tr_grad(fsrl_ctx* c, int32_t which, float* out, int64_t n) {
    CHECK_ARG(c && out, "null argument");
    TrState* t = tr_of(c);
    if (!t->ready) return fail(FSRL_ESTATE, "fsrl_tr_grad before fsrl_tr_begin");
    CHECK_ARG(n == t->na, "expected %d actor parameters", t->na);
    HIPCHK(hipSetDevice(c->device));
    if (which == 0) return tr_actor_grad(c, t, FB_MODE_SUR, 1.0f, 0.0f, out, nullptr);
    if (which == 1) return tr_actor_grad(c, t, FB_MODE_SUR, 0.0f, -1.0f, out, nullptr);
    if (which == 2) return tr_actor_grad(c, t, FB_MODE_KL, 0.0f, 0.0f, out, nullptr);
    return fail(FSRL_EINVAL, "which must be 0, 1 or 2");
}
extern "C" int fsrl_tr_hvp(fsrl_ctx* c, const float* v, float* out, int64_t n) {
    CHECK_ARG(c && v && out, "null argument");
    TrState* t = tr_of(c);
    if (!t->ready) return fail(FSRL_ESTATE, "fsrl_tr_hvp before fsrl_tr_begin");
    CHECK_ARG(n == t->na, "expected %d actor parameters", t->na);
    HIPCHK(hipSetDevice(c->device));
    return tr_hvp(c, t, v, out);
}
extern "C" int fsrl_tr_eval(fsrl_ctx* c, double* stats8) {
    CHECK_ARG(c && stats8, "null argument");
    TrState* t = tr_of(c);
    if (!t->ready) return fail(FSRL_ESTATE, "fsrl_tr_eval before fsrl_tr_begin");
    HIPCHK(hipSetDevice(c->device));
    return tr_eval_means(c, t, stats8);
}

// ---- host float32 vector algebra (sizes ~7e4: microseconds)
static float vdot(const std::vector<float>& a, const std::vector<float>& b) {
    float s = 0.0f;
    for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
    return s;
}

// x = H^-1 g by conjugate gradients with damped HVPs (cpo.py:184-204 / trpo_lag.py:261-283)
static int tr_cg(fsrl_ctx* c, TrState* t, const std::vector<float>& g, float damping, int nsteps, float tol,
                 std::vector<float>& x) {
    // device-resident: g goes up once, x comes back once; no host synchronisation per iteration
    // (after an early convergence the remaining steps are no-ops on the device)
    const size_t n = g.size();
    x.assign(n, 0.0f);
    const int nd = c->md.net[0].end;
    {   // g in device layout (no mirror needed: cg_init writes p and its mirror)
        std::vector<float> tmp((size_t)nd, 0.0f);
        for (int i = 0; i < 7; ++i) memcpy(&tmp[c->tmap[i].dev_off], g.data() + c->tmap[i].api_off, (size_t)c->tmap[i].n * 4);
        HIPCHK(hipStreamSynchronize(c->compute));
        HIPCHK(hipMemcpy(t->cg_g, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
    }
    float* part_pz = t->cg_part; float* part_rr = t->cg_part + CG_NB;
    hipLaunchKernelGGL(cg_init_kernel, dim3(CG_NB), dim3(256), 0, c->compute, t->cg_g, t->cg_r, t->Vdev, t->cg_x, t->cg_sc,
                       part_rr, nd, c->md);
    HIPCHK(hipGetLastError());
    for (int it = 0; it < nsteps; ++it) {
        int rc = tr_hvp_dev(c, t);
        if (rc) return rc;
        hipLaunchKernelGGL(cg_pz_kernel, dim3(CG_NB), dim3(256), 0, c->compute, t->Out, t->Vdev, t->cg_sc, part_pz, nd, damping);
        hipLaunchKernelGGL(cg_xr_kernel, dim3(CG_NB), dim3(256), 0, c->compute, t->Out, t->cg_r, t->Vdev, t->cg_x, t->cg_sc,
                           part_pz, part_rr, nd, it);
        hipLaunchKernelGGL(cg_p_kernel, dim3(CG_NB), dim3(256), 0, c->compute, t->cg_r, t->Vdev, t->cg_sc, part_rr, nd, it, tol,
                           c->md);
        HIPCHK(hipGetLastError());
    }
    return actor_from_dev(c, t->cg_x, x.data());
}

static int tr_mvp(fsrl_ctx* c, TrState* t, const std::vector<float>& v, float damping, std::vector<float>& out) {
    out.resize(v.size());
    int rc = tr_hvp(c, t, v.data(), out.data());
    if (rc) return rc;
    for (size_t i = 0; i < v.size(); ++i) out[i] += v[i] * damping;
    return 0;
}

static int actor_get(fsrl_ctx* c, std::vector<float>& th) {
    th.resize((size_t)fsrl_actor_param_count(c));
    return actor_from_dev(c, c->P, th.data());
}
static int actor_set(fsrl_ctx* c, const std::vector<float>& th) { return actor_to_dev(c, th.data(), c->P); }

// critic regression: `iters` Adam steps on all critics (full batch).  vf_out[C] = last step's losses.
static int tr_critic_steps(fsrl_ctx* c, TrState* t, int iters, float l2, float* vf_out) {
    const int C = c->cfg.n_critics;
    const int begin = c->md.net[1].begin, end = c->md.net[C].end;
    for (int it = 0; it < iters; ++it) {
        int rc = tr_tile(c, t, FB_MODE_VF, 1, C, 0.f, 0.f);
        if (rc) return rc;
        rc = tr_wgrad_plain(c, t, 1, C, c->G);
        if (rc) return rc;
        if (it == iters - 1) {
            double st[FSRL_MAX_CRITICS * FB_NSTAT];
            rc = tr_stats(c, t, C, st);
            if (rc) return rc;
            for (int k = 0; k < C; ++k) {
                float vf = (float)(st[k * FB_NSTAT] / (double)c->N);
                if (l2 > 0.0f) {   // + l2 * sum(theta^2) over the critic's parameters, pre-update
                    hipLaunchKernelGGL(sumsq_range_kernel, dim3(1), dim3(256), 0, c->compute, c->P,
                                       c->md.net[1 + k].begin, c->md.net[1 + k].end, t->d_scal + 40);
                    double ss = 0;
                    HIPCHK(hipMemcpyAsync(&ss, t->d_scal + 40, sizeof(double), hipMemcpyDeviceToHost, c->compute));
                    HIPCHK(hipStreamSynchronize(c->compute));
                    vf += (float)ss * l2;
                }
                vf_out[k] = vf;
            }
        }
        t->critic_t += 1;
        const double b1 = c->cfg.beta1, b2 = c->cfg.beta2;
        const double bc1 = 1.0 - std::pow(b1, (double)t->critic_t), bc2 = 1.0 - std::pow(b2, (double)t->critic_t);
        hipLaunchKernelGGL(adam_range_kernel, dim3((end - begin + 255) / 256), dim3(256), 0, c->compute, c->P, c->M,
                           c->V, c->G, begin, end, l2, (float)(1.0 - b1), c->cfg.beta2, (float)(1.0 - b2),
                           (float)((double)t->cfg.critic_lr / bc1), (float)std::sqrt(bc2), c->cfg.adam_eps, 1, 0, c->md);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

static float actor_entropy(fsrl_ctx* c, const std::vector<float>& th) {
    float e = 0.0f;   // sum_d (0.5 + 0.5 log(2 pi) + log sigma_d); sigma_param is the first tensor
    for (int d = 0; d < c->cfg.act_dim; ++d) e += 1.4189385332046727f + std::log(std::exp(th[d]));
    return e;
}

extern "C" int fsrl_cpo_learn(fsrl_ctx* c, double ave_cost_return, int32_t repeat, float* stats_out) {
    CHECK_ARG(c && stats_out, "null argument");
    TrState* t = tr_of(c);
    if (!t->ready) return fail(FSRL_ESTATE, "fsrl_cpo_learn before fsrl_tr_begin");
    HIPCHK(hipSetDevice(c->device));
    const fsrl_tr_config& k = t->cfg;
    const float EPS = 1e-8f, delta = k.target_kl;
    const size_t n = (size_t)t->na;
    t->ls_iters.clear();
    for (int rep = 0; rep < repeat; ++rep) {
        float* st = stats_out + (size_t)rep * FSRL_CPO_NSTATS;
        int rc = tr_critic_steps(c, t, k.optim_critic_iters, k.l2_reg, st + 14);
        if (rc) return rc;
        st[16] = st[14] + st[15];
        // ---- objective, cost surrogate, KL and their gradients (cpo.py:238-254)
        std::vector<float> g(n), b(n), theta0;
        double m[FB_NSTAT];
        rc = tr_actor_grad(c, t, FB_MODE_SUR, 1.0f, 0.0f, g.data(), m);
        if (rc) return rc;
        rc = tr_actor_grad(c, t, FB_MODE_SUR, 0.0f, -1.0f, b.data(), nullptr);
        if (rc) return rc;
        rc = actor_get(c, theta0);
        if (rc) return rc;
        const float objective = (float)m[0];
        const float cost_sur = ((float)ave_cost_return + (float)m[1]) - (float)m[5];
        const float kl = (float)m[2];
        const float ent = actor_entropy(c, theta0);
        std::vector<float> Hg, Hb, approx_g, approx_b;
        rc = tr_cg(c, t, g, k.damping, k.cg_iters, 1e-8f, Hg);
        if (rc) return rc;
        rc = tr_mvp(c, t, Hg, k.damping, approx_g);
        if (rc) return rc;
        const float c_value = cost_sur - (float)k.cost_limit;
        float s_q, s_r = 0.f, s_s = 0.f, A = 0.f, B = 0.f;
        int ocase;
        if (vdot(b, b) <= EPS && c_value < 0) {
            Hb.assign(n, 0.0f);
            s_q = vdot(approx_g, Hg);
            ocase = 4;
        } else {
            rc = tr_cg(c, t, b, k.damping, k.cg_iters, 1e-8f, Hb);
            if (rc) return rc;
            rc = tr_mvp(c, t, Hb, k.damping, approx_b);
            if (rc) return rc;
            s_q = vdot(approx_g, Hg); s_r = vdot(approx_g, Hb); s_s = vdot(approx_b, Hb);
            A = s_q - s_r * s_r / s_s;
            B = 2.0f * delta - c_value * c_value / s_s;
            if (c_value < 0 && B < 0) ocase = 3;
            else if (c_value < 0 && B >= 0) ocase = 2;
            else if (c_value >= 0 && B >= 0) ocase = 1;
            else ocase = 0;
        }
        float lam, nu;
        if (ocase == 3 || ocase == 4) {
            lam = std::sqrt(s_q / (2.0f * delta));
            nu = 0.0f;
        } else if (ocase == 1 || ocase == 2) {
            const float rc_ = s_r / c_value, inf = INFINITY;
            float LA[2] = {0.0f, rc_}, LB[2] = {rc_, inf};
            if (!(c_value < 0)) { std::swap(LA[0], LB[0]); std::swap(LA[1], LB[1]); }
            auto proj = [](float x, const float* L) { return std::max(L[0], std::min(L[1], x)); };
            const float lam_a = proj(std::sqrt(A / B), LA), lam_b = proj(std::sqrt(s_q / (2.0f * delta)), LB);
            auto f_a = [&](float l) { return -0.5f * (A / (l + EPS) + B * l) - s_r * c_value / (s_s + EPS); };
            auto f_b = [&](float l) { return -0.5f * (s_q / (l + EPS) + 2.0f * delta * l); };
            lam = (f_a(lam_a) >= f_b(lam_b)) ? lam_a : lam_b;
            nu = std::max(0.0f, lam * c_value - s_r) / (s_s + EPS);
        } else {
            nu = std::sqrt(2.0f * delta / (s_s + EPS));
            lam = 0.0f;
        }
        // ---- line search (cpo.py:306-333); on failure the LAST tried theta stays in place
        std::vector<float> dir(n);
        for (size_t i = 0; i < n; ++i)
            dir[i] = (ocase > 0) ? (1.0f / (lam + EPS)) * (Hg[i] + nu * Hb[i]) : nu * Hb[i];
        const float nrm = std::sqrt(vdot(dir, dir));
        for (size_t i = 0; i < n; ++i) dir[i] /= nrm;
        double beta = 1.0;
        int evals = 0;
        if (!std::isnan(lam)) {
            std::vector<float> th(n);
            for (int bt = 0; bt < k.max_backtracks; ++bt) {
                ++evals;
                const float bf = (float)beta;
                for (size_t i = 0; i < n; ++i) th[i] = bf * dir[i] + theta0[i];
                rc = actor_set(c, th);
                if (rc) return rc;
                double e8[FB_NSTAT];
                rc = tr_eval_means(c, t, e8);
                if (rc) return rc;
                const float new_kl = (float)e8[2], new_obj = (float)e8[0];
                const float new_cs = ((float)ave_cost_return + (float)e8[1]) - (float)e8[5];
                const bool ok = ((double)new_kl <= (double)delta) && (ocase > 1 ? new_obj > objective : true) &&
                                ((double)(new_cs - cost_sur) <= std::max(-(double)c_value, 0.0));
                if (ok) break;
                beta *= (double)k.backtrack_coeff;
            }
        }
        st[0] = kl; st[1] = ent; st[2] = objective; st[3] = cost_sur; st[4] = A; st[5] = B; st[6] = c_value;
        st[7] = s_q; st[8] = s_r; st[9] = s_s; st[10] = lam; st[11] = nu; st[12] = (float)ocase; st[13] = (float)beta;
        t->ls_iters.push_back(evals);
    }
    return 0;
}

extern "C" int fsrl_trpo_learn(fsrl_ctx* c, const double* lagrangians, double rescaling, int32_t repeat,
                               float* stats_out) {
    CHECK_ARG(c && stats_out, "null argument");
    TrState* t = tr_of(c);
    if (!t->ready) return fail(FSRL_ESTATE, "fsrl_trpo_learn before fsrl_tr_begin");
    HIPCHK(hipSetDevice(c->device));
    const fsrl_tr_config& k = t->cfg;
    const size_t n = (size_t)t->na;
    const float lam0 = (c->cfg.use_lagrangian && lagrangians) ? (float)lagrangians[0] : 0.0f;
    const float resc = (float)rescaling, delta = k.target_kl;
    t->ls_iters.clear();
    for (int rep = 0; rep < repeat; ++rep) {
        float* st = stats_out + (size_t)rep * FSRL_TRPO_NSTATS;
        int rc = tr_refresh_old(c, t);     // old_dist = pi_theta (detached), trpo_lag.py:189-190
        if (rc) return rc;
        std::vector<float> g(n), theta0, x, dir(n), Hd;
        double m[FB_NSTAT];
        // loss_actor = rescaling * ( -mean(ratio A_r) + lambda * mean(ratio A_c) )
        rc = tr_actor_grad(c, t, FB_MODE_SUR, -resc, resc * lam0, g.data(), m);
        if (rc) return rc;
        rc = actor_get(c, theta0);
        if (rc) return rc;
        const float loss_rew = -(float)m[0];
        const float loss_safety = c->cfg.use_lagrangian ? (float)m[1] * lam0 : 0.0f;
        const float loss_actor = resc * (loss_rew + loss_safety);
        const float ent = actor_entropy(c, theta0);
        rc = tr_cg(c, t, g, k.damping, k.cg_iters, 1e-10f, x);
        if (rc) return rc;
        for (size_t i = 0; i < n; ++i) dir[i] = -x[i];
        rc = tr_mvp(c, t, dir, k.damping, Hd);
        if (rc) return rc;
        float step = std::sqrt(2.0f * delta / vdot(dir, Hd));
        float kl = 0.0f;
        std::vector<float> th(n);
        int evals = 0;
        for (int i = 0; i < k.max_backtracks; ++i) {
            ++evals;
            for (size_t j = 0; j < n; ++j) th[j] = theta0[j] + step * dir[j];
            rc = actor_set(c, th);
            if (rc) return rc;
            double e8[FB_NSTAT];
            rc = tr_eval_means(c, t, e8);
            if (rc) return rc;
            kl = (float)e8[2];
            const float loss_new = resc * (-(float)e8[0] + (c->cfg.use_lagrangian ? (float)e8[1] * lam0 : 0.0f));
            if ((double)kl < (double)delta && loss_new < loss_actor) break;
            else if (i < k.max_backtracks - 1) step *= k.backtrack_coeff;
            else step = 0.0f;   // total failure: the last tried parameters stay (trpo_lag.py:225-227)
        }
        float vf[FSRL_MAX_CRITICS] = {0, 0, 0, 0};
        rc = tr_critic_steps(c, t, k.optim_critic_iters, 0.0f, vf);
        if (rc) return rc;
        st[0] = resc; st[1] = lam0; st[2] = loss_safety; st[3] = loss_rew; st[4] = loss_actor;
        st[5] = vf[0]; st[6] = vf[1]; st[7] = vf[0] + vf[1]; st[8] = kl; st[9] = step; st[10] = ent;
        t->ls_iters.push_back(evals);
    }
    return 0;
}

// ====================================================================================== FOCOPS
struct FocState {
    fsrl_focops_config cfg{};
    double nu = 0.0, nu_loss = 0.0;
    int64_t t_actor = 0, t_critic = 0;
    float *statp_pi = nullptr, *psq = nullptr, *gsq = nullptr, *sig_stash = nullptr;   // statp_pi: [tiles][3 networks][FB_NSTAT]
    int cap_tiles = 0, cap_psq = 0;
};
static void foc_free(fsrl_ctx* c) {
    FocState* f = c->foc;
    if (!f) return;
    for (float* p : {f->statp_pi, f->psq, f->gsq, f->sig_stash}) if (p) (void)hipFree(p);
    delete f;
    c->foc = nullptr;
}
extern "C" int fsrl_focops_init(fsrl_ctx* c, const fsrl_focops_config* cfg) {
    CHECK_ARG(c && cfg, "null argument");
    CHECK_ARG(c->cfg.algo == FSRL_ALGO_FOCOPS, "context was not created with FSRL_ALGO_FOCOPS");
    CHECK_ARG(c->cfg.n_critics == 2, "FOCOPS uses a reward and a cost critic");
    CHECK_ARG(cfg->tem_lambda > 0.0f, "tem_lambda must be positive");
    foc_free(c);
    c->foc = new FocState();
    c->foc->cfg = *cfg;
    return 0;
}
extern "C" int fsrl_focops_set_nu(fsrl_ctx* c, double nu, double nu_loss) {
    CHECK_ARG(c, "null ctx");
    if (!c->foc) return fail(FSRL_ESTATE, "fsrl_focops_init first");
    c->foc->nu = nu; c->foc->nu_loss = nu_loss;
    return 0;
}

// one pass of FOCOPS minibatch steps over the batch prepared by fsrl_ppo_pass (permuted rows, per-minibatch
// normalised advantages, old means / stds in the row data).  Per minibatch (focops.py:226-241):
//   critics:  fb_tile(VF, 2 nets) -> fb_wgrad -> Adam per critic (+ l2, records sum(theta^2))
//   actor:    fb_tile(FOCOPS)     -> fb_wgrad -> sum (+ ||g||^2 partials) -> clip + Adam
//   stats row + pass KL sum (focops_finalize_kernel)
static int focops_pass(fsrl_ctx* c, int32_t* stopped_out) {
    FocState* f = c->foc;
    if (!f) return fail(FSRL_ESTATE, "fsrl_focops_init first");
    hipStream_t s = c->compute;
    const int H = c->cfg.hidden, Do = c->cfg.obs_dim;
    const int nmb = (int)c->mb_start.size();
    const int max_tiles = c->mbp_max / 16;
    const int nb_c0 = (c->md.net[1].end - c->md.net[1].begin + 255) / 256, nb_c1 = (c->md.net[2].end - c->md.net[2].begin + 255) / 256;
    const int nb_a = (c->md.net[0].end - c->md.net[0].begin + 255) / 256;
    if (f->cap_tiles < max_tiles || f->cap_psq < nb_c0 + nb_c1) {
        HIPCHK(hipStreamSynchronize(s));
        for (float** p : {&f->statp_pi, &f->psq, &f->gsq}) { if (*p) HIPCHK(hipFree(*p)); *p = nullptr; }
        HIPCHK(hipMalloc(&f->statp_pi, (size_t)(4 * max_tiles + 4) * 3 * FB_NSTAT * 4));      // three networks per tile
        if (!f->sig_stash) HIPCHK(hipMalloc(&f->sig_stash, FSRL_MAX_ACT * 4));
        HIPCHK(hipMalloc(&f->psq, (size_t)(nb_c0 + nb_c1) * 4));
        HIPCHK(hipMalloc(&f->gsq, (size_t)nb_a * 4));
        f->cap_tiles = max_tiles; f->cap_psq = nb_c0 + nb_c1;
    }
    const double b1 = c->cfg.beta1, b2 = c->cfg.beta2;
    // One minibatch step = four launches: the activation side of all three networks (actor: FOCOPS loss head, critics:
    // regression head), their weight gradients, then the parameter side in two (focops_prep_kernel, focops_step_kernel).
    // The critics' and the actor's updates do not read each other's parameters, so one launch each is the reference's
    // critics_loss -> policy_loss order (focops.py:236-246) with nothing reordered inside a network.
    for (int mb = 0; mb < nmb; ++mb) {
        const int start = c->mb_start[(size_t)mb], size = c->mb_size[(size_t)mb];
        const int tiles = (size + 15) / 16, rows_pad = tiles * 16;
        FbArgs a{};
        a.obs = c->obs_p + (size_t)start * Do; a.rd = c->rd_p + (size_t)start * FSRL_RD;
        a.A1 = c->A1; a.A2 = c->A2; a.D1 = c->D1; a.D2 = c->D2; a.DO = c->DO;
        a.N = size; a.rows_pad = rows_pad; a.max_action = c->cfg.max_action;
        a.cr = 1.0f / f->cfg.tem_lambda; a.cc = (float)f->nu; a.eta = f->cfg.eta;
        a.mode = FB_MODE_FOCOPS; a.net0 = 0; a.statp = f->statp_pi;      // [tiles][3][FB_NSTAT]
        const bool rows4 = 4 * tiles * 3 <= c->n_cus;
        int rc = dispatch_H(H, [&](auto hc) {
            constexpr int HH = decltype(hc)::value;
            if (rows4) hipLaunchKernelGGL((fb_tile_kernel<HH, 4>), dim3(4 * tiles, 3), dim3(4 * HH), 0, s, c->P, c->md, a);
            else hipLaunchKernelGGL((fb_tile_kernel<HH, 16>), dim3(tiles, 3), dim3(4 * HH), 0, s, c->P, c->md, a);
            HIPCHK(hipGetLastError());
            return 0;
        });
        if (rc) return rc;
        FbWgradArgs wa{};
        for (int y = 0; y < 3; ++y) {
            const size_t nb = (size_t)y * rows_pad;
            FbWgradNet& wn = wa.nets[y];
            wn.w2_ya = c->D2 + nb * H; wn.w2_xa = c->A1 + nb * H; wn.w1_y = c->D1 + nb * H;
            wn.w3_xa = c->A2 + nb * H; wn.w3_ya = c->DO + nb * FSRL_DOW;
            wn.b1_src = c->D1 + nb * H; wn.b2_src = c->D2 + nb * H; wn.do_src = c->DO + nb * FSRL_DOW;
            wn.net = y;
        }
        wa.obs = a.obs; wa.rows = rows_pad; wa.N = size;
        int nsplit = 1;
        rc = wgrad_launch<false>(c, c->md, wa, 3, c->n_dev, &nsplit);
        if (rc) return rc;
        f->t_critic += 1; f->t_actor += 1;
        FocopsStepArgs sa{};
        sa.P = c->P; sa.M = c->M; sa.V = c->V; sa.parts = c->wg_parts; sa.nparts = nsplit; sa.stride = c->n_dev;
        sa.G = c->G; sa.gsq = f->gsq; sa.psq = f->psq; sa.sig_stash = f->sig_stash;
        sa.nb_a = nb_a; sa.nb_c0 = nb_c0; sa.nb_c1 = nb_c1; sa.max_norm = f->cfg.max_grad_norm; sa.l2 = f->cfg.l2_reg;
        sa.one_minus_b1 = (float)(1.0 - b1); sa.beta2 = c->cfg.beta2; sa.one_minus_b2 = (float)(1.0 - b2);
        sa.adam_eps = c->cfg.adam_eps;
        sa.step_a = (float)((double)f->cfg.actor_lr / (1.0 - std::pow(b1, (double)f->t_actor)));
        sa.bc2s_a = (float)std::sqrt(1.0 - std::pow(b2, (double)f->t_actor));
        sa.step_c = (float)((double)f->cfg.critic_lr / (1.0 - std::pow(b1, (double)f->t_critic)));
        sa.bc2s_c = (float)std::sqrt(1.0 - std::pow(b2, (double)f->t_critic));
        // ---- logged row, pass KL sum, pass-level early stop flag (the extra block of the step launch)
        FocopsFinalArgs& fa = sa.fin;
        fa.statp_vf = f->statp_pi + FB_NSTAT; fa.statp_pi = f->statp_pi; fa.vf_stride = 3; fa.pi_stride = 3;
        fa.psq0 = f->psq; fa.psq1 = f->psq + nb_c0; fa.n_psq0 = nb_c0; fa.n_psq1 = nb_c1;
        fa.P = f->sig_stash; fa.sigma_off = 0; fa.Da = c->cfg.act_dim;          // entropy of the PRE-update policy
        fa.stats = c->d_stats + (size_t)(c->n_steps + mb) * FSRL_PPO_NSTATS; fa.ctrl = c->ctrl;
        fa.n_tiles = rows4 ? 4 * tiles : tiles; fa.n_tiles_pi = fa.n_tiles; fa.mb = size;
        fa.first_in_pass = mb == 0; fa.last_in_pass = mb == nmb - 1;
        fa.iters_in_pass = nmb; fa.pass = (int)c->pass_index; fa.l2 = f->cfg.l2_reg; fa.nu_loss = (float)f->nu_loss;
        fa.nu_value = (float)f->nu; fa.delta = f->cfg.delta;
        const int nb_all = nb_a + nb_c0 + nb_c1;
        hipLaunchKernelGGL(focops_prep_kernel, dim3(nb_all), dim3(256), 0, s, c->md, sa);
        hipLaunchKernelGGL(focops_step_kernel, dim3(nb_all + 1), dim3(256), 0, s, c->md, sa);
        HIPCHK(hipGetLastError());
    }
    c->n_steps += nmb;
    c->pass_index += 1;
    HIPCHK(hipMemcpyAsync(c->h_ctrl, c->ctrl, sizeof(CtrlBlock), hipMemcpyDeviceToHost, s));   // the reference's `break`
    HIPCHK(hipStreamSynchronize(s));
    if (c->h_ctrl->stopped_after != INT_MAX && stopped_out) *stopped_out = 1;
    return 0;
}

// ====================================================================================== SAC-Lagrangian
#include "kernels_sac.hpp"
#include "kernels_cvpo.hpp"

struct SacState {
    fsrl_sac_config cfg{};
    ModelDesc mda{}, mdq{};
    std::vector<TensorMap> tmap_a, tmap_q;     // API order -> device offsets
    int na_api = 0, nq_api = 0, na_dev = 0, nq_dev = 0;
    float *PA = nullptr, *MA = nullptr, *VA = nullptr;          // gradients live in the split-K partial buffers
    float *PQ = nullptr, *PQT = nullptr, *MQ = nullptr, *VQ = nullptr;
    float* PAT = nullptr;                      // target actor (DDPG-Lag mode only)
    int n_q = 4;                               // Q-networks: 4 (two double critics) or 2 (DDPG-Lag)
    bool ddpg = false;
    SacScalars* sc = nullptr;
    int64_t t_actor = 0, t_critic = 0;
    // per-batch buffers
    int cap_B = 0, n_tiles = 0;               // n_tiles = 16-row tiles of the batch
    bool q_rows4 = false, a_rows4 = false;    // 4-row tile variants for the Q / actor launches
    int *d_idx = nullptr, *d_chain = nullptr; uint8_t* d_end = nullptr;
    int *h_idx = nullptr, *h_chain = nullptr; uint8_t* h_end = nullptr;          // pinned
    float *XQ = nullptr, *OBS = nullptr, *OBSN = nullptr, *XN = nullptr, *XP = nullptr;
    float *eps_t = nullptr, *eps_p = nullptr, *h_eps = nullptr;                   // device / pinned
    float *LPN = nullptr, *LP = nullptr, *QT = nullptr, *QP = nullptr, *Y = nullptr, *DA = nullptr;
    float *A1 = nullptr, *A2 = nullptr, *D1 = nullptr, *D2 = nullptr, *DO = nullptr;   // [4][Bpad]
    float *stq = nullptr, *stdin_ = nullptr, *stpi = nullptr;   // stdin_: per-tile scratch of the Q_DIN launch (unused sums)
    float* d_stats = nullptr;                 // ring [SAC_RING][FSRL_SAC_NSTATS]: one row per update
    int64_t n_updates = 0, n_drained = 0;
    uint64_t key = 0x243F6A8885A308D3ull;     // Philox key of the library-RNG mode
    SacBook* d_book = nullptr; SacBook* h_book = nullptr;      // device / pinned sub-buffer bookkeeping
    uint64_t book_version = 0;
    int last_B = 0;
    int nstats = FSRL_SAC_NSTATS_K;           // floats per row of the statistics ring
    // ---- CVPO mode (fsrl_cvpo_init): Gaussian actor + actor_old (PAT), E-step / M-step state
    bool cvpo = false;
    fsrl_cvpo_config ccfg{};
    CvpoScalars* csc = nullptr;
    int n_tiles_k = 0; bool k_rows4 = false;  // tiles of the K*B particle launch
    float *MU_OLD = nullptr, *STD_OLD = nullptr, *XK = nullptr, *QK = nullptr, *q0 = nullptr, *q1 = nullptr;
    float *Wk = nullptr, *eps_k = nullptr, *h_epsk = nullptr, *stqk = nullptr;
};
static constexpr int SAC_RING = 4096;

static SacState* sac_of(fsrl_ctx* c) { return reinterpret_cast<SacState*>(c->sac); }

static void sac_layout(fsrl_ctx* c, SacState* s) {
    const int Do = c->cfg.obs_dim, Da = c->cfg.act_dim, H = c->cfg.hidden, Din = Do + Da;
    // ---- actor: device W3 = [Wmu ; Wsig], b3 = [bmu ; bsig]
    {
        ModelDesc& md = s->mda;
        md.Do = Do; md.Da = Da; md.H = H; md.n_nets = 1;
        int dev = 0, api = 0;
        auto place = [&](int n) { int o = dev; dev = round_up(dev + n, 64); return o; };
        NetOff& no = md.net[0];
        no.sigma = -1; no.out = 2 * Da; no.begin = 0;
        no.W1 = place(H * Do); no.b1 = place(H); no.W2 = place(H * H); no.b2 = place(H);
        const int heads = s->ddpg ? 1 : 2;                        // DDPG-Lag: the mean head only
        no.out = heads * Da;
        no.W3 = place(heads * Da * H); no.b3 = place(heads * Da);
        no.end = dev;
        auto add = [&](int dev_off, int n) { s->tmap_a.push_back(TensorMap{api, dev_off, n}); api += n; };
        add(no.W1, H * Do); add(no.b1, H); add(no.W2, H * H); add(no.b2, H);
        add(no.W3, Da * H); add(no.b3, Da);                       // mu head
        if (!s->ddpg) { add(no.W3 + Da * H, Da * H); add(no.b3 + Da, Da); }   // sigma head
        s->na_api = api; s->na_dev = round_up(dev, 1024);
        no.W2f = s->na_dev;                                       // forward-fragment mirror behind the main vector
    }
    // ---- four Q-nets: device order Qr1, Qr2, Qc1, Qc2 ; API order per DoubleCritic: pre1 pre2 last1 last2
    {
        ModelDesc& md = s->mdq;
        md.Do = Din; md.Da = Da; md.H = H; md.n_nets = s->n_q;
        int dev = 0;
        auto place = [&](int n) { int o = dev; dev = round_up(dev + n, 64); return o; };
        for (int n = 0; n < s->n_q; ++n) {
            NetOff& no = md.net[n];
            no.sigma = -1; no.out = 1; no.begin = dev;
            no.W1 = place(H * Din); no.b1 = place(H); no.W2 = place(H * H); no.b2 = place(H);
            no.W3 = place(H); no.b3 = place(1);
            no.end = dev;
        }
        int api = 0;
        auto add = [&](int dev_off, int n) { s->tmap_q.push_back(TensorMap{api, dev_off, n}); api += n; };
        for (int i = 0; i < 2 && s->n_q == 4; ++i) {
            for (int j = 0; j < 2; ++j) {
                const NetOff& no = md.net[2 * i + j];
                add(no.W1, H * Din); add(no.b1, H); add(no.W2, H * H); add(no.b2, H);
            }
            for (int j = 0; j < 2; ++j) { const NetOff& no = md.net[2 * i + j]; add(no.W3, H); add(no.b3, 1); }
        }
        for (int i = 0; i < 2 && s->n_q == 2; ++i) {     // tianshou Critic / SingleCritic: preprocess MLP then the last layer
            const NetOff& no = md.net[i];
            add(no.W1, H * Din); add(no.b1, H); add(no.W2, H * H); add(no.b2, H); add(no.W3, H); add(no.b3, 1);
        }
        s->nq_api = api; s->nq_dev = round_up(dev, 1024);
        for (int n = 0; n < s->n_q; ++n) md.net[n].W2f = s->nq_dev + n * H * H;
    }
}

extern "C" int fsrl_sac_init(fsrl_ctx* c, const fsrl_sac_config* cfg) {
    CHECK_ARG(c && cfg, "null argument");
    CHECK_ARG(c->cfg.algo == FSRL_ALGO_SAC_LAG, "context was not created with FSRL_ALGO_SAC_LAG");
    CHECK_ARG(cfg->deterministic ? c->cfg.act_dim <= 16 : c->cfg.act_dim <= 8,
              "the actor head has at most 16 outputs (act_dim <= 8 for SAC, <= 16 for DDPG)");
    CHECK_ARG(c->cfg.obs_dim + c->cfg.act_dim <= FSRL_MAX_OBS, "obs_dim + act_dim too large");
    CHECK_ARG(cfg->n_step >= 1 && cfg->n_step <= 8, "n_step must be in [1, 8]");
    CHECK_ARG(cfg->tau >= 0.0f && cfg->tau <= 1.0f, "tau should be in [0, 1]");
    HIPCHK(hipSetDevice(c->device));
    if (c->sac) sac_free(c);
    SacState* s = new SacState();
    c->sac = s;
    s->cfg = *cfg;
    s->ddpg = cfg->deterministic != 0;
    s->n_q = s->ddpg ? 2 : 4;
    if (s->ddpg) { s->cfg.auto_alpha = 0; s->cfg.alpha = 0.0f; }      // no entropy term anywhere
    sac_layout(c, s);
    const size_t HH = (size_t)c->cfg.hidden * c->cfg.hidden;
    const size_t ab = ((size_t)s->na_dev + HH) * 4, qb = ((size_t)s->nq_dev + 4 * HH) * 4;   // + W2 mirrors (used in P only)
    for (float** p : {&s->PA, &s->MA, &s->VA, &s->PAT}) { HIPCHK(hipMalloc(p, ab)); HIPCHK(hipMemsetAsync(*p, 0, ab, c->compute)); }
    for (float** p : {&s->PQ, &s->PQT, &s->MQ, &s->VQ}) { HIPCHK(hipMalloc(p, qb)); HIPCHK(hipMemsetAsync(*p, 0, qb, c->compute)); }
    HIPCHK(hipStreamSynchronize(c->compute));
    HIPCHK(hipMalloc(&s->sc, sizeof(SacScalars)));
    SacScalars init{cfg->auto_alpha ? 1.0f : cfg->alpha, 0.0f, 0.0f, 0.0f, 0, 0};
    HIPCHK(hipMemcpy(s->sc, &init, sizeof(init), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&s->d_stats, (size_t)SAC_RING * s->nstats * 4));
    HIPCHK(hipMalloc(&s->d_book, (size_t)c->cfg.env_num * sizeof(SacBook)));
    HIPCHK(hipHostMalloc(&s->h_book, (size_t)c->cfg.env_num * sizeof(SacBook)));
    return 0;
}

static void sac_free(fsrl_ctx* c) {
    SacState* s = sac_of(c);
    if (!s) return;
    for (void* p : {(void*)s->PAT, (void*)s->PA, (void*)s->MA, (void*)s->VA, (void*)s->PQ, (void*)s->PQT, (void*)s->MQ,
                    (void*)s->VQ, (void*)s->sc, (void*)s->d_idx, (void*)s->d_chain, (void*)s->d_end,
                    (void*)s->XQ, (void*)s->OBS, (void*)s->OBSN, (void*)s->XN, (void*)s->XP, (void*)s->eps_t,
                    (void*)s->eps_p, (void*)s->LPN, (void*)s->LP, (void*)s->QT, (void*)s->QP, (void*)s->Y,
                    (void*)s->DA, (void*)s->A1, (void*)s->A2, (void*)s->D1, (void*)s->D2, (void*)s->DO,
                    (void*)s->stq, (void*)s->stdin_, (void*)s->stpi, (void*)s->d_stats, (void*)s->d_book,
                    (void*)s->csc, (void*)s->MU_OLD, (void*)s->STD_OLD, (void*)s->XK, (void*)s->QK, (void*)s->q0,
                    (void*)s->q1, (void*)s->Wk, (void*)s->eps_k, (void*)s->stqk})
        if (p) (void)hipFree(p);
    for (void* p : {(void*)s->h_idx, (void*)s->h_chain, (void*)s->h_end, (void*)s->h_eps, (void*)s->h_book, (void*)s->h_epsk})
        if (p) (void)hipHostFree(p);
    delete s;
    c->sac = nullptr;
}

extern "C" int64_t fsrl_sac_param_count(const fsrl_ctx* c, int32_t which) {
    if (!c || !c->sac) return 0;
    const SacState* s = reinterpret_cast<const SacState*>(c->sac);
    return which == 0 ? s->na_api : s->nq_api;
}

static int sac_copy(fsrl_ctx* c, const std::vector<TensorMap>& tm, const ModelDesc& md, int n_dev, float* dev,
                    const float* in, float* out) {
    const size_t n_alloc = (size_t)n_dev + (size_t)md.n_nets * md.H * md.H;
    std::vector<float> tmp(in ? n_alloc : (size_t)n_dev, 0.0f);
    HIPCHK(hipStreamSynchronize(c->compute));
    if (in) {
        for (const TensorMap& t : tm) memcpy(&tmp[t.dev_off], in + t.api_off, (size_t)t.n * 4);
        fill_mirrors(md, tmp);
        HIPCHK(hipMemcpy(dev, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
    } else {
        HIPCHK(hipMemcpy(tmp.data(), dev, tmp.size() * 4, hipMemcpyDeviceToHost));
        for (const TensorMap& t : tm) memcpy(out + t.api_off, &tmp[t.dev_off], (size_t)t.n * 4);
    }
    return 0;
}

extern "C" int fsrl_sac_params_set(fsrl_ctx* c, const float* actor, int64_t na, const float* critics, int64_t nc,
                                   float log_alpha) {
    CHECK_ARG(c && actor && critics, "null argument");
    SacState* s = sac_of(c);
    if (!s) return fail(FSRL_ESTATE, "fsrl_sac_init first");
    CHECK_ARG(na == s->na_api && nc == s->nq_api, "expected %d actor / %d critic parameters", s->na_api, s->nq_api);
    HIPCHK(hipSetDevice(c->device));
    int rc = sac_copy(c, s->tmap_a, s->mda, s->na_dev, s->PA, actor, nullptr);
    if (rc) return rc;
    rc = sac_copy(c, s->tmap_q, s->mdq, s->nq_dev, s->PQ, critics, nullptr);
    if (rc) return rc;
    HIPCHK(hipMemcpy(s->PQT, s->PQ, ((size_t)s->nq_dev + 4 * (size_t)c->cfg.hidden * c->cfg.hidden) * 4,
                     hipMemcpyDeviceToDevice));   // critics_old = deepcopy (with the W2 mirrors)
    HIPCHK(hipMemcpy(s->PAT, s->PA, ((size_t)s->na_dev + (size_t)c->cfg.hidden * c->cfg.hidden) * 4,
                     hipMemcpyDeviceToDevice));   // actor_old = deepcopy (DDPG-Lag)
    for (float* p : {s->MA, s->VA}) HIPCHK(hipMemsetAsync(p, 0, (size_t)s->na_dev * 4, c->compute));
    for (float* p : {s->MQ, s->VQ}) HIPCHK(hipMemsetAsync(p, 0, (size_t)s->nq_dev * 4, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    s->t_actor = s->t_critic = 0;
    SacScalars init{s->cfg.auto_alpha ? std::exp(log_alpha) : s->cfg.alpha, log_alpha, 0.0f, 0.0f, 0, 0};
    HIPCHK(hipMemcpy(s->sc, &init, sizeof(init), hipMemcpyHostToDevice));
    return 0;
}

// Overwrite ONE parameter set (checkpoint load): which = 0 actor, 1 critics, 2 critics_old, 3 actor_old.
// Unlike fsrl_sac_params_set nothing else changes (targets, Adam moments, step counts stay).
extern "C" int fsrl_sac_params_put(fsrl_ctx* c, int32_t which, const float* in, int64_t n) {
    CHECK_ARG(c && in, "null argument");
    SacState* s = sac_of(c);
    if (!s) return fail(FSRL_ESTATE, "fsrl_sac_init first");
    CHECK_ARG(which >= 0 && which <= 3, "which must be 0..3");
    CHECK_ARG(which != 3 || s->ddpg || s->cvpo, "actor_old exists in the DDPG-Lagrangian and CVPO modes only");
    HIPCHK(hipSetDevice(c->device));
    if (which == 0 || which == 3) {
        CHECK_ARG(n == s->na_api, "expected %d actor parameters", s->na_api);
        return sac_copy(c, s->tmap_a, s->mda, s->na_dev, which == 0 ? s->PA : s->PAT, in, nullptr);
    }
    CHECK_ARG(n == s->nq_api, "expected %d critic parameters", s->nq_api);
    return sac_copy(c, s->tmap_q, s->mdq, s->nq_dev, which == 1 ? s->PQ : s->PQT, in, nullptr);
}

extern "C" int fsrl_sac_params_get(fsrl_ctx* c, int32_t which, float* out, int64_t n, float* alpha_out) {
    CHECK_ARG(c && out, "null argument");
    SacState* s = sac_of(c);
    if (!s) return fail(FSRL_ESTATE, "fsrl_sac_init first");
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if (which == 0 || which == 3) {
        CHECK_ARG(n == s->na_api, "bad size");
        rc = sac_copy(c, s->tmap_a, s->mda, s->na_dev, which == 0 ? s->PA : s->PAT, nullptr, out);
    }
    else { CHECK_ARG(n == s->nq_api, "bad size"); rc = sac_copy(c, s->tmap_q, s->mdq, s->nq_dev, which == 1 ? s->PQ : s->PQT, nullptr, out); }
    if (rc) return rc;
    if (alpha_out) {
        SacScalars sc;
        HIPCHK(hipMemcpy(&sc, s->sc, sizeof(sc), hipMemcpyDeviceToHost));
        *alpha_out = s->cfg.auto_alpha ? sc.alpha : s->cfg.alpha;
    }
    return 0;
}

static int sac_alloc_batch(fsrl_ctx* c, SacState* s, int B) {
    s->n_tiles = (B + 15) / 16;
    s->q_rows4 = 4 * s->n_tiles * s->n_q <= c->n_cus && !getenv("FSRL_TILE16");
    s->a_rows4 = 4 * s->n_tiles <= c->n_cus && !getenv("FSRL_TILE16");
    if (s->cvpo) {
        s->n_tiles_k = (B * s->ccfg.sample_act_num + 15) / 16;
        s->k_rows4 = 4 * s->n_tiles_k * s->n_q <= c->n_cus && !getenv("FSRL_TILE16");
    }
    if (B <= s->cap_B) return 0;
    HIPCHK(hipStreamSynchronize(c->compute));
    const int Do = c->cfg.obs_dim, Da = c->cfg.act_dim, Din = Do + Da, H = c->cfg.hidden, ns = s->cfg.n_step;
    const size_t Bp = (size_t)s->n_tiles * 16 + 64;
    auto re = [&](auto** p, size_t bytes) -> int {
        if (*p) HIPCHK(hipFree(*p));
        *p = nullptr;
        HIPCHK(hipMalloc(p, bytes));
        HIPCHK(hipMemsetAsync(*p, 0, bytes, c->compute));
        return 0;
    };
    auto reh = [&](auto** p, size_t bytes) -> int {
        if (*p) HIPCHK(hipHostFree(*p));
        *p = nullptr;
        HIPCHK(hipHostMalloc(p, bytes));
        return 0;
    };
    int rc = 0;
    rc |= re(&s->d_idx, Bp * 4); rc |= re(&s->d_chain, Bp * ns * 4); rc |= re(&s->d_end, Bp * ns);
    rc |= reh(&s->h_idx, Bp * 4); rc |= reh(&s->h_chain, Bp * ns * 4); rc |= reh(&s->h_end, Bp * ns);
    rc |= reh(&s->h_eps, Bp * Da * 4 * 2);
    rc |= re(&s->XQ, Bp * Din * 4); rc |= re(&s->XN, Bp * Din * 4); rc |= re(&s->XP, Bp * Din * 4);
    rc |= re(&s->OBS, Bp * Do * 4); rc |= re(&s->OBSN, Bp * Do * 4);
    rc |= re(&s->eps_t, Bp * Da * 4); rc |= re(&s->eps_p, Bp * Da * 4);
    rc |= re(&s->LPN, Bp * 4); rc |= re(&s->LP, Bp * 4); rc |= re(&s->QT, 4 * Bp * 4); rc |= re(&s->QP, 4 * Bp * 4);
    rc |= re(&s->Y, 2 * Bp * 4); rc |= re(&s->DA, 4 * Bp * Da * 4);
    rc |= re(&s->A1, 4 * Bp * H * 4); rc |= re(&s->A2, 4 * Bp * H * 4); rc |= re(&s->D1, 4 * Bp * H * 4);
    rc |= re(&s->D2, 4 * Bp * H * 4); rc |= re(&s->DO, 4 * Bp * FSRL_DOW * 4);
    // per-tile partial statistics: room for 4-row tiles (4 x the 16-row tile count)
    rc |= re(&s->stq, (size_t)(4 * s->n_tiles + 4) * 4 * FB_NSTAT * 4); rc |= re(&s->stdin_, (size_t)(4 * s->n_tiles + 4) * 4 * FB_NSTAT * 4);
    rc |= re(&s->stpi, (size_t)(4 * s->n_tiles + 4) * FB_NSTAT * 4);
    if (s->cvpo) {
        const size_t K = (size_t)s->ccfg.sample_act_num, KB = K * Bp;
        rc |= re(&s->MU_OLD, Bp * Da * 4); rc |= re(&s->STD_OLD, Bp * Da * 4);
        rc |= re(&s->XK, KB * Din * 4); rc |= re(&s->QK, 4 * KB * 4); rc |= re(&s->q0, KB * 4); rc |= re(&s->q1, KB * 4);
        rc |= re(&s->Wk, KB * 4); rc |= re(&s->eps_k, KB * Da * 4); rc |= reh(&s->h_epsk, KB * Da * 4);
        rc |= re(&s->stqk, (size_t)(4 * ((KB + 15) / 16) + 4) * 4 * FB_NSTAT * 4);
    }
    if (rc) return FSRL_EHIP;
    s->cap_B = B;
    return 0;
}

// tianshou ReplayBuffer.next inside the owning sub-buffer
static inline int64_t store_next(const fsrl_ctx* c, int64_t idx) {
    const int64_t e = idx / c->sub_size, local = idx % c->sub_size;
    const EnvBook& eb = c->env[(size_t)e];
    const bool end = c->h_flags[(size_t)idx] != 0 || local == eb.last_index;
    if (end || eb.size == 0) return idx;
    return e * c->sub_size + (local + 1) % eb.size;
}
static inline bool store_end_flag(const fsrl_ctx* c, int64_t idx) {
    if (c->h_flags[(size_t)idx] != 0) return true;
    const int64_t e = idx / c->sub_size, local = idx % c->sub_size;
    const EnvBook& eb = c->env[(size_t)e];
    return eb.size > 0 && local == (eb.index - 1 + eb.size) % eb.size;   // unfinished tail
}

// qout_override / n_tiles / rows4: the K*B particle launch of CVPO's E-step (forward only) reuses the Q-net kernel
static int sac_q_launch(fsrl_ctx* c, SacState* s, const float* params, const float* X, int mode, float cr, float cc,
                        float* statp, int B, float* qout_override = nullptr, int n_tiles = -1, int rows4 = -1) {
    FbArgs a{};
    if (n_tiles < 0) n_tiles = s->n_tiles;
    const bool r4 = rows4 < 0 ? s->q_rows4 : rows4 != 0;
    a.obs = X; a.rd = nullptr; a.A1 = s->A1; a.A2 = s->A2; a.D1 = s->D1; a.D2 = s->D2; a.DO = s->DO; a.statp = statp;
    a.N = B; a.rows_pad = n_tiles * 16; a.mode = mode; a.net0 = 0; a.cr = cr; a.cc = cc; a.max_action = 1.0f;
    a.tgt = s->Y; a.qout = (mode == FB_MODE_Q_FWD && params == s->PQT) ? s->QT : s->QP; a.qin = s->QP; a.da_out = s->DA;
    if (qout_override) a.qout = qout_override;
    a.act_cols = c->cfg.act_dim; a.pair_shift = s->n_q == 2 ? 0 : 1;
    return dispatch_H(c->cfg.hidden, [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        // 4-row tiles while they still fit the chip in one round (batch <= 256 for the four Q-nets)
        if (r4) hipLaunchKernelGGL((fb_tile_kernel<H, 4>), dim3(4 * n_tiles, s->n_q), dim3(4 * H), 0, c->compute, params, s->mdq, a);
        else hipLaunchKernelGGL((fb_tile_kernel<H, 16>), dim3(n_tiles, s->n_q), dim3(4 * H), 0, c->compute, params, s->mdq, a);
        HIPCHK(hipGetLastError());
        return 0;
    });
}

// weight gradients of `ny` networks of `md` as split-K partials in c->wg_parts (stride = n_dev)
static int sac_wgrad(fsrl_ctx* c, SacState* s, const ModelDesc& md, int ny, const float* X, int n_dev, int B, int* nsplit) {
    FbWgradArgs wa{};
    const size_t H = c->cfg.hidden, rp = (size_t)s->n_tiles * 16;
    for (int y = 0; y < ny; ++y) {
        FbWgradNet& wn = wa.nets[y];
        const size_t nb = (size_t)y * rp;
        wn.w2_ya = s->D2 + nb * H; wn.w2_xa = s->A1 + nb * H; wn.w2_yb = nullptr; wn.w2_xb = nullptr;
        wn.w1_y = s->D1 + nb * H; wn.w3_xa = s->A2 + nb * H; wn.w3_ya = s->DO + nb * FSRL_DOW;
        wn.w3_xb = nullptr; wn.w3_yb = nullptr; wn.b1_src = s->D1 + nb * H; wn.b2_src = s->D2 + nb * H;
        wn.do_src = s->DO + nb * FSRL_DOW; wn.net = y;
    }
    wa.obs = X; wa.rows = (int)rp; wa.N = B;
    return wgrad_launch<false>(c, md, wa, ny, n_dev, nsplit);
}

// Adam with the gradient read as the z-ordered sum of `nparts` split-K partials
// tgt != NULL: the Polyak update of the target copy rides on the same pass (target <- tau * new + (1 - tau) * target)
static void adam_launch(fsrl_ctx* c, const ModelDesc& md, float* P, float* M, float* V, const float* G, int n, float lr,
                        int64_t t, int nparts, int stride, float* tgt = nullptr, float tau = 0.0f) {
    const double b1 = c->cfg.beta1, b2 = c->cfg.beta2;
    const double bc1 = 1.0 - std::pow(b1, (double)t), bc2 = 1.0 - std::pow(b2, (double)t);
    hipLaunchKernelGGL(adam_range_kernel, dim3((n + 255) / 256), dim3(256), 0, c->compute, P, M, V, G, 0, n, 0.0f,
                       (float)(1.0 - b1), c->cfg.beta2, (float)(1.0 - b2), (float)((double)lr / bc1),
                       (float)std::sqrt(bc2), c->cfg.adam_eps, nparts, stride, md, (const float*)nullptr, 0, 0.0f,
                       (float*)nullptr, tgt, tau, (float)(1.0 - (double)tau));
}

extern "C" int fsrl_sac_update(fsrl_ctx* c, int32_t B, const int64_t* indices, const float* eps_target,
                               const float* eps_pi, uint64_t seed, const double* lagrangians, double rescaling,
                               float* stats_out) {
    CHECK_ARG(c, "null ctx");
    SacState* s = sac_of(c);
    if (!s) return fail(FSRL_ESTATE, "fsrl_sac_init first");
    CHECK_ARG(B >= 1, "batch_size must be >= 1");
    if (s->cvpo) return fail(FSRL_ESTATE, "this context runs CVPO: call fsrl_cvpo_update");
    const int64_t stored = fsrl_store_len(c);
    CHECK_ARG(stored > 0, "empty replay store");
    HIPCHK(hipSetDevice(c->device));
    int rc = join_store(c);
    if (rc) return rc;
    rc = sac_alloc_batch(c, s, B);
    if (rc) return rc;
    hipStream_t st = c->compute;
    const int Do = c->cfg.obs_dim, Da = c->cfg.act_dim, ns = s->cfg.n_step;
    CHECK_ARG((indices != nullptr) == (eps_target != nullptr) && (indices != nullptr) == (eps_pi != nullptr),
              "indices, eps_target and eps_pi are given together (caller RNG) or all NULL (library RNG)");
    if (seed) s->key = seed * 0x9E3779B97F4A7C15ull + 0x243F6A8885A308D3ull;
    if (indices) {
        // ---- caller-provided sample (parity mode): index chains on the host, staged through pinned memory
        HIPCHK(hipStreamSynchronize(st));      // pinned staging of the previous update has landed
        for (int b = 0; b < B; ++b) {
            const int64_t idx = indices[b];
            CHECK_ARG(idx >= 0 && idx < c->maxsize, "index %lld out of range", (long long)idx);
            s->h_idx[b] = (int)idx;
            int64_t cur = idx;
            for (int n = 0; n < ns; ++n) {      // indices[n] = buffer.next(indices[n-1])
                if (n > 0) cur = store_next(c, cur);
                s->h_chain[(size_t)n * B + b] = (int)cur;
                s->h_end[(size_t)n * B + b] = store_end_flag(c, cur) ? 1 : 0;
            }
        }
        float* he_t = s->h_eps; float* he_p = s->h_eps + (size_t)B * Da;
        memcpy(he_t, eps_target, (size_t)B * Da * 4);
        memcpy(he_p, eps_pi, (size_t)B * Da * 4);
        HIPCHK(hipMemcpyAsync(s->d_idx, s->h_idx, (size_t)B * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->d_chain, s->h_chain, (size_t)B * ns * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->d_end, s->h_end, (size_t)B * ns, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->eps_t, he_t, (size_t)B * Da * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->eps_p, he_p, (size_t)B * Da * 4, hipMemcpyHostToDevice, st));
    } else {
        // ---- library RNG: everything on the device, nothing to wait for
        if (s->book_version != c->store_version) {     // the store changed since the last upload
            HIPCHK(hipStreamSynchronize(st));           // h_book may still be in flight
            for (int e = 0; e < c->cfg.env_num; ++e) {
                const EnvBook& eb = c->env[(size_t)e];
                s->h_book[e] = SacBook{(int)eb.size, (int)eb.index, (int)eb.last_index, 0};
            }
            HIPCHK(hipMemcpyAsync(s->d_book, s->h_book, (size_t)c->cfg.env_num * sizeof(SacBook), hipMemcpyHostToDevice, st));
            s->book_version = c->store_version;
        }
        SacSampleArgs sa{};
        sa.book = s->d_book; sa.flags = c->st.flags; sa.idx = s->d_idx; sa.chain = s->d_chain; sa.endbits = s->d_end;
        sa.eps_t = s->eps_t; sa.eps_p = s->eps_p; sa.env_num = c->cfg.env_num; sa.sub_size = (int)c->sub_size; sa.B = B;
        sa.n_step = ns; sa.Da = Da; sa.stored = (unsigned long long)stored; sa.key = s->key;
        sa.counter = (unsigned long long)s->n_updates;
        hipLaunchKernelGGL(sac_sample_kernel, dim3((B + 255) / 256), dim3(256), 0, st, sa);
        HIPCHK(hipGetLastError());
    }
    s->last_B = B;
    const float lam = (s->cfg.use_lagrangian && lagrangians) ? (float)lagrangians[0] : 0.0f;
    const float resc = (float)rescaling;
    // ---- gather
    SacGatherArgs ga{};
    ga.st = c->st; ga.idx = s->d_idx; ga.term = s->d_chain + (size_t)(ns - 1) * B; ga.XQ = s->XQ; ga.OBS = s->OBS;
    ga.OBSN = s->OBSN; ga.XN = s->XN; ga.XP = s->XP; ga.B = B; ga.Do = Do; ga.Da = Da;
    hipLaunchKernelGGL(sac_gather_kernel, dim3(std::min(1024, (B * (Do + Da) + 255) / 256)), dim3(256), 0, st, ga);
    HIPCHK(hipGetLastError());
    // ---- target: a', log pi' at s_{t+n}; target Q-nets; float64 n-step return
    auto actor_launch = [&](const float* obs, const float* eps, float* X, float* lp, int mode, const float* PAx) {
        SacActorArgs aa{};
        aa.deterministic = s->ddpg ? 1 : 0; aa.max_action = c->cfg.max_action;
        aa.obs = obs; aa.eps = eps; aa.X = X; aa.lp_out = lp; aa.DA = s->DA; aa.QP = s->QP; aa.sc = s->sc; aa.A1 = s->A1; aa.A2 = s->A2;
        aa.cr = -resc; aa.cc = s->cfg.use_lagrangian ? resc * lam : 0.0f;
        aa.D1 = s->D1; aa.D2 = s->D2; aa.DO = s->DO; aa.statp = s->stpi; aa.B = B; aa.mode = mode; aa.rescale = resc;
        aa.auto_alpha = s->cfg.auto_alpha; aa.alpha_fixed = s->cfg.alpha;
        return dispatch_H(c->cfg.hidden, [&](auto hc) {
            constexpr int H = decltype(hc)::value;
            if (s->a_rows4) hipLaunchKernelGGL((sac_actor_tile_kernel<H, 4>), dim3(4 * s->n_tiles), dim3(4 * H), 0, st, PAx, s->mda, aa);
            else hipLaunchKernelGGL((sac_actor_tile_kernel<H, 16>), dim3(s->n_tiles), dim3(4 * H), 0, st, PAx, s->mda, aa);
            HIPCHK(hipGetLastError());
            return 0;
        });
    };
    rc = actor_launch(s->OBSN, s->eps_t, s->XN, s->LPN, SAC_A_FWD, s->ddpg ? s->PAT : s->PA);   // DDPG: target actor
    if (rc) return rc;
    rc = sac_q_launch(c, s, s->PQT, s->XN, FB_MODE_Q_FWD, 0.f, 0.f, s->stq, B);
    if (rc) return rc;
    SacNstepArgs na{};
    na.QT = s->QT; na.lpn = s->LPN; na.chain = s->d_chain; na.endbits = s->d_end; na.rew = c->st.rew; na.cost = c->st.cost;
    na.flags = c->st.flags; na.sc = s->sc; na.Y = s->Y; na.B = B; na.n_step = ns; na.gamma = c->cfg.gamma;
    na.auto_alpha = s->cfg.auto_alpha; na.alpha_fixed = s->cfg.alpha; na.single = s->ddpg ? 1 : 0;
    hipLaunchKernelGGL(sac_nstep_kernel, dim3((B + 255) / 256), dim3(256), 0, st, na);
    HIPCHK(hipGetLastError());
    // ---- critic step (all four Q-nets, one Adam)
    rc = sac_q_launch(c, s, s->PQ, s->XQ, FB_MODE_Q_TRAIN, 0.f, 0.f, s->stq, B);
    if (rc) return rc;
    int nsplit = 1;
    rc = sac_wgrad(c, s, s->mdq, s->n_q, s->XQ, s->nq_dev, B, &nsplit);
    if (rc) return rc;
    s->t_critic += 1;
    // sync_weight (sac_lag.py:132-134) is folded into this pass: the critics do not change again within the update
    adam_launch(c, s->mdq, s->PQ, s->MQ, s->VQ, c->wg_parts, s->nq_dev, s->cfg.critic_lr, s->t_critic, nsplit, s->nq_dev,
                s->PQT, s->cfg.tau);
    // ---- actor step: a ~ pi(s), Q(s, a) with the UPDATED critics, dL/da, actor backward
    rc = actor_launch(s->OBS, s->eps_p, s->XP, s->LP, SAC_A_FWD, s->PA);
    if (rc) return rc;
    rc = sac_q_launch(c, s, s->PQ, s->XP, FB_MODE_Q_DIN, 0.f, 0.f, s->stdin_, B);   // Q values + unit-seed dQ/da
    if (rc) return rc;
    rc = actor_launch(s->OBS, s->eps_p, s->XP, s->LP, SAC_A_BWD, s->PA);
    if (rc) return rc;
    rc = sac_wgrad(c, s, s->mda, 1, s->OBS, s->na_dev, B, &nsplit);
    if (rc) return rc;
    s->t_actor += 1;
    // DDPG-Lag: actor_old <- tau * actor + (1 - tau) * actor_old in the same pass (ddpg_lag.py:120-123)
    adam_launch(c, s->mda, s->PA, s->MA, s->VA, c->wg_parts, s->na_dev, s->cfg.actor_lr, s->t_actor, nsplit, s->na_dev,
                s->ddpg ? s->PAT : nullptr, s->cfg.tau);
    // ---- alpha step + logged stats
    SacFinalArgs fa{};
    float* stats_row = s->d_stats + (size_t)(s->n_updates % SAC_RING) * s->nstats;
    fa.statp_q = s->stq; fa.statp_pi = s->stpi; fa.sc = s->sc; fa.stats = stats_row;
    fa.n_tiles_q = s->q_rows4 ? 4 * s->n_tiles : s->n_tiles; fa.n_tiles_pi = s->a_rows4 ? 4 * s->n_tiles : s->n_tiles; fa.B = B; fa.rescale = resc; fa.lam = lam; fa.target_entropy = s->cfg.target_entropy;
    fa.alpha_lr = s->cfg.alpha_lr; fa.beta1 = c->cfg.beta1; fa.beta2 = c->cfg.beta2; fa.adam_eps = c->cfg.adam_eps;
    fa.alpha_fixed = s->cfg.alpha; fa.auto_alpha = s->cfg.auto_alpha; fa.use_lagrangian = s->cfg.use_lagrangian;
    fa.n_q = s->n_q;
    hipLaunchKernelGGL(sac_finalize_kernel, dim3(1), dim3(64), 0, st, fa);
    HIPCHK(hipGetLastError());
    s->n_updates += 1;
    if (stats_out) {                           // synchronous: this update's row (and mark it drained)
        HIPCHK(hipMemcpyAsync(stats_out, stats_row, FSRL_SAC_NSTATS_K * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        s->n_drained = s->n_updates;
    }
    return 0;
}

// ============================================================================== CVPO (cvpo.py:71-430)
extern "C" int fsrl_cvpo_init(fsrl_ctx* c, const fsrl_cvpo_config* cfg) {
    CHECK_ARG(c && cfg, "null argument");
    CHECK_ARG(c->cfg.algo == FSRL_ALGO_SAC_LAG, "CVPO runs on a replay context (FSRL_ALGO_SAC_LAG)");
    CHECK_ARG(c->cfg.act_dim <= 8, "the actor head has at most 16 outputs (act_dim <= 8)");
    CHECK_ARG(c->cfg.obs_dim + c->cfg.act_dim <= FSRL_MAX_OBS, "obs_dim + act_dim too large");
    CHECK_ARG(cfg->n_step >= 1 && cfg->n_step <= 8, "n_step must be in [1, 8]");
    CHECK_ARG(cfg->tau >= 0.0f && cfg->tau <= 1.0f, "tau should be in [0, 1]");
    CHECK_ARG(cfg->sample_act_num >= 1 && cfg->sample_act_num <= 64, "sample_act_num must be in [1, 64]");
    CHECK_ARG(cfg->estep_iter_num >= 1 && cfg->mstep_iter_num >= 1, "estep_iter_num and mstep_iter_num must be >= 1");
    HIPCHK(hipSetDevice(c->device));
    if (c->sac) sac_free(c);
    SacState* s = new SacState();
    c->sac = s;
    s->cvpo = true; s->ccfg = *cfg; s->nstats = FSRL_CVPO_NSTATS_K;
    s->n_q = cfg->double_critic ? 4 : 2;
    s->cfg.actor_lr = cfg->actor_lr; s->cfg.critic_lr = cfg->critic_lr; s->cfg.tau = cfg->tau; s->cfg.n_step = cfg->n_step;
    s->cfg.auto_alpha = 0; s->cfg.alpha = 0.0f; s->cfg.use_lagrangian = 0;      // no entropy term, no PID multiplier
    sac_layout(c, s);
    const size_t HH = (size_t)c->cfg.hidden * c->cfg.hidden;
    const size_t ab = ((size_t)s->na_dev + HH) * 4, qb = ((size_t)s->nq_dev + 4 * HH) * 4;
    for (float** p : {&s->PA, &s->MA, &s->VA, &s->PAT}) { HIPCHK(hipMalloc(p, ab)); HIPCHK(hipMemsetAsync(*p, 0, ab, c->compute)); }
    for (float** p : {&s->PQ, &s->PQT, &s->MQ, &s->VQ}) { HIPCHK(hipMalloc(p, qb)); HIPCHK(hipMemsetAsync(*p, 0, qb, c->compute)); }
    HIPCHK(hipMalloc(&s->sc, sizeof(SacScalars)));
    HIPCHK(hipMemsetAsync(s->sc, 0, sizeof(SacScalars), c->compute));
    HIPCHK(hipMalloc(&s->csc, sizeof(CvpoScalars)));
    HIPCHK(hipMemsetAsync(s->csc, 0, sizeof(CvpoScalars), c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    const float eta0 = 1.0f;                                                     // estep_dual = [1, 0]  (cvpo.py:150-152)
    HIPCHK(hipMemcpy(&s->csc->eta, &eta0, 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&s->d_stats, (size_t)SAC_RING * s->nstats * 4));
    HIPCHK(hipMalloc(&s->d_book, (size_t)c->cfg.env_num * sizeof(SacBook)));
    HIPCHK(hipHostMalloc(&s->h_book, (size_t)c->cfg.env_num * sizeof(SacBook)));
    return 0;
}

static SacState* cvpo_of(fsrl_ctx* c) { SacState* s = c ? sac_of(c) : nullptr; return (s && s->cvpo) ? s : nullptr; }

extern "C" int fsrl_cvpo_pre_update(fsrl_ctx* c) {
    SacState* s = cvpo_of(c);
    if (!s) return fail(FSRL_ESTATE, "fsrl_cvpo_init first");
    HIPCHK(hipSetDevice(c->device));
    // mdual .. dual_std are contiguous: multipliers, Adam moments, step count, clipped copies
    const size_t off = offsetof(CvpoScalars, mdual), end = offsetof(CvpoScalars, estep_loss);
    HIPCHK(hipMemsetAsync((char*)s->csc + off, 0, end - off, c->compute));
    return 0;
}

extern "C" int fsrl_cvpo_post_update(fsrl_ctx* c) {
    SacState* s = cvpo_of(c);
    if (!s) return fail(FSRL_ESTATE, "fsrl_cvpo_init first");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(s->PAT, s->PA, ((size_t)s->na_dev + (size_t)c->cfg.hidden * c->cfg.hidden) * 4,
                          hipMemcpyDeviceToDevice, c->compute));
    return 0;
}

extern "C" int fsrl_cvpo_set_thres(fsrl_ctx* c, double qc_thres) {
    SacState* s = cvpo_of(c);
    if (!s) return fail(FSRL_ESTATE, "fsrl_cvpo_init first");
    s->ccfg.qc_thres = qc_thres;
    return 0;
}

extern "C" int fsrl_cvpo_duals_get(fsrl_ctx* c, float* out4) {
    CHECK_ARG(c && out4, "null argument");
    SacState* s = cvpo_of(c);
    if (!s) return fail(FSRL_ESTATE, "fsrl_cvpo_init first");
    HIPCHK(hipSetDevice(c->device));
    CvpoScalars h;
    HIPCHK(hipMemcpyAsync(&h, s->csc, sizeof(h), hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    out4[0] = h.eta; out4[1] = h.lam; out4[2] = h.mdual[0]; out4[3] = h.mdual[1];
    return 0;
}

extern "C" int fsrl_cvpo_last_particles(fsrl_ctx* c, float* eps_particles, int64_t n) {
    CHECK_ARG(c && eps_particles, "null argument");
    SacState* s = cvpo_of(c);
    if (!s) return fail(FSRL_ESTATE, "fsrl_cvpo_init first");
    CHECK_ARG(s->last_B > 0 && n == (int64_t)s->ccfg.sample_act_num * s->last_B * c->cfg.act_dim,
              "expected K * batch_size * act_dim floats of the last update (batch_size %d)", s->last_B);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(eps_particles, s->eps_k, (size_t)n * 4, hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    return 0;
}

extern "C" int fsrl_cvpo_update(fsrl_ctx* c, int32_t B, const int64_t* indices, const float* eps_target,
                                const float* eps_particles, uint64_t seed, float* stats_out) {
    CHECK_ARG(c, "null ctx");
    SacState* s = cvpo_of(c);
    if (!s) return fail(FSRL_ESTATE, "fsrl_cvpo_init first");
    CHECK_ARG(B >= 1, "batch_size must be >= 1");
    const int64_t stored = fsrl_store_len(c);
    CHECK_ARG(stored > 0, "empty replay store");
    CHECK_ARG((indices != nullptr) == (eps_target != nullptr) && (indices != nullptr) == (eps_particles != nullptr),
              "indices, eps_target and eps_particles are given together (caller RNG) or all NULL (library RNG)");
    HIPCHK(hipSetDevice(c->device));
    int rc = join_store(c);
    if (rc) return rc;
    rc = sac_alloc_batch(c, s, B);
    if (rc) return rc;
    hipStream_t st = c->compute;
    const fsrl_cvpo_config& cc = s->ccfg;
    const int Do = c->cfg.obs_dim, Da = c->cfg.act_dim, ns = cc.n_step, K = cc.sample_act_num;
    const size_t nk = (size_t)K * B * Da;
    if (seed) s->key = seed * 0x9E3779B97F4A7C15ull + 0x243F6A8885A308D3ull;
    if (indices) {
        HIPCHK(hipStreamSynchronize(st));      // pinned staging of the previous update has landed
        for (int b = 0; b < B; ++b) {
            const int64_t idx = indices[b];
            CHECK_ARG(idx >= 0 && idx < c->maxsize, "index %lld out of range", (long long)idx);
            s->h_idx[b] = (int)idx;
            int64_t cur = idx;
            for (int n = 0; n < ns; ++n) {
                if (n > 0) cur = store_next(c, cur);
                s->h_chain[(size_t)n * B + b] = (int)cur;
                s->h_end[(size_t)n * B + b] = store_end_flag(c, cur) ? 1 : 0;
            }
        }
        memcpy(s->h_eps, eps_target, (size_t)B * Da * 4);
        memcpy(s->h_epsk, eps_particles, nk * 4);
        HIPCHK(hipMemcpyAsync(s->d_idx, s->h_idx, (size_t)B * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->d_chain, s->h_chain, (size_t)B * ns * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->d_end, s->h_end, (size_t)B * ns, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->eps_t, s->h_eps, (size_t)B * Da * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->eps_k, s->h_epsk, nk * 4, hipMemcpyHostToDevice, st));
    } else {
        if (s->book_version != c->store_version) {
            HIPCHK(hipStreamSynchronize(st));
            for (int e = 0; e < c->cfg.env_num; ++e) {
                const EnvBook& eb = c->env[(size_t)e];
                s->h_book[e] = SacBook{(int)eb.size, (int)eb.index, (int)eb.last_index, 0};
            }
            HIPCHK(hipMemcpyAsync(s->d_book, s->h_book, (size_t)c->cfg.env_num * sizeof(SacBook), hipMemcpyHostToDevice, st));
            s->book_version = c->store_version;
        }
        SacSampleArgs sa{};
        sa.book = s->d_book; sa.flags = c->st.flags; sa.idx = s->d_idx; sa.chain = s->d_chain; sa.endbits = s->d_end;
        sa.eps_t = s->eps_t; sa.eps_p = s->eps_p; sa.env_num = c->cfg.env_num; sa.sub_size = (int)c->sub_size; sa.B = B;
        sa.n_step = ns; sa.Da = Da; sa.stored = (unsigned long long)stored; sa.key = s->key;
        sa.counter = (unsigned long long)s->n_updates;
        sa.eps_k = s->eps_k; sa.K = K;
        hipLaunchKernelGGL(sac_sample_kernel, dim3((B * K + 255) / 256), dim3(256), 0, st, sa);
        HIPCHK(hipGetLastError());
    }
    s->last_B = B;
    // ---- gather
    SacGatherArgs ga{};
    ga.st = c->st; ga.idx = s->d_idx; ga.term = s->d_chain + (size_t)(ns - 1) * B; ga.XQ = s->XQ; ga.OBS = s->OBS;
    ga.OBSN = s->OBSN; ga.XN = s->XN; ga.XP = s->XP; ga.B = B; ga.Do = Do; ga.Da = Da;
    hipLaunchKernelGGL(sac_gather_kernel, dim3(std::min(1024, (B * (Do + Da) + 255) / 256)), dim3(256), 0, st, ga);
    HIPCHK(hipGetLastError());
    auto actor_launch = [&](int mode, const float* PAx, const float* obs, const float* eps, float* X) {
        CvpoActorArgs aa{};
        aa.obs = obs; aa.eps = eps; aa.X = X; aa.mu_old = s->MU_OLD; aa.std_old = s->STD_OLD; aa.W = s->Wk; aa.XK = s->XK;
        aa.sc = s->csc; aa.A1 = s->A1; aa.A2 = s->A2; aa.D1 = s->D1; aa.D2 = s->D2; aa.DO = s->DO; aa.statp = s->stpi;
        aa.B = B; aa.K = K; aa.mode = mode; aa.max_action = c->cfg.max_action;
        return dispatch_H(c->cfg.hidden, [&](auto hc) {
            constexpr int H = decltype(hc)::value;
            if (s->a_rows4) hipLaunchKernelGGL((cvpo_actor_tile_kernel<H, 4>), dim3(4 * s->n_tiles), dim3(4 * H), 0, st, PAx, s->mda, aa);
            else hipLaunchKernelGGL((cvpo_actor_tile_kernel<H, 16>), dim3(s->n_tiles), dim3(4 * H), 0, st, PAx, s->mda, aa);
            HIPCHK(hipGetLastError());
            return 0;
        });
    };
    // ---- n-step target: a' ~ actor(s_{t+n}), critics_old.predict, float64 return      (cvpo.py:206-222)
    rc = actor_launch(CVPO_A_TARGET, s->PA, s->OBSN, s->eps_t, s->XN);
    if (rc) return rc;
    rc = sac_q_launch(c, s, s->PQT, s->XN, FB_MODE_Q_FWD, 0.f, 0.f, s->stq, B);
    if (rc) return rc;
    SacNstepArgs na{};
    na.QT = s->QT; na.lpn = s->LPN; na.chain = s->d_chain; na.endbits = s->d_end; na.rew = c->st.rew; na.cost = c->st.cost;
    na.flags = c->st.flags; na.sc = s->sc; na.Y = s->Y; na.B = B; na.n_step = ns; na.gamma = c->cfg.gamma;
    na.auto_alpha = 0; na.alpha_fixed = 0.0f; na.single = s->n_q == 2 ? 1 : 0;    // LPN stays zero: no entropy term
    hipLaunchKernelGGL(sac_nstep_kernel, dim3((B + 255) / 256), dim3(256), 0, st, na);
    HIPCHK(hipGetLastError());
    // ---- critic step                                                                (cvpo.py:248-276)
    rc = sac_q_launch(c, s, s->PQ, s->XQ, FB_MODE_Q_TRAIN, 0.f, 0.f, s->stq, B);
    if (rc) return rc;
    int nsplit = 1;
    rc = sac_wgrad(c, s, s->mdq, s->n_q, s->XQ, s->nq_dev, B, &nsplit);
    if (rc) return rc;
    s->t_critic += 1;
    // sync_weight (cvpo.py:202-204) rides on the same pass: the critics do not change again within the update
    adam_launch(c, s->mdq, s->PQ, s->MQ, s->VQ, c->wg_parts, s->nq_dev, cc.critic_lr, s->t_critic, nsplit, s->nq_dev, s->PQT, cc.tau);
    // ---- E-step: K particles of actor_old through the UPDATED critics                (cvpo.py:319-371)
    rc = actor_launch(CVPO_A_PARTICLES, s->PAT, s->OBS, s->eps_k, s->XK);
    if (rc) return rc;
    rc = sac_q_launch(c, s, s->PQ, s->XK, FB_MODE_Q_FWD, 0.f, 0.f, s->stqk, K * B, s->QK, s->n_tiles_k, s->k_rows4 ? 1 : 0);
    if (rc) return rc;
    CvpoEstepArgs ea{};
    ea.QK = s->QK; ea.q0 = s->q0; ea.q1 = s->q1; ea.W = s->Wk; ea.sc = s->csc; ea.B = B; ea.K = K; ea.n_q = s->n_q;
    ea.iters = cc.estep_iter_num; ea.kl = cc.estep_kl; ea.thres = (float)cc.qc_thres; ea.lr = cc.estep_dual_lr;
    ea.dual_max = cc.estep_dual_max; ea.beta1 = c->cfg.beta1; ea.beta2 = c->cfg.beta2; ea.adam_eps = c->cfg.adam_eps;
    hipLaunchKernelGGL(cvpo_estep_kernel, dim3(1), dim3(1024), 0, st, ea);
    HIPCHK(hipGetLastError());
    // ---- M-step                                                                     (cvpo.py:378-417)
    const int n_tiles_pi = s->a_rows4 ? 4 * s->n_tiles : s->n_tiles;
    for (int it = 0; it < cc.mstep_iter_num; ++it) {
        rc = actor_launch(CVPO_A_MFWD, s->PA, s->OBS, nullptr, nullptr);
        if (rc) return rc;
        CvpoMdualArgs ma{};
        ma.statp = s->stpi; ma.n_tiles = n_tiles_pi; ma.B = B; ma.K = K; ma.sc = s->csc; ma.kl_mu_eps = cc.mstep_kl_mu;
        ma.kl_std_eps = cc.mstep_kl_std; ma.dual_max = cc.mstep_dual_max; ma.lr = cc.mstep_dual_lr; ma.beta1 = c->cfg.beta1;
        ma.beta2 = c->cfg.beta2; ma.adam_eps = c->cfg.adam_eps; ma.log_it = it == 0;
        hipLaunchKernelGGL(cvpo_mdual_kernel, dim3(1), dim3(64), 0, st, ma);
        HIPCHK(hipGetLastError());
        rc = actor_launch(CVPO_A_MBWD, s->PA, s->OBS, nullptr, nullptr);
        if (rc) return rc;
        rc = sac_wgrad(c, s, s->mda, 1, s->OBS, s->na_dev, B, &nsplit);
        if (rc) return rc;
        s->t_actor += 1;
        adam_launch(c, s->mda, s->PA, s->MA, s->VA, c->wg_parts, s->na_dev, cc.actor_lr, s->t_actor, nsplit, s->na_dev);
    }
    // ---- logged stats, then Polyak of the critics                                    (cvpo.py:202-204, 422-430)
    CvpoFinalArgs fa{};
    float* stats_row = s->d_stats + (size_t)(s->n_updates % SAC_RING) * s->nstats;
    fa.statp_q = s->stq; fa.Y = s->Y; fa.sc = s->csc; fa.stats = stats_row;
    fa.n_tiles_q = s->q_rows4 ? 4 * s->n_tiles : s->n_tiles; fa.n_q = s->n_q; fa.B = B; fa.thres = (float)cc.qc_thres;
    hipLaunchKernelGGL(cvpo_finalize_kernel, dim3(1), dim3(64), 0, st, fa);
    HIPCHK(hipGetLastError());
    s->n_updates += 1;
    if (stats_out) {
        HIPCHK(hipMemcpyAsync(stats_out, stats_row, (size_t)s->nstats * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        s->n_drained = s->n_updates;
    }
    return 0;
}

// Rows of logged statistics of the updates issued with stats_out == NULL since the last drain
// (oldest first; at most SAC_RING = 4096 are kept).  Returns the number of rows written, < 0 on error.
extern "C" int64_t fsrl_sac_stats_drain(fsrl_ctx* c, float* out, int64_t max_rows) {
    CHECK_ARG(c && out && max_rows >= 0, "bad argument");
    SacState* s = sac_of(c);
    if (!s) return fail(FSRL_ESTATE, "fsrl_sac_init first");
    HIPCHK(hipSetDevice(c->device));
    int64_t first = std::max(s->n_drained, s->n_updates - SAC_RING);
    int64_t n = std::min(s->n_updates - first, max_rows);
    for (int64_t i = 0; i < n;) {              // at most two contiguous pieces of the ring
        const int64_t slot = (first + i) % SAC_RING;
        const int64_t run = std::min(n - i, (int64_t)SAC_RING - slot);
        HIPCHK(hipMemcpyAsync(out + i * s->nstats, s->d_stats + slot * s->nstats,
                              (size_t)run * s->nstats * 4, hipMemcpyDeviceToHost, c->compute));
        i += run;
    }
    HIPCHK(hipStreamSynchronize(c->compute));
    s->n_drained = first + n;
    return n;
}

// The sample the last fsrl_sac_update used (either RNG mode): store indices and both N(0,1) blocks.
extern "C" int fsrl_sac_last_sample(fsrl_ctx* c, int64_t* indices, float* eps_target, float* eps_pi, int32_t B) {
    CHECK_ARG(c && indices && eps_target && eps_pi, "null argument");
    SacState* s = sac_of(c);
    if (!s) return fail(FSRL_ESTATE, "fsrl_sac_init first");
    CHECK_ARG(B == s->last_B && B > 0, "last update used batch_size %d", s->last_B);
    HIPCHK(hipSetDevice(c->device));
    std::vector<int> idx((size_t)B);
    const size_t eb = (size_t)B * c->cfg.act_dim * 4;
    HIPCHK(hipMemcpyAsync(idx.data(), s->d_idx, (size_t)B * 4, hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipMemcpyAsync(eps_target, s->eps_t, eb, hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipMemcpyAsync(eps_pi, s->eps_p, eb, hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    for (int b = 0; b < B; ++b) indices[b] = idx[(size_t)b];
    return 0;
}

// replay-context halves of actor_eval_launch / actor_eval_finish: mlp_infer_kernel writes the raw head outputs
// [mu | log sigma] (2*Da per row; DDPG-Lag: the mean head in the first Da) straight into pinned host memory
static int sac_actor_launch(fsrl_ctx* c, const float* h_obs, float* h_raw, int k) {
    SacState* s = sac_of(c);
    if (!s) return fail(FSRL_ESTATE, "fsrl_sac_init / fsrl_cvpo_init first");
    InferArgs ia{};
    ia.obs = h_obs; ia.obs_next = h_obs; ia.N = k; ia.C = 0; ia.max_action = 1.0f; ia.raw_out = h_raw;
    ia.raw_cols = 2 * c->cfg.act_dim; ia.done = c->h_done; ia.seq = c->actor_seq;
    return dispatch_H(c->cfg.hidden, [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        hipLaunchKernelGGL(mlp_infer_kernel<H>, dim3((k + 15) / 16, 1), dim3(4 * H), 0, c->compute, s->PA, s->mda, ia);
        HIPCHK(hipGetLastError());
        return 0;
    });
}
static void sac_actor_finish(fsrl_ctx* c, const float* raw, int k, float* mu_out, float* sigma_out) {
    SacState* s = sac_of(c);
    const int Da = c->cfg.act_dim;
    for (int r = 0; r < k; ++r)
        for (int d = 0; d < Da; ++d) {
            if (s->ddpg) {     // deterministic actor: the action itself, and the exploration-noise std
                mu_out[(size_t)r * Da + d] = c->cfg.max_action * std::tanh(raw[(size_t)r * 2 * Da + d]);
                if (sigma_out) sigma_out[(size_t)r * Da + d] = s->cfg.exploration_sigma;
                continue;
            }
            mu_out[(size_t)r * Da + d] = s->cvpo ? c->cfg.max_action * std::tanh(raw[(size_t)r * 2 * Da + d])
                                                 : raw[(size_t)r * 2 * Da + d];
            if (sigma_out) {
                const float l = std::min(std::max(raw[(size_t)r * 2 * Da + Da + d], -20.0f), 2.0f);
                sigma_out[(size_t)r * Da + d] = std::exp(l);
            }
        }
}

extern "C" int fsrl_sac_actor_forward(fsrl_ctx* c, const float* obs, int32_t k, float* mu_out, float* sigma_out) {
    CHECK_ARG(c && obs && mu_out && sigma_out, "null argument");
    if (!sac_of(c)) return fail(FSRL_ESTATE, "fsrl_sac_init first");
    if (k <= 0) return 0;
    HIPCHK(hipSetDevice(c->device));
    int rc = actor_eval_launch(c, obs, k, true);
    if (rc) return rc;
    return actor_eval_finish(c, mu_out, sigma_out);
}

static bool sac_squashes(fsrl_ctx* c) { SacState* s = sac_of(c); return s && !s->ddpg && !s->cvpo; }
