// kernels_fb.hpp -- full-batch kernels of the trust-region updates (CPO, TRPO-Lagrangian):
// surrogate / KL / critic-regression gradients, line-search evaluation, and the exact
// Hessian-vector product of the mean KL by an analytic R-op through the tanh-mean MLP.
//
// Reference: fsrl/policy/cpo.py:147-162 (critics_loss), :177-182 (_MVP), :234-254 (objective,
// cost surrogate, kl and their flat gradients), :306-333 (line-search evaluations);
// fsrl/policy/trpo_lag.py:148-171, :189-213, :234-239, :253-259.
//
// Same tile decomposition as the PPO step kernel (one workgroup = one 16-row tile of one
// network, 4*H threads, MFMA 16x16x4 fp32), but rows are the whole batch in store order, the
// grid is ceil(N/16) tiles, and the weight-gradient kernel loops over all N rows.
#pragma once
#include "kernels_mlp.hpp"
#include "kernels_sample.hpp"

#define FB_MODE_VF 0     // critics: d/dtheta mean((ret - V)^2)
#define FB_MODE_SUR 1    // actor: d/dtheta mean((cr*A_r + cc*A_c) * ratio)
#define FB_MODE_KL 2     // actor: d/dtheta mean KL(N(mu_old, sigma_old) || N(mu, sigma))
#define FB_MODE_EVAL 3   // actor: statistics only (line search), nothing stored
#define FB_MODE_Q_TRAIN 4 // SAC Q-nets: d/dtheta mean((Q - y_i)^2), i = net >> 1 ; also writes Q
#define FB_MODE_Q_FWD 5   // SAC Q-nets: forward only, writes Q
#define FB_MODE_Q_DIN 6   // SAC Q-nets: backward to the action input with min-routing coefficients
#define FB_MODE_FOCOPS 7  // actor: d/dtheta mean((KL(new||old) - cr * ratio * (A_r - cc * A_c)) * [KL <= eta])  (focops.py:179-203)
#define FB_NSTAT 8

// ---- SAC / DDPG-Lag n-step regression target of one batch row and one metric (0 = reward, 1 = cost), float64 in the
//      reference's order (base_policy.py:453-512 + nstep_return :543-567; sac_lag.py:136-145 for the entropy term).  Called by
//      the Q-networks' training tile (FB_MODE_Q_TRAIN with FbArgs::ns_on) and by the stand-alone sac_nstep_kernel.
// ---- scalars that live on the device between updates
struct SacScalars {
    float alpha, log_alpha;         // temperature
    float m, v;                     // Adam moments of log_alpha
    int t;                          // Adam step count of log_alpha
    int pad;
};

struct SacNstepArgs {
    const float* QT;        // [4][B] target-net Q(s', a')
    const float* lpn;       // [B] log pi(a'|s')
    const int* chain;       // [n_step][B] index chain (host: buffer.next)
    const uint8_t* endbits; // [n_step][B] end_flag (done | unfinished) at each chain element
    const double* rew; const double* cost; const uint8_t* flags;   // store columns
    const SacScalars* sc;
    float* Y;               // [2][B]
    int B, n_step;
    double gamma;
    int auto_alpha; float alpha_fixed;
    int single;             // DDPG-Lag: one target critic per metric, no entropy term (ddpg_lag.py:125-131)
};
__device__ __forceinline__ float sac_nstep_target(const SacNstepArgs& a, const int b, const int metric) {
#pragma clang fp contract(off)
    const float alpha = a.auto_alpha ? a.sc->alpha : a.alpha_fixed;
    const double* __restrict__ m = metric == 0 ? a.rew : a.cost;
    double gpow = 1.0;
    int gammas = a.n_step;
    double ret = 0.0;
    for (int n = a.n_step - 1; n >= 0; --n) {
        const int now = a.chain[(size_t)n * a.B + b];
        if (a.endbits[(size_t)n * a.B + b]) { gammas = n + 1; ret = 0.0; }
        const double t = a.gamma * ret;
        ret = m[now] + t;
    }
    for (int i = 0; i < gammas; ++i) gpow = gpow * a.gamma;          // gamma_buffer[gammas]
    const int terminal = a.chain[(size_t)(a.n_step - 1) * a.B + b];
    const bool term = (a.flags[terminal] & 1) != 0;
    const float lp = a.single ? 0.0f : alpha * a.lpn[b];
    float tq = a.single ? a.QT[(size_t)metric * a.B + b]
                        : fminf(a.QT[(size_t)(2 * metric) * a.B + b], a.QT[(size_t)(2 * metric + 1) * a.B + b]) - lp;
    if (term) tq = 0.0f;
    const double prod = (double)tq * gpow;
    return (float)(prod + ret);
}

struct FbArgs {
    const float* obs;     // [N][Do]  batch, store order
    const float* rd;      // [N][FSRL_RD] act | logp_old | adv_n | ret | mean_old | std_old
    float* A1; float* A2; float* D1; float* D2; float* DO;   // [nets][n_rows_pad][...]
    float* statp;         // [n_tiles][nets][FB_NSTAT]
    int N, rows_pad;      // rows_pad = n_tiles*16 (stride of the side buffers per net)
    int mode, net0;       // first network handled (0 = actor, 1 = first critic)
    float cr, cc;         // surrogate coefficients (FB_MODE_SUR); Q_DIN: cr = dL/dQr scale, cc = dL/dQc scale
    float max_action;
    // SAC Q-net modes
    const float* tgt;     // [2][N] regression targets y_i           (Q_TRAIN)
    float* qout;          // [nets][N] Q values written              (Q_TRAIN, Q_FWD)
    const float* qin;     // [nets][N] Q values of all nets          (Q_DIN: min routing)
    float* da_out;        // [nets][N][act_cols] dL/da contributions (Q_DIN)
    int act_cols;         // number of action columns at the end of x
    float eta;            // FOCOPS: rows whose KL(new||old) exceeds eta drop out of the loss
    int pair_shift;       // Q_TRAIN target of net n is tgt[n >> pair_shift]: 1 = double critics (SAC), 0 = single (DDPG)
    int ns_on;            // Q_TRAIN: the regression targets are computed HERE (sac_nstep_target) instead of read from tgt: one
    SacNstepArgs ns;      //   launch less per update; the same float64 operations, so the same bits
};

// ---- the actor's loss head of the trust-region modes (SUR, KL, EVAL): ONE definition with floating-point contraction off, called
// by fb_tile_body (every tile height) and by the co-resident kernel (kernels_fbco.hpp), so that the kernel plans agree bit for bit
// whatever the compiler would fuse in one inlining context and not in another (r5: `cr * ar + cc * ac` came out as an fma in one
// instantiation only).  Thread = (row, action dim d) of a 16-lane row group; dout_mu / dout_sig are this thread's entries of the
// row's dL/dout | dL/dsigma_param (0 in EVAL mode), st[0..5] the row statistics (valid rows only; every d holds the same).
__device__ __forceinline__ void fb_tr_actor_head(const float x, const float sigma_param, const float act_d, const float mean_old_d,
                                                 const float std_old_d, const float lpo, const float ar, const float ac, const int d,
                                                 const int Da, const int lane, const bool valid, const int mode, const float cr,
                                                 const float cc, const float max_action, const float invN, const int unbounded,
                                                 float& dout_mu, float& dout_sig, float (&st)[FB_NSTAT]) {
#pragma clang fp contract(off)
    float th = 0.f, var = 1.f, df = 0.f, lp = 0.f, klp = 0.f, dmu = 0.f, so2 = 0.f;
    float hs = max_action;
    if (d < Da) {
        th = tanhf(x);
        const float sig = expf(sigma_param);
        var = sig * sig;
        const float mu = max_action * th;
        df = act_d - mu;
        dmu = mu - mean_old_d;
        if (unbounded) {
            df = act_d - x;
            dmu = x - mean_old_d;
            th = 0.0f; hs = 1.0f;
        }
        lp = -(df * df) / (2.0f * var) - logf(sig) - LOG_SQRT_2PI;
        const float so = std_old_d;                      // KL(old || new), torch.distributions.kl._kl_normal_normal
        so2 = so * so;
        const float var_ratio = (so / sig) * (so / sig);
        const float t1 = (dmu / sig) * (dmu / sig);
        klp = 0.5f * (var_ratio + t1 - 1.0f - logf(var_ratio));
    }
    float logp = 0.0f, klrow = 0.0f;
    for (int dd = 0; dd < Da; ++dd) {
        logp += __shfl(lp, (lane & 48) + dd, 64);
        klrow += __shfl(klp, (lane & 48) + dd, 64);
    }
    const float ratio = expf(logp - lpo);
    dout_mu = 0.0f; dout_sig = 0.0f;
    if (valid && d < Da) {
        if (mode == FB_MODE_SUR) {
            const float dL_dlogp = (cr * ar + cc * ac) * ratio * invN;
            dout_mu = dL_dlogp * (df / var) * hs * (1.0f - th * th);
            dout_sig = dL_dlogp * (df * df / var - 1.0f);
        } else if (mode == FB_MODE_KL) {
            dout_mu = (dmu / var) * invN * hs * (1.0f - th * th);
            dout_sig = (1.0f - (so2 + dmu * dmu) / var) * invN;
        }
    }
    if (valid) {
        st[0] = ratio * ar; st[1] = ratio * ac; st[2] = klrow; st[3] = lpo - logp;
        st[4] = ar; st[5] = ac;
    }
}

// Activation backward of one tile given sm.dout: spills relu(z1), relu(z2), dz2, dout, dz1 for the
// weight-gradient kernel.  wb = this wave's column slice of W2 (lane (li,q): W2[16jc+4q+s][16w+li]).
// din (kernel-uniform; FB_MODE_Q_DIN, whose launch has no weight-gradient launch behind it): nothing is spilled -- 16 MB of
// stores per SAC update that nobody reads -- and dz1 is left in sm.d2 for the input gradient instead; the other callers skip that
// LDS pass and its barrier.
template <int H, int R = 16>
__device__ __forceinline__ void tile_backward(TileSmem<H, tile_rows(R)>& sm, const NetOff no, const float (&wb)[H / 16][4],
                                              float* __restrict__ A1, float* __restrict__ A2,
                                              float* __restrict__ D1, float* __restrict__ D2,
                                              float* __restrict__ DOb, const int tid, const bool din) {
    constexpr int LD = TileSmem<H>::LD;
    constexpr int NT = TileGeom<H>::NT;
    constexpr int H4 = H / 4;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
    if (!din) {
        for (int e = tid; e < R * H4; e += NT) {
            const int i = e / H4, c4 = e - i * H4;
            store4_fb(&A1[(size_t)i * H + 4 * c4], *reinterpret_cast<const f32x4*>(&sm.h1[i * LD + 4 * c4]));
            store4_fb(&A2[(size_t)i * H + 4 * c4], *reinterpret_cast<const f32x4*>(&sm.h2[i * LD + 4 * c4]));
        }
    }
    for (int t = tid; t < (R / 4) * H; t += NT) {   // dz2 = (dout @ W3) * relu'(z2); one trip up to 16 rows, two for 32
        const int k = t % H, rg = t / H;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        for (int o = 0; o < no.out; ++o) {
            const float w = sm.w3[o * H + k];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = fmaf(sm.dout[(4 * rg + e) * FSRL_DOW + o], w, g[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * rg + e;
            sm.d2[i * LD + k] = (sm.h2[i * LD + k] > 0.0f) ? g[e] : 0.0f;
        }
    }
    __syncthreads();
    if (!din) {
        for (int e = tid; e < R * H4; e += NT) {
            const int i = e / H4, c4 = e - i * H4;
            store4_fb(&D2[(size_t)i * H + 4 * c4], *reinterpret_cast<const f32x4*>(&sm.d2[i * LD + 4 * c4]));
        }
        for (int e = tid; e < R * FSRL_DOW; e += NT) DOb[e] = sm.dout[e];
    }
    // dz1 = (dz2 @ W2) * relu'(z1)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int col = wave * 16 + li;
    if constexpr (R == 4) {
        const float* arow = &sm.d2[(lane & 3) * LD + 4 * q];
#pragma unroll
        for (int jc = 0; jc < H / 16; ++jc) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * jc);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma_4x4x1(av[s], wb[jc][s], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {            // add the four k-classes
            acc[r] += __shfl_xor(acc[r], 16, 64);
            acc[r] += __shfl_xor(acc[r], 32, 64);
        }
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = (sm.h1[r * LD + col] > 0.0f) ? acc[r] : 0.0f;
            if (q == 0 && !din) D1[(size_t)r * H + col] = v[r];
        }
        if (din) {
            __syncthreads();          // every wave is done reading dz2 from sm.d2
            if (q == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sm.d2[r * LD + col] = v[r];   // dz1, for input gradients
            }
        }
    } else {
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};                  // rows 16..31 of a 32-row tile: same fragment, second pass
        const float* arow = &sm.d2[li * LD + 4 * q];
#pragma unroll
        for (int jc = 0; jc < H / 16; ++jc) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * jc);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma_16x16x4(av[s], wb[jc][s], acc);
            if constexpr (R == 32) {
                const f32x4 av2 = *reinterpret_cast<const f32x4*>(arow + 16 * LD + 16 * jc);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc2 = mfma_16x16x4(av2[s], wb[jc][s], acc2);
            }
        }
        float v[4], v2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            v[r] = (sm.h1[i * LD + col] > 0.0f) ? acc[r] : 0.0f;
            if (!din) D1[(size_t)i * H + col] = v[r];
            if constexpr (R == 32) {
                v2[r] = (sm.h1[(16 + i) * LD + col] > 0.0f) ? acc2[r] : 0.0f;
                if (!din) D1[(size_t)(16 + i) * H + col] = v2[r];
            }
        }
        if (din) {
            __syncthreads();          // every wave is done reading dz2 from sm.d2
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sm.d2[(4 * q + r) * LD + col] = v[r];   // dz1, for input gradients
                if constexpr (R == 32) sm.d2[(16 + 4 * q + r) * LD + col] = v2[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// R = rows per tile (16, or 4 on v_mfma_f32_4x4x1 when 16-row tiles would leave most CUs idle)
// One tile of R rows starting at row0.  stat_tile: the tile's slot in a.statp (a 32-row tile fills stat_tile and stat_tile + 1).
template <int H, int R>
__device__ __forceinline__ void fb_tile_body(TileSmem<H, tile_rows(R)>& sm, const float* __restrict__ P, const ModelDesc& md,
                                             const FbArgs& a, const int row0, const int stat_tile, const int y, const int ny) {
    constexpr int ROWS = tile_rows(R);
    constexpr int LD = TileSmem<H>::LD;
    constexpr int NT = TileGeom<H>::NT;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
    const int net = a.net0 + y;
    const NetOff no = md.net[net];
    const int Do = md.Do, Da = md.Da;
    const int n_valid = max(0, min(R, a.N - row0));
    const float invN = 1.0f / (float)a.N;

    TileStage<H, ROWS> stg;
    stg.issue(P, no, Do, Da, a.obs + (size_t)row0 * Do, a.rd ? a.rd + (size_t)row0 * FSRL_RD : nullptr, n_valid, tid);
    if (a.mode == FB_MODE_Q_TRAIN && a.ns_on && tid < n_valid) {     // its loads travel underneath the forward pass; read by the head
        const float yv = sac_nstep_target(a.ns, row0 + tid, net >> a.pair_shift);
        sm.st[tid] = yv;
        // the first network of every target group also leaves the target where the stand-alone launch would (CVPO logs its mean)
        if ((net & ((1 << a.pair_shift) - 1)) == 0) a.ns.Y[(size_t)(net >> a.pair_shift) * a.N + row0 + tid] = yv;
    }
    FwdW2Frag<H> wf;
    wf.load(P + no.W2f, wave, lane);
    for (int e = tid; e < ROWS * FSRL_DOW; e += NT) sm.dout[e] = 0.0f;
    stg.commit(sm, no, Do, tid);
    __syncthreads();
#ifdef FSRL_PROBES
#define FBT_PROBE(k) (a.mode >= FB_MODE_Q_TRAIN && a.eta == (float)(k))      /* Q launches of the replay agents: FbArgs::eta is free there */
#else
#define FBT_PROBE(k) false
#endif
    if (FBT_PROBE(1)) return;                                        // behind the prologue
    tile_forward<H, R>(sm, P, no, Do, tid, wf);
    if (FBT_PROBE(2)) return;                                        // behind the forward pass
    const bool backward = (a.mode != FB_MODE_EVAL && a.mode != FB_MODE_Q_FWD);
    const bool qmode = a.mode >= FB_MODE_Q_TRAIN && a.mode <= FB_MODE_Q_DIN;

    float wb[H / 16][4];
    if (backward) {
        const float* __restrict__ W2c = P + no.W2 + wave * 16 + li;
#pragma unroll
        for (int jc = 0; jc < H / 16; ++jc) {
#pragma unroll
            for (int s = 0; s < 4; ++s) wb[jc][s] = W2c[(size_t)(16 * jc + 4 * q + s) * H];
        }
    }

    // ---- head: thread (row i = tid>>4, dim d = tid&15)
    static_assert(16 * R <= NT, "one head thread per (row, dim): a 32-row tile needs H >= 128");
    if (tid < 16 * R) {
        const int i = tid >> 4, d = tid & 15;
        const bool valid = i < n_valid;
        const float* rd = &sm.rd[i * FSRL_RD];
        float st[FB_NSTAT];
#pragma unroll
        for (int k = 0; k < FB_NSTAT; ++k) st[k] = 0.0f;
        if (net == 0 && !qmode && a.mode != FB_MODE_FOCOPS) {
            // the trust-region modes: the shared, contraction-free head (also the co-resident kernel's)
            float dmu_ = 0.0f, dsg_ = 0.0f;
            fb_tr_actor_head(d < Da ? sm.out[i * FSRL_MAX_ACT + d] : 0.0f, d < Da ? sm.sig[d] : 0.0f, rd[d], rd[FSRL_RD_MEAN + d],
                             rd[FSRL_RD_STD + d], rd[FSRL_RD_LOGP], rd[FSRL_RD_ADV], rd[FSRL_RD_ADV + 1], d, Da, lane, valid, a.mode,
                             a.cr, a.cc, a.max_action, invN, md.unbounded, dmu_, dsg_, st);
            if (valid && d < Da && (a.mode == FB_MODE_SUR || a.mode == FB_MODE_KL)) {
                sm.dout[i * FSRL_DOW + d] = dmu_;
                sm.dout[i * FSRL_DOW + 16 + d] = dsg_;
            }
        } else if (net == 0 && !qmode) {
            float th = 0.f, var = 1.f, df = 0.f, lp = 0.f, klp = 0.f, dmu = 0.f, so2 = 0.f;
            float hs = a.max_action;       // d mu / d head = hs * (1 - th * th); an unbounded head: th = 0, hs = 1
            if (d < Da) {
                const float x = sm.out[i * FSRL_MAX_ACT + d];
                th = tanhf(x);
                const float sig = expf(sm.sig[d]);
                var = sig * sig;
                const float mu = a.max_action * th;
                df = rd[d] - mu;
                dmu = mu - rd[FSRL_RD_MEAN + d];
                if (md.unbounded) {            // ActorProb(unbounded=True): the mean is the head itself (the bounded path's
                    df = rd[d] - x;            // expressions above stay as they were, so its rounding does not move)
                    dmu = x - rd[FSRL_RD_MEAN + d];
                    th = 0.0f; hs = 1.0f;
                }
                lp = -(df * df) / (2.0f * var) - logf(sig) - LOG_SQRT_2PI;
                // KL(old || new), torch.distributions.kl._kl_normal_normal
                const float so = rd[FSRL_RD_STD + d];
                so2 = so * so;
                const float var_ratio = (so / sig) * (so / sig);
                const float t1 = (dmu / sig) * (dmu / sig);
                klp = 0.5f * (var_ratio + t1 - 1.0f - logf(var_ratio));
                if (a.mode == FB_MODE_FOCOPS) {          // KL(new || old): _kl_normal_normal(p = new, q = old)
                    const float vr = (sig / so) * (sig / so);
                    const float t1n = (dmu / so) * (dmu / so);
                    klp = 0.5f * (vr + t1n - 1.0f - logf(vr));
                    so2 = vr;                             // reused below: d KL / d log sigma_new = vr - 1
                    dmu = dmu / (so * so);                // d KL / d mu_new
                }
            }
            float logp = 0.0f, klrow = 0.0f;
            for (int dd = 0; dd < Da; ++dd) {
                logp += __shfl(lp, (lane & 48) + dd, 64);
                klrow += __shfl(klp, (lane & 48) + dd, 64);
            }
            const float lpo = rd[FSRL_RD_LOGP];
            const float ratio = expf(logp - lpo);
            const float ar = rd[FSRL_RD_ADV], ac = rd[FSRL_RD_ADV + 1];
            if (valid && d < Da) {
                if (a.mode == FB_MODE_SUR) {
                    const float dL_dlogp = (a.cr * ar + a.cc * ac) * ratio * invN;
                    sm.dout[i * FSRL_DOW + d] = dL_dlogp * (df / var) * hs * (1.0f - th * th);
                    sm.dout[i * FSRL_DOW + 16 + d] = dL_dlogp * (df * df / var - 1.0f);
                } else if (a.mode == FB_MODE_KL) {
                    sm.dout[i * FSRL_DOW + d] = (dmu / var) * invN * hs * (1.0f - th * th);
                    sm.dout[i * FSRL_DOW + 16 + d] = (1.0f - (so2 + dmu * dmu) / var) * invN;
                } else if (a.mode == FB_MODE_FOCOPS) {
                    // loss_row = (KL - cr * ratio * (A_r - cc * A_c)) * mask ; cr = 1/lambda, cc = nu
                    const float mask = (klrow <= a.eta) ? invN : 0.0f;
                    const float dL_dlogp = -a.cr * (ar - a.cc * ac) * ratio;
                    sm.dout[i * FSRL_DOW + d] = (dL_dlogp * (df / var) + dmu) * hs * (1.0f - th * th) * mask;
                    sm.dout[i * FSRL_DOW + 16 + d] = (dL_dlogp * (df * df / var - 1.0f) + (so2 - 1.0f)) * mask;
                }
            }
            if (valid) {
                st[0] = ratio * ar; st[1] = ratio * ac; st[2] = klrow; st[3] = lpo - logp;
                st[4] = ar; st[5] = ac;
                if (a.mode == FB_MODE_FOCOPS)
                    st[0] = (klrow <= a.eta) ? (klrow - a.cr * ratio * (ar - a.cc * ac)) : 0.0f;
            }
        } else if (qmode) {
            const float qv = sm.out[i * FSRL_MAX_ACT];
            const int r = row0 + i;
            if (valid && d == 0) {
                if (a.mode == FB_MODE_Q_TRAIN) {
                    const float td = qv - (a.ns_on ? sm.st[i] : a.tgt[(size_t)(net >> a.pair_shift) * a.N + r]);
                    sm.dout[i * FSRL_DOW] = 2.0f * td * invN;
                    st[0] = td * td;
                    a.qout[(size_t)net * a.N + r] = qv;
                } else if (a.mode == FB_MODE_Q_FWD) {
                    a.qout[(size_t)net * a.N + r] = qv;
                } else {   // Q_DIN: dQ/dx with a unit seed; the consumer (sac_actor_tile_kernel BWD) routes
                           // min(Q1,Q2) with torch's tie rule and applies the loss scale (backward is linear)
                    sm.dout[i * FSRL_DOW] = 1.0f;
                    a.qout[(size_t)net * a.N + r] = qv;
                }
            }
        } else {
            const int c = net - 1;
            const float dd = rd[FSRL_RD_RET + c] - sm.out[i * FSRL_MAX_ACT];
            if (valid) {
                if (d == 0) sm.dout[i * FSRL_DOW] = -2.0f * dd * invN;
                st[0] = dd * dd;
            }
        }
        if (d == 0) {
#pragma unroll
            for (int k = 0; k < FB_NSTAT; ++k) sm.w1[i * FB_NSTAT + k] = st[k];   // w1 is free now
        }
    }
    __syncthreads();
    if constexpr (R <= 16) {
        if (tid < FB_NSTAT) {   // rows summed in ascending order (fixed => deterministic)
            float t = 0.0f;
            for (int i = 0; i < R; ++i) t += sm.w1[i * FB_NSTAT + tid];
            a.statp[((size_t)stat_tile * ny + y) * FB_NSTAT + tid] = t;
        }
    } else {
        // a 32-row tile leaves the partial sums of its two 16-row halves in the slots the 16-row tiles 2t, 2t + 1 would
        // write: the reduction over the tiles (fb_reduce_stats_kernel) then adds the same numbers in the same order
        if (tid < 2 * FB_NSTAT) {
            const int half = tid >> 3, f = tid & 7;
            float t = 0.0f;
            for (int i = 0; i < 16; ++i) t += sm.w1[(16 * half + i) * FB_NSTAT + f];
            a.statp[((size_t)(stat_tile + half) * ny + y) * FB_NSTAT + f] = t;
        }
    }
    if (!backward) return;
    if (FBT_PROBE(3)) return;                                        // behind the loss head and the statistics

    const size_t nb = (size_t)y * a.rows_pad;
    const bool din = a.mode == FB_MODE_Q_DIN;
    // Q_DIN: the action columns of W1 ([H][Dact], 2 .. 16 floats per thread) start their trip now, underneath the backward pass
    // (16-row tiles: the 4-row kernel at 256 wide sits at its 128-register cap and fetches them behind the backward pass as before)
    constexpr bool PRE = (R == 16);
    constexpr int NWA = (H * FSRL_MAX_ACT + NT - 1) / NT;
    float wav[NWA];
    const int Dact = a.act_cols, Dobs = Do - Dact;
    if (din && PRE) {
        const unsigned magic = div_magic(Dact);        // e / Dact by multiply-high: exact for e < 2^16
#pragma unroll
        for (int u = 0; u < NWA; ++u) {
            const int e = min(tid + u * NT, H * Dact - 1);
            const int j = div_by_magic((unsigned)e, magic), kk = e - j * Dact;
            wav[u] = P[no.W1 + (size_t)j * Do + Dobs + kk];
        }
    }
    tile_backward<H, R>(sm, no, wb, a.A1 + (nb + row0) * H, a.A2 + (nb + row0) * H, a.D1 + (nb + row0) * H,
                     a.D2 + (nb + row0) * H, a.DO + (nb + row0) * FSRL_DOW, tid, din);
    if (FBT_PROBE(4)) return;                                        // behind the activation backward
    if (din) {
        // input gradient w.r.t. the action columns of x = concat(obs, act):
        //   da[i][k] = sum_j dz1[i][j] * W1[j][Do_obs + k]       (dz1 left in sm.d2 by tile_backward)
        // the action columns of W1 go to LDS (sm.h2 is free: its last readers sit before tile_backward's first barrier), then
        // thread (row i, column kk, j-phase jp) sums 1/8 of the j range
        float* __restrict__ wact = sm.h2;
        if constexpr (PRE) {
#pragma unroll
            for (int u = 0; u < NWA; ++u) {
                const int e = tid + u * NT;
                if (e < H * Dact) wact[e] = wav[u];
            }
        } else {
            for (int e = tid; e < H * Dact; e += NT) {
                const int j = e / Dact, kk = e - j * Dact;
                wact[e] = P[no.W1 + (size_t)j * Do + Dobs + kk];
            }
        }
        __syncthreads();
        for (int e0 = 0; e0 < R * Dact * 8; e0 += NT) {
            const int e = e0 + tid;
            const int jp = e & 7, ik = e >> 3;
            const int i = ik / Dact, kk = ik - i * Dact;
            float s_ = 0.0f;
            if (ik < R * Dact)
                for (int j = jp; j < H; j += 8) s_ = fmaf(sm.d2[i * LD + j], wact[j * Dact + kk], s_);
            s_ += __shfl_xor(s_, 1, 64);
            s_ += __shfl_xor(s_, 2, 64);
            s_ += __shfl_xor(s_, 4, 64);
            if (jp == 0 && ik < R * Dact && i < n_valid)
                a.da_out[((size_t)y * a.N + row0 + i) * Dact + kk] = s_;
        }
    }
}

template <int H, int R>
__global__ __launch_bounds__(4 * H) void fb_tile_kernel(const float* __restrict__ P,
                                                       const ModelDesc md, const FbArgs a) {
    __shared__ TileSmem<H, tile_rows(R)> sm;
    fb_tile_body<H, R>(sm, P, md, a, blockIdx.x * R, blockIdx.x, blockIdx.y, gridDim.y);
}

// Mixed-height 1-D grid of the full-batch launches (host: mixed_plan): the first ny * n32 blocks take the 32-row tiles of the
// ny networks (two MFMA passes per weight fragment: half the L2 -> register weight traffic per row), the ny * n16 blocks
// behind them 16-row tiles of the remaining rows, so that the last round of workgroups is a round of the cheap tiles.
// A row's arithmetic does not depend on its tile's height and the per-tile statistics keep the 16-row slots: bit-identical
// to fb_tile_kernel<H, 16>.
template <int H>
__global__ __launch_bounds__(4 * H) void fb_tile_mixed_kernel(const float* __restrict__ P, const ModelDesc md, const FbArgs a,
                                                             const int n32, const int n16, const int ny) {
    __shared__ TileSmem<H, 32> sm;
    int b = blockIdx.x;
    if (b < ny * n32) {
        const int y = b / n32, t = b - y * n32;
        fb_tile_body<H, 32>(sm, P, md, a, 32 * t, 2 * t, y, ny);
    } else {
        b -= ny * n32;
        const int y = b / n16, t = b - y * n16;
        fb_tile_body<H, 16>(*reinterpret_cast<TileSmem<H, 16>*>(&sm), P, md, a, 32 * n32 + 16 * t, 2 * n32 + t, y, ny);
    }
}

// ------------------------------------------------------------------------------------------
// Hessian-vector product of KLbar(theta) = mean_r KL(N(mu_old, sigma_old) || N(mu_theta, sigma_theta))
// along the tangent V (same layout as the actor parameters), activation side.  R{.} denotes the
// directional derivative along V (Pearlmutter).  Forward: z1,h1,z2,h2,out and R{h1},R{h2},R{out};
// head: dout = dKLbar/dout, R{dout};  backward: dz2, R{dz2}, R{dz1}.  The weight-side products
// (R{dW} = R{d}^T a + d^T R{a}) are done by fb_wgrad_kernel with two operand pairs.
struct HvpArgs {
    const float* obs; const float* rd;
    const float* V;          // tangent, parameter layout (device offsets of net 0)
    float* A1; float* RA1; float* A2; float* RA2; float* D2; float* RD2; float* RD1;
    float* DO; float* RDO;   // [rows_pad][FSRL_DOW]
    int N, rows_pad;
    float max_action;
    unsigned long long* ts;  // probe builds: [blocks][16] shader-clock stamps of the phase boundaries (else null)
    int gn;                  // r5: mean_old / std_old of the batch ARE the policy at this theta (TRPO-Lag always: trpo_lag.py:189-190;
                             // CPO until its first accepted step): the KL gradient is identically zero, the product is the
                             // Gauss-Newton one.  The heads then use mu - mean_old = 0 and std_old = sigma EXACTLY (what the reference's
                             // autograd sees: old_dist is a detached copy of the same forward), so dout, dz2 and every term they
                             // multiply are exact zeros; kernels_fbco.hpp and the weight-side launch skip them
};

template <int H>
struct HvpSmem {
    static constexpr int LD = H + 4;
    float xT[FSRL_MAX_OBS * 16];
    float h1[16 * LD], rh1[16 * LD], h2[16 * LD], rh2[16 * LD], d2[16 * LD], rd2[16 * LD];
    float out[16 * FSRL_MAX_ACT], rout[16 * FSRL_MAX_ACT];
    float dout[16 * FSRL_DOW], rdout[16 * FSRL_DOW];
    float rd[16 * FSRL_RD];
};

// acc += A[16 x H](LDS, leading dim LD) @ Wrows^T where lane (li,q) of wave w holds row
// (w*16+li) of W, columns 16*kc + 4q..+3 (the FwdW2Frag layout)
template <int H>
__device__ __forceinline__ f32x4 mma_rows(const float* A, const FwdW2Frag<H>& wf, int li, int q, f32x4 acc) {
    constexpr int LD = H + 4;
    const float* arow = A + li * LD + 4 * q;
#pragma unroll
    for (int kc = 0; kc < H / 16; ++kc) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * kc);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma_16x16x4(av[s], wf.b[kc][s], acc);
    }
    return acc;
}

// acc += A[16 x H](LDS) @ W[:, wave's 16 columns]   (column-slice layout of the backward GEMM)
template <int H>
__device__ __forceinline__ f32x4 mma_cols(const float* A, const float* __restrict__ W, int wave, int li,
                                          int q, f32x4 acc) {
    constexpr int LD = H + 4;
    const float* __restrict__ Wc = W + wave * 16 + li;
    float wb[H / 16][4];
#pragma unroll
    for (int jc = 0; jc < H / 16; ++jc) {
#pragma unroll
        for (int s = 0; s < 4; ++s) wb[jc][s] = Wc[(size_t)(16 * jc + 4 * q + s) * H];
    }
    const float* arow = A + li * LD + 4 * q;
#pragma unroll
    for (int jc = 0; jc < H / 16; ++jc) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * jc);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma_16x16x4(av[s], wb[jc][s], acc);
    }
    return acc;
}

template <int H>
__global__ __launch_bounds__(4 * H) void fb_hvp_tile_kernel(const float* __restrict__ P,
                                                           const ModelDesc md, const HvpArgs a) {
    __shared__ HvpSmem<H> sm;
    constexpr int LD = HvpSmem<H>::LD;
    constexpr int NT = 4 * H;
    constexpr int WAVES = H / 16;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * 16;
    const NetOff no = md.net[0];
    const float* __restrict__ V = a.V;
    const int Do = md.Do, Da = md.Da;
    const int n_valid = min(16, a.N - row0);
    const float invN = 1.0f / (float)a.N;

    for (int e = tid; e < 16 * Do; e += NT) {
        const int i = e / Do, k = e - i * Do;
        sm.xT[k * 16 + i] = (i < n_valid) ? a.obs[(size_t)row0 * Do + e] : 0.0f;
    }
    for (int e = tid; e < 16 * FSRL_RD; e += NT)
        sm.rd[e] = (e / FSRL_RD < n_valid) ? a.rd[(size_t)row0 * FSRL_RD + e] : 0.0f;
    for (int e = tid; e < 16 * FSRL_DOW; e += NT) { sm.dout[e] = 0.0f; sm.rdout[e] = 0.0f; }
    FwdW2Frag<H> wf;
    wf.load(P + no.W2f, wave, lane);
    __syncthreads();

    // ---- layer 1 and its tangent on MFMA: the wave's 16 rows of W1 and of V1 in one load burst
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, racc = {0.f, 0.f, 0.f, 0.f};
        const float* __restrict__ wrow = P + no.W1 + (size_t)(wave * 16 + li) * Do;
        const float* __restrict__ vrow = V + no.W1 + (size_t)(wave * 16 + li) * Do;
        for (int k0 = 0; k0 < Do; k0 += 64) {
            float b[16], vb_[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = k0 + 4 * s + q;
                b[s] = (k < Do) ? wrow[k] : 0.0f;
                vb_[s] = (k < Do) ? vrow[k] : 0.0f;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = k0 + 4 * s + q;
                if (k0 + 4 * s < Do) {
                    const float a_ = (k < Do) ? sm.xT[k * 16 + li] : 0.0f;
                    acc = mfma_16x16x4(a_, b[s], acc);
                    racc = mfma_16x16x4(a_, vb_[s], racc);
                }
            }
        }
        const int j = wave * 16 + li;
        const float b1 = P[no.b1 + j], vb1 = V[no.b1 + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = acc[r] + b1;
            const bool on = z > 0.0f;
            sm.h1[(4 * q + r) * LD + j] = on ? z : 0.0f;
            sm.rh1[(4 * q + r) * LD + j] = on ? racc[r] + vb1 : 0.0f;
        }
    }
    __syncthreads();
    // ---- layer 2:  z2 = W2 h1 + b2 ; R{z2} = W2 R{h1} + V2 h1 + vb2
    {
        f32x4 z = {0, 0, 0, 0}, rz = {0, 0, 0, 0};
        z = mma_rows<H>(sm.h1, wf, li, q, z);
        rz = mma_rows<H>(sm.rh1, wf, li, q, rz);
        wf.load(V + no.W2f, wave, lane);
        rz = mma_rows<H>(sm.h1, wf, li, q, rz);
        const int j = wave * 16 + li;
        const float bias = P[no.b2 + j], vbias = V[no.b2 + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float zz = z[r] + bias;
            const bool on = zz > 0.0f;
            sm.h2[(4 * q + r) * LD + j] = on ? zz : 0.0f;
            sm.rh2[(4 * q + r) * LD + j] = on ? rz[r] + vbias : 0.0f;
        }
    }
    __syncthreads();
    // ---- head pre-activations: out = W3 h2 + b3 ; R{out} = W3 R{h2} + V3 h2 + vb3
    for (int i = wave; i < 16; i += WAVES) {
        for (int o = 0; o < Da; ++o) {
            const float* __restrict__ w3 = P + no.W3 + (size_t)o * H;
            const float* __restrict__ v3 = V + no.W3 + (size_t)o * H;
            float s = 0.0f, rs = 0.0f;
#pragma unroll
            for (int k = lane; k < H; k += 64) {
                const float h = sm.h2[i * LD + k];
                s = fmaf(h, w3[k], s);
                rs = fmaf(sm.rh2[i * LD + k], w3[k], rs);
                rs = fmaf(h, v3[k], rs);
            }
            s = wave_sum(s);
            rs = wave_sum(rs);
            if (lane == 0) {
                sm.out[i * FSRL_MAX_ACT + o] = s + P[no.b3 + o];
                sm.rout[i * FSRL_MAX_ACT + o] = rs + V[no.b3 + o];
            }
        }
    }
    __syncthreads();
    // ---- KL head (per row, per action dim): dout, R{dout}, and the sigma_param rows
    if (tid < 256) {
#pragma clang fp contract(off)       // r5: see fb_tile_body
        const int i = tid >> 4, d = tid & 15;
        if (i < n_valid && d < Da) {
            const float x = sm.out[i * FSRL_MAX_ACT + d];
            const float t = md.unbounded ? 0.0f : tanhf(x);           // unbounded head: mu = x, dmu/dout = 1, second derivative 0
            const float hs = md.unbounded ? 1.0f : a.max_action;
            const float ro = sm.rout[i * FSRL_MAX_ACT + d];
            const float sp = P[no.sigma + d], rls = V[no.sigma + d];   // R{log sigma} = v_sigma
            const float sig = expf(sp), var = sig * sig;
            const float dt = hs * (1.0f - t * t);                       // dmu/dout
            const float rmu = dt * ro;
            const float dmu_b = a.max_action * t - sm.rd[i * FSRL_RD + FSRL_RD_MEAN + d];
            const float dmu = a.gn ? 0.0f : (md.unbounded ? x - sm.rd[i * FSRL_RD + FSRL_RD_MEAN + d] : dmu_b);
            const float so = sm.rd[i * FSRL_RD + FSRL_RD_STD + d], so2 = a.gn ? var : so * so;
            const float gmu = dmu / var;                                // dKL/dmu
            const float rgmu = rmu / var - 2.0f * gmu * rls;
            const float rgls = -2.0f * dmu * rmu / var + 2.0f * (so2 + dmu * dmu) / var * rls;
            const float rdt = hs * (-2.0f * t) * (1.0f - t * t) * ro;   // R{dmu/dout}
            sm.dout[i * FSRL_DOW + d] = invN * gmu * dt;
            sm.rdout[i * FSRL_DOW + d] = invN * (rgmu * dt + gmu * rdt);
            sm.dout[i * FSRL_DOW + 16 + d] = invN * (1.0f - (so2 + dmu * dmu) / var);
            sm.rdout[i * FSRL_DOW + 16 + d] = invN * rgls;
        }
    }
    __syncthreads();
    // ---- dz2 = relu'(z2) (dout W3) ; R{dz2} = relu'(z2) (R{dout} W3 + dout V3)
    {
        const int k = tid % H, rg = tid / H;
        float g[4] = {0, 0, 0, 0}, rg_[4] = {0, 0, 0, 0};
        for (int o = 0; o < Da; ++o) {
            const float w = P[no.W3 + (size_t)o * H + k], v = V[no.W3 + (size_t)o * H + k];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dd = sm.dout[(4 * rg + e) * FSRL_DOW + o];
                g[e] = fmaf(dd, w, g[e]);
                rg_[e] = fmaf(sm.rdout[(4 * rg + e) * FSRL_DOW + o], w, rg_[e]);
                rg_[e] = fmaf(dd, v, rg_[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * rg + e;
            const bool on = sm.h2[i * LD + k] > 0.0f;
            sm.d2[i * LD + k] = on ? g[e] : 0.0f;
            sm.rd2[i * LD + k] = on ? rg_[e] : 0.0f;
        }
    }
    __syncthreads();
    // ---- R{dz1} = relu'(z1) (R{dz2} W2 + dz2 V2)
    {
        f32x4 acc = {0, 0, 0, 0};
        acc = mma_cols<H>(sm.rd2, P + no.W2, wave, li, q, acc);
        acc = mma_cols<H>(sm.d2, V + no.W2, wave, li, q, acc);
        float* __restrict__ RD1 = a.RD1 + (size_t)row0 * H;
        const int col = wave * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            RD1[(size_t)i * H + col] = (sm.h1[i * LD + col] > 0.0f) ? acc[r] : 0.0f;
        }
    }
    // ---- spill the operands of the weight-side products
    {
        constexpr int H4 = H / 4;
        const size_t base = (size_t)row0 * H;
        for (int e = tid; e < 16 * H4; e += NT) {
            const int i = e / H4, c4 = e - i * H4;
            const size_t o = base + (size_t)i * H + 4 * c4;
            const int l = i * LD + 4 * c4;
            *reinterpret_cast<f32x4*>(a.A1 + o) = *reinterpret_cast<const f32x4*>(&sm.h1[l]);
            *reinterpret_cast<f32x4*>(a.RA1 + o) = *reinterpret_cast<const f32x4*>(&sm.rh1[l]);
            *reinterpret_cast<f32x4*>(a.A2 + o) = *reinterpret_cast<const f32x4*>(&sm.h2[l]);
            *reinterpret_cast<f32x4*>(a.RA2 + o) = *reinterpret_cast<const f32x4*>(&sm.rh2[l]);
            *reinterpret_cast<f32x4*>(a.D2 + o) = *reinterpret_cast<const f32x4*>(&sm.d2[l]);
            *reinterpret_cast<f32x4*>(a.RD2 + o) = *reinterpret_cast<const f32x4*>(&sm.rd2[l]);
        }
        for (int e = tid; e < 16 * FSRL_DOW; e += NT) {
            a.DO[(size_t)row0 * FSRL_DOW + e] = sm.dout[e];
            a.RDO[(size_t)row0 * FSRL_DOW + e] = sm.rdout[e];
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same product on 32-row tiles, with the theta-only part cached.  Conjugate gradients call the HVP 22 times per
// CPO repeat at ONE theta on ONE batch (cpo.py:255-268: H^-1 g, then H^-1 b): h1, h2, dout and dz2 are the same in all
// of them.  CACHED = false computes everything and leaves A1 / A2 / D2 / DO in the side buffers (first product at a theta);
// CACHED = true reads h1 / h2 back (2 x 32 KB per tile, MALL-resident: N x 2 x H floats = 41 MB at N = 20 000), skips the
// z GEMMs and the four theta-only spills.  32 rows per tile = two MFMA passes per weight fragment: the four 256 KB
// fragment ingests (W2 and V2, forward and backward order) are paid per 32 rows instead of per 16.  Every output element
// sees the arithmetic of fb_hvp_tile_kernel in the same order (K order inside a row does not depend on the tile height;
// relu'(z) is read off h > 0), so the three kernels agree bit for bit (tests/test_gpu_trust.py).
// LDS: four 32 x (H + 4) slots -- h1 | R{h1}, later dz2 | h2 | R{h2}, later R{dz2} -- 158 KB at H = 256.
template <int H>
struct Hvp32Smem {
    static constexpr int LD = H + 4;
    float s0[32 * LD], s1[32 * LD], s2[32 * LD], s3[32 * LD];
    float xT[64 * 32];                                        // obs tile transposed [k][i], obs_dim <= 64
    float out[32 * FSRL_MAX_ACT], rout[32 * FSRL_MAX_ACT];
    float dout[32 * FSRL_DOW], rdout[32 * FSRL_DOW];
    float mo[32 * 32];                                        // per row: mean_old[16] | std_old[16]
};

// NH 16-row passes over one fragment set: acc[hf] = rows 16 hf .. 16 hf + 15
template <int H, int NH>
__device__ __forceinline__ void mma_rows_n(const float* A, const FwdW2Frag<H>& wf, int li, int q, f32x4 (&acc)[NH]) {
    constexpr int LD = H + 4;
    const float* arow = A + li * LD + 4 * q;
#pragma unroll
    for (int kc = 0; kc < H / 16; ++kc) {
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * hf * LD + 16 * kc);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[hf] = mfma_16x16x4(av[s], wf.b[kc][s], acc[hf]);
        }
    }
}
// BUF (r6, the co-resident kernels): the 64 dword loads of the column burst as buffer loads -- one resource, one lane offset, the row
// offset in an SGPR -- instead of 64 address pairs (see FwdW2Frag::load_buf for the measurements)
template <int H, int NH, bool BUF = false>
__device__ __forceinline__ void mma_cols_n(const float* A, const float* __restrict__ W, int wave, int li, int q,
                                           f32x4 (&acc)[NH]) {
    constexpr int LD = H + 4;
    float wb[H / 16][4];
    if constexpr (BUF) {
        const Wg3Buf bw = wg3_buf_here(W);
        const unsigned lane_off = (unsigned)(wave * 16 + li) + (unsigned)(4 * q) * (unsigned)H;
#pragma unroll
        for (int jc = 0; jc < H / 16; ++jc) {
#pragma unroll
            for (int s = 0; s < 4; ++s) wb[jc][s] = wg3_ld1s(bw, lane_off, (unsigned)(16 * jc + s) * (unsigned)H);
        }
    } else {
        const float* __restrict__ Wc = W + wave * 16 + li;
#pragma unroll
        for (int jc = 0; jc < H / 16; ++jc) {
#pragma unroll
            for (int s = 0; s < 4; ++s) wb[jc][s] = Wc[(size_t)(16 * jc + 4 * q + s) * H];
        }
    }

    const float* arow = A + li * LD + 4 * q;
#pragma unroll
    for (int jc = 0; jc < H / 16; ++jc) {
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * hf * LD + 16 * jc);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[hf] = mfma_16x16x4(av[s], wb[jc][s], acc[hf]);
        }
    }
}

// one tile of 16 NH rows starting at row0
template <int H, bool CACHED, int NH>
__device__ __forceinline__ void hvp_tile_body(Hvp32Smem<H>& sm, const float* __restrict__ P, const ModelDesc& md,
                                              const HvpArgs& a, const int row0) {
    constexpr int LD = Hvp32Smem<H>::LD;
    constexpr int NT = 4 * H;
    constexpr int WAVES = H / 16;
    constexpr int H4 = H / 4;
    constexpr int R = 16 * NH;
    static_assert(NT >= 16 * R, "one head thread per (row, dim)");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
    const NetOff no = md.net[0];
    const float* __restrict__ V = a.V;
    const int Do = md.Do, Da = md.Da;
    const int n_valid = min(R, a.N - row0);
    const float invN = 1.0f / (float)a.N;
    const size_t base = (size_t)row0 * H;
    float* h1 = sm.s0; float* rh1 = sm.s1; float* h2 = sm.s2; float* rh2 = sm.s3;
    float* d2 = sm.s1; float* rd2 = sm.s3;          // later owners of the two tangent slots

    FSRL_TS(a.ts, 0);                                // in-kernel timeline of probe builds: tools/tstamp_hvp.py
    FwdW2Frag<H> wf;
    wf.load(P + no.W2f, wave, lane);
    if constexpr (CACHED) {                          // h1, h2 of this theta, left by the CACHED = false launch
        for (int e = tid; e < R * H4; e += NT) {
            const int i = e / H4, c4 = e - i * H4;
            *reinterpret_cast<f32x4*>(&h1[i * LD + 4 * c4]) = *reinterpret_cast<const f32x4*>(a.A1 + base + (size_t)i * H + 4 * c4);
            *reinterpret_cast<f32x4*>(&h2[i * LD + 4 * c4]) = *reinterpret_cast<const f32x4*>(a.A2 + base + (size_t)i * H + 4 * c4);
        }
    }
    for (int e = tid; e < R * Do; e += NT) {
        const int i = e / Do, k = e - i * Do;
        sm.xT[k * 32 + i] = (i < n_valid) ? a.obs[(size_t)row0 * Do + e] : 0.0f;
    }
    for (int e = tid; e < R * 32; e += NT) {
        const int i = e >> 5, f = e & 31;
        sm.mo[e] = (i < n_valid) ? a.rd[(size_t)(row0 + i) * FSRL_RD + FSRL_RD_MEAN + f] : 0.0f;   // MEAN and STD are adjacent
    }
    for (int e = tid; e < R * FSRL_DOW; e += NT) { sm.dout[e] = 0.0f; sm.rdout[e] = 0.0f; }
    FSRL_TS(a.ts, 1);
    __syncthreads();
    FSRL_TS(a.ts, 2);

    // ---- layer 1 and its tangent on MFMA: the wave's 16 rows of W1 and of V1 in one load burst, NH row halves
    {
        f32x4 acc[NH], racc[NH];
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) acc[hf] = racc[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* __restrict__ wrow = P + no.W1 + (size_t)(wave * 16 + li) * Do;
        const float* __restrict__ vrow = V + no.W1 + (size_t)(wave * 16 + li) * Do;
        for (int k0 = 0; k0 < Do; k0 += 64) {
            float b[16], vb_[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = k0 + 4 * s + q;
                if constexpr (!CACHED) b[s] = (k < Do) ? wrow[k] : 0.0f;
                vb_[s] = (k < Do) ? vrow[k] : 0.0f;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = k0 + 4 * s + q;
                if (k0 + 4 * s < Do) {
#pragma unroll
                    for (int hf = 0; hf < NH; ++hf) {
                        const float a_ = (k < Do) ? sm.xT[k * 32 + 16 * hf + li] : 0.0f;
                        if constexpr (!CACHED) acc[hf] = mfma_16x16x4(a_, b[s], acc[hf]);
                        racc[hf] = mfma_16x16x4(a_, vb_[s], racc[hf]);
                    }
                }
            }
        }
        const int j = wave * 16 + li;
        const float b1 = P[no.b1 + j], vb1 = V[no.b1 + j];
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int l = (16 * hf + 4 * q + r) * LD + j;
                bool on;
                if constexpr (CACHED) on = h1[l] > 0.0f;
                else {
                    const float z = acc[hf][r] + b1;
                    on = z > 0.0f;
                    h1[l] = on ? z : 0.0f;
                }
                rh1[l] = on ? racc[hf][r] + vb1 : 0.0f;
            }
        }
    }
    __syncthreads();
    FSRL_TS(a.ts, 3);
    // ---- layer 2:  z2 = W2 h1 + b2 ; R{z2} = W2 R{h1} + V2 h1 + vb2
    {
        f32x4 z[NH], rz[NH];
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) z[hf] = rz[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!CACHED) mma_rows_n<H, NH>(h1, wf, li, q, z);
        mma_rows_n<H, NH>(rh1, wf, li, q, rz);
        FSRL_TS(a.ts, 4);
        wf.load(V + no.W2f, wave, lane);
        mma_rows_n<H, NH>(h1, wf, li, q, rz);
        FSRL_TS(a.ts, 5);
        const int j = wave * 16 + li;
        const float bias = P[no.b2 + j], vbias = V[no.b2 + j];
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int l = (16 * hf + 4 * q + r) * LD + j;
                bool on;
                if constexpr (CACHED) on = h2[l] > 0.0f;
                else {
                    const float zz = z[hf][r] + bias;
                    on = zz > 0.0f;
                    h2[l] = on ? zz : 0.0f;
                }
                rh2[l] = on ? rz[hf][r] + vbias : 0.0f;
            }
        }
    }
    __syncthreads();
    FSRL_TS(a.ts, 6);
    // R{h1} (and h1 on the first product) leave for the weight-side kernel; slot 1 is free after the next barrier
    for (int e = tid; e < R * H4; e += NT) {
        const int i = e / H4, c4 = e - i * H4;
        const size_t o = base + (size_t)i * H + 4 * c4;
        const int l = i * LD + 4 * c4;
        store4_fb(a.RA1 + o, *reinterpret_cast<const f32x4*>(&rh1[l]));
        if constexpr (!CACHED) store4_fb(a.A1 + o, *reinterpret_cast<const f32x4*>(&h1[l]));
    }
    // ---- head pre-activations: out = W3 h2 + b3 ; R{out} = W3 R{h2} + V3 h2 + vb3
    for (int i = wave; i < R; i += WAVES) {
        for (int o = 0; o < Da; ++o) {
            const float* __restrict__ w3 = P + no.W3 + (size_t)o * H;
            const float* __restrict__ v3 = V + no.W3 + (size_t)o * H;
            float s = 0.0f, rs = 0.0f;
#pragma unroll
            for (int k = lane; k < H; k += 64) {
                const float h = h2[i * LD + k];
                s = fmaf(h, w3[k], s);
                rs = fmaf(rh2[i * LD + k], w3[k], rs);
                rs = fmaf(h, v3[k], rs);
            }
            s = wave_sum(s);
            rs = wave_sum(rs);
            if (lane == 0) {
                sm.out[i * FSRL_MAX_ACT + o] = s + P[no.b3 + o];
                sm.rout[i * FSRL_MAX_ACT + o] = rs + V[no.b3 + o];
            }
        }
    }
    __syncthreads();
    FSRL_TS(a.ts, 7);
    // ---- KL head (per row, per action dim): dout, R{dout}, and the sigma_param rows
    if (tid < 16 * R) {
#pragma clang fp contract(off)       // r5: see fb_tile_body
        const int i = tid >> 4, d = tid & 15;
        if (i < n_valid && d < Da) {
            const float x = sm.out[i * FSRL_MAX_ACT + d];
            const float t = md.unbounded ? 0.0f : tanhf(x);           // unbounded head: mu = x, dmu/dout = 1, second derivative 0
            const float hs = md.unbounded ? 1.0f : a.max_action;
            const float ro = sm.rout[i * FSRL_MAX_ACT + d];
            const float sp = P[no.sigma + d], rls = V[no.sigma + d];   // R{log sigma} = v_sigma
            const float sig = expf(sp), var = sig * sig;
            const float dt = hs * (1.0f - t * t);                       // dmu/dout
            const float rmu = dt * ro;
            const float dmu_b = a.max_action * t - sm.mo[i * 32 + d];
            const float dmu = a.gn ? 0.0f : (md.unbounded ? x - sm.mo[i * 32 + d] : dmu_b);
            const float so = sm.mo[i * 32 + 16 + d], so2 = a.gn ? var : so * so;
            const float gmu = dmu / var;                                // dKL/dmu
            const float rgmu = rmu / var - 2.0f * gmu * rls;
            const float rgls = -2.0f * dmu * rmu / var + 2.0f * (so2 + dmu * dmu) / var * rls;
            const float rdt = hs * (-2.0f * t) * (1.0f - t * t) * ro;   // R{dmu/dout}
            sm.dout[i * FSRL_DOW + d] = invN * gmu * dt;
            sm.rdout[i * FSRL_DOW + d] = invN * (rgmu * dt + gmu * rdt);
            sm.dout[i * FSRL_DOW + 16 + d] = invN * (1.0f - (so2 + dmu * dmu) / var);
            sm.rdout[i * FSRL_DOW + 16 + d] = invN * rgls;
        }
    }
    // R{h2} (and h2) leave; slot 3 is free after the barrier
    for (int e = tid; e < R * H4; e += NT) {
        const int i = e / H4, c4 = e - i * H4;
        const size_t o = base + (size_t)i * H + 4 * c4;
        const int l = i * LD + 4 * c4;
        store4_fb(a.RA2 + o, *reinterpret_cast<const f32x4*>(&rh2[l]));
        if constexpr (!CACHED) store4_fb(a.A2 + o, *reinterpret_cast<const f32x4*>(&h2[l]));
    }
    __syncthreads();
    FSRL_TS(a.ts, 8);
    // ---- dz2 = relu'(z2) (dout W3) ; R{dz2} = relu'(z2) (R{dout} W3 + dout V3)      (dz2 -> slot 1, R{dz2} -> slot 3)
    for (int t = tid; t < (R / 4) * H; t += NT) {
        const int k = t % H, rg = t / H;
        float g[4] = {0, 0, 0, 0}, rg_[4] = {0, 0, 0, 0};
        for (int o = 0; o < Da; ++o) {
            const float w = P[no.W3 + (size_t)o * H + k], v = V[no.W3 + (size_t)o * H + k];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dd = sm.dout[(4 * rg + e) * FSRL_DOW + o];
                g[e] = fmaf(dd, w, g[e]);
                rg_[e] = fmaf(sm.rdout[(4 * rg + e) * FSRL_DOW + o], w, rg_[e]);
                rg_[e] = fmaf(dd, v, rg_[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * rg + e;
            const bool on = h2[i * LD + k] > 0.0f;
            d2[i * LD + k] = on ? g[e] : 0.0f;
            rd2[i * LD + k] = on ? rg_[e] : 0.0f;
        }
    }
    __syncthreads();
    FSRL_TS(a.ts, 9);
    // ---- R{dz1} = relu'(z1) (R{dz2} W2 + dz2 V2)
    {
        f32x4 acc[NH];
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) acc[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        mma_cols_n<H, NH>(rd2, P + no.W2, wave, li, q, acc);
        FSRL_TS(a.ts, 10);
        mma_cols_n<H, NH>(d2, V + no.W2, wave, li, q, acc);
        FSRL_TS(a.ts, 11);
        float* __restrict__ RD1 = a.RD1 + base;
        const int col = wave * 16 + li;
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * hf + 4 * q + r;
                RD1[(size_t)i * H + col] = (h1[i * LD + col] > 0.0f) ? acc[hf][r] : 0.0f;
            }
        }
    }
    // ---- the remaining operands of the weight-side products
    for (int e = tid; e < R * H4; e += NT) {
        const int i = e / H4, c4 = e - i * H4;
        const size_t o = base + (size_t)i * H + 4 * c4;
        const int l = i * LD + 4 * c4;
        store4_fb(a.RD2 + o, *reinterpret_cast<const f32x4*>(&rd2[l]));
        if constexpr (!CACHED) store4_fb(a.D2 + o, *reinterpret_cast<const f32x4*>(&d2[l]));
    }
    for (int e = tid; e < R * FSRL_DOW; e += NT) {
        a.RDO[(size_t)row0 * FSRL_DOW + e] = sm.rdout[e];
        if constexpr (!CACHED) a.DO[(size_t)row0 * FSRL_DOW + e] = sm.dout[e];
    }
    FSRL_TS(a.ts, 12);
}

// Mixed-height grid: blocks [0, n32) take 32-row tiles, the rest 16-row tiles behind them (host: hvp_plan): the tile
// counts are chosen so that the LAST round of workgroups is a round of the cheap tiles (N = 20 000 on 256 CUs: 512 x 32
// rows = two full rounds, then 226 x 16 rows, instead of 625 x 32 rows = two rounds and a 44 %-full third).
template <int H, bool CACHED>
__global__ __launch_bounds__(4 * H) void fb_hvp_mixed_kernel(const float* __restrict__ P, const ModelDesc md, const HvpArgs a,
                                                            const int n32) {
    __shared__ Hvp32Smem<H> sm;
    const int b = blockIdx.x;
    if (b < n32) hvp_tile_body<H, CACHED, 2>(sm, P, md, a, 32 * b);
    else hvp_tile_body<H, CACHED, 1>(sm, P, md, a, 32 * n32 + 16 * (b - n32));
}

// ------------------------------------------------------------------------------------------
// Weight-side reduction over ALL rows of the batch, one network per grid.y.  Every product is
//   out[j][k] = sum_r ( Ya[r][j] * Xa[r][k]  +  Yb[r][j] * Xb[r][k] )      (pair b optional)
// grid.x = NT2 (dW2 32x32 tiles) + NA (32-column aux blocks: dW1, dW3^T, db1, db2) + 1 (db3, dsigma)
struct FbWgradNet {
    const float* w2_ya; const float* w2_xa; const float* w2_yb; const float* w2_xb;   // [rows][H]
    const float* w1_y;                           // [rows][H]   (x = observations)
    const float* w3_xa; const float* w3_ya;      // A2-like [rows][H], DO-like [rows][DOW]
    const float* w3_xb; const float* w3_yb;      // optional second pair
    const float* b1_src; const float* b2_src;    // column sums -> db1, db2
    const float* do_src;                         // column sums -> db3 / dsigma
    int net;                                     // which network's slice of `out` is written
};
struct FbWgradArgs {
    FbWgradNet nets[FSRL_MAX_NETS];
    const float* obs;    // [N][Do]
    float* out;          // flat, parameter layout; split z writes its partial at out + z * split_stride
    int rows;            // padded row count (multiple of 16; rows beyond N hold zeros in Y)
    int N;
    int ks_per_split;    // k-steps (4 rows each) per blockIdx.z
    int split_stride;    // floats between the partial gradients of consecutive splits
    int dbg_skip;        // timing experiments: bit0 skip dW2 tiles, bit1 skip aux blocks, bit2 skip the db3 block
    // XCD-aware placement (0 = off: 3-D grid as is).  Else the grid is 1-D, padded to a multiple of 8, and carries
    // remap_total logical blocks in the order (block of a split, network, split): hardware block L runs on XCD L % 8
    // (observed placement, used for speed only), so logical block (L % 8) * (grid / 8) + L / 8 puts CONSECUTIVE logical
    // blocks -- the blocks of one split, which stream the same rows -- behind one L2.
    int remap_total, remap_ny;
    int aux_passes;     // passes over the rows an aux column block needs (1 + the dW1 chunks beyond the first pass): one BLOCK per pass
    // fb_wgrad3_kernel (kernels_wgrad3.hpp) only: the observations re-laid for float4 operand loads ([rows][64 * obs_ko], zero-filled;
    // null = the caller has none and the launch keeps fb_wgrad_kernel), and bit 0 of wg3_flags = XCD-aware block order
    const float* obs_pad; int obs_ko; int wg3_flags;
};

// one pass of an aux block of fb_wgrad_kernel over its rows: NCH 16-column chunks of dW1 (columns k0 ..), and with FIRST the
// dW3 / db1 / db2 outputs of the block's FB_AUX_COLS hidden columns.  Accumulation order per output = row order: the result
// does not depend on NCH.
//
// 64 hidden columns per aux block (round 3, late; 32 before).  The role is bound by its load REQUESTS, not by bytes in flight
// or MFMA time (probe build, CPO at N = 20 000: the aux blocks alone took 93 of the critic steps' 98 us and 68 of the R-op
// product's 82 us while doing 1/16 of the FLOPs; 1, 2 or 4 k-steps of loads in flight made no difference): a lane now loads
// one float4 per operand and k-step instead of a float2, the observation columns a block reads serve twice the outputs, the
// db1 sums reuse the dW1 operand (the same array in every caller: b1_src == w1_y), and a split has 21 blocks instead of 25, so
// the same chip takes more splits with fewer rows each.
#define FB_AUX_COLS 64
#define FB_AUX_SLOT (FB_AUX_COLS * 34)       // [CW x 16 dW1 chunk][CW x 16 dW3^T][CW db1][CW db2] floats per partial slot
template <int H, bool PAIR2, bool FIRST, int NCH>
__device__ __forceinline__ void wgrad_aux_pass(const FbWgradArgs& wa, const FbWgradNet& wn, const NetOff& no, float* red,
                                               float* __restrict__ gout, const int j0, const int k0, const int KS0,
                                               const int KS, const int Do, const int out, const int tid) {
    constexpr int AUXU = 2;                      // k-steps per load burst
    constexpr int CW = FB_AUX_COLS, T = CW / 16; // T = 4 columns per lane: lane (c, q) holds columns j0 + 4c .. + 3 of row q
    constexpr int SL = FB_AUX_SLOT;
    const int lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
    constexpr bool first = FIRST;
    f32x4 ax[NCH][T];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int t = 0; t < T; ++t) ax[ch][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ad[T];
#pragma unroll
    for (int t = 0; t < T; ++t) ad[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, s3 = {0.f, 0.f, 0.f, 0.f};
    // a head with ONE output (every critic / Q-net): dW3 = A2^T dout is a matrix-vector product -- four FMAs per lane and
    // k-step instead of four 16-wide MFMAs that would carry one useful column (a third of this pass's matrix-pipe time)
    const bool vec3 = first && !PAIR2 && out == 1;
    // r6: buffer loads (see wg3_ld4); a null second pair gets pair a's array (never loaded from)
    const Wg3Buf b_y1 = wg3_buf(wn.w1_y), b_obs = wg3_buf(wa.obs), b_xa3 = wg3_buf(wn.w3_xa), b_ya3 = wg3_buf(wn.w3_ya),
                 b_xb3 = wg3_buf(PAIR2 ? wn.w3_xb : wn.w3_xa), b_yb3 = wg3_buf(PAIR2 ? wn.w3_yb : wn.w3_ya), b_b2 = wg3_buf(wn.b2_src);
    for (int sb = KS0 + wave; sb < KS; sb += 16 * AUXU) {
        f32x4 y1[AUXU], xa3[AUXU], xb3[AUXU], b2v[AUXU];
        float bx[AUXU][NCH], bda[AUXU], bdb[AUXU];
#pragma unroll
        for (int u = 0; u < AUXU; ++u) {
            const int s = sb + 16 * u;
            y1[u] = xa3[u] = xb3[u] = b2v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            bda[u] = bdb[u] = 0.f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) bx[u][ch] = 0.f;
            if (s < KS) {
                const unsigned r = (unsigned)(4 * s + q);
                const unsigned rh = r * (unsigned)H + (unsigned)(j0 + T * c);
                y1[u] = wg3_ld4(b_y1, rh);
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch)
                    if (k0 + 16 * ch + c < Do && r < (unsigned)wa.N) bx[u][ch] = wg3_ld1(b_obs, r * (unsigned)Do + (unsigned)(k0 + 16 * ch + c));
                if (first) {
                    xa3[u] = wg3_ld4(b_xa3, rh);
                    bda[u] = wg3_ld1(b_ya3, r * (unsigned)FSRL_DOW + (unsigned)(vec3 ? 0 : c));
                    if constexpr (PAIR2) {
                        xb3[u] = wg3_ld4(b_xb3, rh);
                        bdb[u] = wg3_ld1(b_yb3, r * (unsigned)FSRL_DOW + (unsigned)c);
                    }
                    b2v[u] = wg3_ld4(b_b2, rh);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < AUXU; ++u) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                if (k0 + 16 * ch < Do) {           // block-uniform: chunks past Do cost nothing
#pragma unroll
                    for (int t = 0; t < T; ++t) ax[ch][t] = mfma_16x16x4(y1[u][t], bx[u][ch], ax[ch][t]);
                }
            }
            if (first) {
                if (vec3) {
#pragma unroll
                    for (int t = 0; t < T; ++t) s3[t] = fmaf(xa3[u][t], bda[u], s3[t]);
                } else {
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        ad[t] = mfma_16x16x4(xa3[u][t], bda[u], ad[t]);
                        if constexpr (PAIR2) ad[t] = mfma_16x16x4(xb3[u][t], bdb[u], ad[t]);
                    }
                }
                s1 += y1[u];                       // db1: column sums of the dW1 operand itself (b1_src == w1_y)
                s2 += b2v[u];
            }
        }
    }
    if (first) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            s1[t] += __shfl_xor(s1[t], 16, 64); s1[t] += __shfl_xor(s1[t], 32, 64);
            s2[t] += __shfl_xor(s2[t], 16, 64); s2[t] += __shfl_xor(s2[t], 32, 64);
            s3[t] += __shfl_xor(s3[t], 16, 64); s3[t] += __shfl_xor(s3[t], 32, 64);
        }
    }
    // acc register r of MFMA tile t in lane (c, q) is output (j = j0 + T (4 q + r) + t, k = chunk column c).  Eight partial slots,
    // two rounds: waves 0-7 store, waves 8-15 add into the same slot; the final sum walks the slots in order.
    float* slot = red + (wave & 7) * SL;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (k0 + 16 * ch >= Do) continue;             // block-uniform
        const bool f0 = first && ch == 0;           // dW3 and the bias sums ride with the first chunk
        __syncthreads();
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            if ((wave >> 3) == round) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jl = T * (4 * q + r);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        if (round == 0) {
                            slot[(jl + t) * 16 + c] = ax[ch][t][r];
                            if (f0 && !vec3) slot[CW * 16 + (jl + t) * 16 + c] = ad[t][r];
                        } else {
                            slot[(jl + t) * 16 + c] += ax[ch][t][r];
                            if (f0 && !vec3) slot[CW * 16 + (jl + t) * 16 + c] += ad[t][r];
                        }
                    }
                }
                if (f0 && q == 0) {
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        if (round == 0) { slot[2 * CW * 16 + T * c + t] = s1[t]; slot[2 * CW * 16 + CW + T * c + t] = s2[t]; }
                        else { slot[2 * CW * 16 + T * c + t] += s1[t]; slot[2 * CW * 16 + CW + T * c + t] += s2[t]; }
                        if (vec3) {          // output column 0 of hidden column T c + t; the other 15 slots of the row are never read
                            if (round == 0) slot[CW * 16 + (T * c + t) * 16] = s3[t];
                            else slot[CW * 16 + (T * c + t) * 16] += s3[t];
                        }
                    }
                }
            }
            __syncthreads();
        }
        const int kc0 = k0 + 16 * ch;
        {
            const int jl = tid >> 4, kk = tid & 15;          // 1024 threads = CW x 16 outputs of the dW1 chunk
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[w * SL + tid];
            if (kc0 + kk < Do) gout[no.W1 + (size_t)(j0 + jl) * Do + kc0 + kk] = v;
            if (f0) {
                float v3 = 0.0f;
#pragma unroll
                for (int w = 0; w < 8; ++w) v3 += red[w * SL + CW * 16 + tid];
                if (kk < out) gout[no.W3 + (size_t)kk * H + j0 + jl] = v3;
                if (tid < 2 * CW) {
                    float bsum = 0.0f;
#pragma unroll
                    for (int w = 0; w < 8; ++w) bsum += red[w * SL + 2 * CW * 16 + tid];
                    if (tid < CW) gout[no.b1 + j0 + tid] = bsum;
                    else gout[no.b2 + j0 + tid - CW] = bsum;
                }
            }
        }
        __syncthreads();
    }
}

// Weight-side products over a row range.  grid = (NT2 + NA + 1, ny, nsplit): blockIdx.z owns the rows
// [4*z*ks_per_split, 4*(z+1)*ks_per_split) and writes a PARTIAL gradient; the consumer
// (fb_sum_parts_kernel or adam_range_kernel's nparts) adds the partials in z order, so the result
// does not depend on scheduling.  PAIR2: second operand pair (R-op products of the HVP).
// Rider (kernels_sample.hpp): SgRider appends blocks along x that draw + gather the replay agents' next batch on the CUs this launch
// leaves idle; NoRider (every other caller) compiles to the kernel as it was.
template <int H, bool PAIR2, class Rider = NoRider>
__global__ __launch_bounds__(1024) void fb_wgrad_kernel(const ModelDesc md, const FbWgradArgs wa, const Rider rider = Rider{}) {
    if constexpr (Rider::on) {
        if ((int)blockIdx.x >= rider.x0) {
            sac_sample_gather_block(rider.sa, rider.ga, (int)((blockIdx.z * gridDim.y + blockIdx.y) * rider.nx + blockIdx.x - rider.x0));
            return;
        }
    }
    constexpr int TPD = H / 64;             // dW2 tiles of 64 x 64 outputs
    constexpr int NT2 = TPD * TPD;
    constexpr int NA = H / FB_AUX_COLS;
    constexpr int BU = 1;                   // k-steps per load burst (A/B at N = 20 000: 1 -> CPO 37.9 ms, 2 -> 38.6, 3 -> 39.6, 4 -> 41.5)
    constexpr int SLOT = 64 * 65;           // one 64 x 64 partial tile (+1 column of padding)
    __shared__ float red[(4 * SLOT > 8 * FB_AUX_SLOT) ? 4 * SLOT : 8 * FB_AUX_SLOT];     // tile role: 4 partial tiles; aux role: 8 partial slots
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int rb = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (wa.remap_total) {
        const int L = blockIdx.x, per = gridDim.x >> 3;
        const int Lp = (L & 7) * per + (L >> 3);
        if (Lp >= wa.remap_total) return;
        const int NB = NT2 + NA * wa.aux_passes + 1;
        rb = Lp % NB;
        const int g = Lp / NB;
        by = g % wa.remap_ny; bz = g / wa.remap_ny;
    }
    const FbWgradNet wn = wa.nets[by];
    const NetOff no = md.net[wn.net];
    const int KS0 = bz * wa.ks_per_split;
    const int KS = min(wa.rows >> 2, KS0 + wa.ks_per_split);     // this split's k-step range [KS0, KS)
    float* __restrict__ gout = wa.out + (size_t)bz * wa.split_stride;
    const int c = lane & 15, q = lane >> 4;
    const int Do = md.Do, out = no.out;

#ifdef FSRL_PROBES
    if (wa.dbg_skip && ((rb < NT2 && (wa.dbg_skip & 1)) || (rb >= NT2 && rb < NT2 + NA * wa.aux_passes && (wa.dbg_skip & 2)) ||
                        (rb >= NT2 + NA * wa.aux_passes && (wa.dbg_skip & 4)))) return;
#endif
    if (rb < NT2) {
        // ---- dW2[j][k] += sum_r Ya[r][j] Xa[r][k] (+ Yb Xb): a 64 x 64 tile per block, 16-way split-K over
        // the waves.  Per k-step (4 rows) a lane loads ONE float4 of each operand (256 B contiguous per
        // row and operand) and feeds 16 MFMAs: 8 FLOP per loaded byte (32 x 32 tiles: 4 -- the kernel
        // was L2-bandwidth bound at N = 20 000).  Lane (c, q) holds columns 4c..4c+3 of row q, so
        // acc[t][u][r] is output (j = 4(4q+r)+t, k = 4c+u) of the tile.
        const int tj = rb / TPD, tk = rb % TPD;
        f32x4 acc[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        // Software-pipelined: the operands of the wave's NEXT k-step are requested before the 16 (32) MFMAs of the current one
        // issue, so a wave's own MFMA time hides part of its load latency too (load -> wait -> MFMAs left that to the other
        // three waves of the SIMD).  Same k order per wave, same accumulation order: bit-identical to the loop it replaces.
        const Wg3Buf bya = wg3_buf(wn.w2_ya + tj * 64), bxa = wg3_buf(wn.w2_xa + tk * 64);
        const Wg3Buf byb = wg3_buf(PAIR2 ? wn.w2_yb + tj * 64 : wn.w2_ya), bxb = wg3_buf(PAIR2 ? wn.w2_xb + tk * 64 : wn.w2_xa);
        auto fetch = [&](const int s, f32x4& ya, f32x4& xa, f32x4& yb, f32x4& xb) {
            const unsigned r = (unsigned)(4 * s + q) * (unsigned)H + 4u * (unsigned)c;      // r6: buffer loads (see wg3_ld4)
            ya = wg3_ld4(bya, r);
            xa = wg3_ld4(bxa, r);
            if constexpr (PAIR2) {
                yb = wg3_ld4(byb, r);
                xb = wg3_ld4(bxb, r);
            }
        };
        auto fma16 = [&](const f32x4& ya, const f32x4& xa, const f32x4& yb, const f32x4& xb) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[t][u] = mfma_16x16x4(ya[t], xa[u], acc[t][u]);
                    if constexpr (PAIR2) acc[t][u] = mfma_16x16x4(yb[t], xb[u], acc[t][u]);
                }
        };
        static_assert(BU == 1, "the pipelined loop replaces the burst loop");
        {
            f32x4 ya, xa, yb, xb;
            ya = xa = yb = xb = f32x4{0.f, 0.f, 0.f, 0.f};
            int s0 = KS0 + wave;
            if (s0 < KS) fetch(s0, ya, xa, yb, xb);
            for (; s0 < KS; s0 += 16) {
                f32x4 nya = ya, nxa = xa, nyb = yb, nxb = xb;
                if (s0 + 16 < KS) fetch(s0 + 16, nya, nxa, nyb, nxb);
                fma16(ya, xa, yb, xb);
                ya = nya; xa = nxa; yb = nyb; xb = nxb;
            }
        }
        // four partial-tile slots, four rounds: waves 4k..4k+3 add into slot (wave & 3) in round k
        float* myred = red + (wave & 3) * SLOT;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            if ((wave >> 2) == round) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* p = &myred[(4 * (4 * q + r) + t) * 65 + 4 * c];
                        if (round == 0) { p[0] = acc[t][0][r]; p[1] = acc[t][1][r]; p[2] = acc[t][2][r]; p[3] = acc[t][3][r]; }
                        else { p[0] += acc[t][0][r]; p[1] += acc[t][1][r]; p[2] += acc[t][2][r]; p[3] += acc[t][3][r]; }
                    }
            }
            __syncthreads();
        }
#pragma unroll
        for (int e0 = 0; e0 < 4096; e0 += 1024) {
            const int e = e0 + tid, jl = e >> 6, kl = e & 63;
            const float v = (red[jl * 65 + kl] + red[SLOT + jl * 65 + kl]) + (red[2 * SLOT + jl * 65 + kl] + red[3 * SLOT + jl * 65 + kl]);
            gout[no.W2 + (size_t)(tj * 64 + jl) * H + tk * 64 + kl] = v;
        }
    } else if (rb < NT2 + NA * wa.aux_passes) {
        // dW1 is [H][Do]: the obs columns go through the MFMA 16 at a time, several 16-column chunks per pass over the rows (one
        // read of the dz1 column block serves them all).  The first pass also carries dW3 and the bias sums (more operands in
        // flight), so it takes two chunks (one in the R-op instantiation), the later ones four.  Every pass of a column block is
        // its own workgroup (grid.x = NT2 + NA * passes + 1): with one workgroup walking its rows once per pass, the aux blocks
        // of a Do = 60 network were the critical path of the launch (probe build, CPO critic steps at N = 20 000: aux blocks
        // alone 93 us, tile blocks alone 69 us, together 98 us).
        const int ai = rb - NT2;
        const int j0 = (ai % NA) * FB_AUX_COLS, pass = ai / NA;
        constexpr int NCH0 = PAIR2 ? 1 : 2;          // the R-op instantiation has two more operands in flight in its first pass
        if (pass == 0) wgrad_aux_pass<H, PAIR2, true, NCH0>(wa, wn, no, red, gout, j0, 0, KS0, KS, Do, out, tid);
        else wgrad_aux_pass<H, PAIR2, false, 4>(wa, wn, no, red, gout, j0, 16 * NCH0 + 64 * (pass - 1), KS0, KS, Do, out, tid);
    } else {
        // db3[o] / dsigma[d]: column sums of the dout-like buffer over all rows
        const int col = tid & 31, php = tid >> 5;
        float t = 0.0f;
        const int rend = 4 * KS;
        for (int r0 = 4 * KS0 + php; r0 < rend; r0 += 32 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + 32 * u;
                v[u] = (r < rend) ? wn.do_src[(size_t)r * FSRL_DOW + col] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        red[php * 33 + col] = t;
        __syncthreads();
        if (tid < 32) {
            float tot = 0.0f;
#pragma unroll
            for (int p2 = 0; p2 < 32; ++p2) tot += red[p2 * 33 + tid];
            if (tid < out) gout[no.b3 + tid] = tot;
            if (no.sigma >= 0 && tid >= 16 && tid < 16 + md.Da) gout[no.sigma + tid - 16] = tot;
        }
    }
}

// sum of the split partials of one element (or four adjacent ones) in float64, z ascending: the loads of EIGHT partials are
// issued before the first add (the one-pass weight-gradient kernel leaves 32-64 partials; one dependent load per add took
// cg_pz_kernel from 5 to 18 us).  Same additions in the same order as the plain loop.
__device__ __forceinline__ double sum_parts1(const float* __restrict__ parts, const size_t stride, const int nparts, const size_t i) {
    double acc = 0.0;
    int z = 0;
    for (; z + 8 <= nparts; z += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = parts[(size_t)(z + u) * stride + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    for (; z < nparts; ++z) acc += (double)parts[(size_t)z * stride + i];
    return acc;
}
__device__ __forceinline__ void sum_parts4(const float* __restrict__ parts, const size_t stride, const int nparts, const size_t i4,
                                           double (&a)[4]) {
    a[0] = a[1] = a[2] = a[3] = 0.0;
    int z = 0;
    for (; z + 8 <= nparts; z += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(parts + (size_t)(z + u) * stride + i4);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[0] += (double)v[u][0]; a[1] += (double)v[u][1]; a[2] += (double)v[u][2]; a[3] += (double)v[u][3]; }
    }
    for (; z < nparts; ++z) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(parts + (size_t)z * stride + i4);
        a[0] += (double)v[0]; a[1] += (double)v[1]; a[2] += (double)v[2]; a[3] += (double)v[3];
    }
}

// out[i] = sum_z parts[z * stride + i], z ascending (fixed order), i in [begin, end)
__global__ __launch_bounds__(256) void fb_sum_parts_kernel(float* __restrict__ out, const float* __restrict__ parts,
                                                          int begin, int end, int nparts, int stride,
                                                          float* __restrict__ gsq_part = nullptr) {
    __shared__ float sh[4];
    const int i = begin + blockIdx.x * 256 + threadIdx.x;
    float v = 0.0f;
    if (i < end) {
        // the split-K partials (each an fp32 MFMA chain over >= 256 rows) are combined in float64, z ascending, and rounded
        // once: at N = 20 000 that is up to 24 terms -- the summation-order noise fp32 conjugate gradients amplify
        v = (float)sum_parts1(parts, (size_t)stride, nparts, (size_t)i);
        out[i] = v;
    }
    if (gsq_part) {                          // per-block sum of squares (clip_grad_norm_ of the consumer)
        float q = wave_sum(v * v);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0) gsq_part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    }
}

// ------------------------------------------------------------------------------------------
// sum the per-tile statistics: statp[n_tiles][ny][FB_NSTAT] -> out[ny][FB_NSTAT] (float64)
// r6 (late): `hout` / `done` (pinned host memory) -- every block also writes its sum straight to the host and then its completion word
// `seq` (system-scope release): the host polls the words instead of paying a device-to-host copy launch and a stream synchronisation per
// read-back (tr_stats / tr_dots: ~40 per CPO update, ~25 us each).
__device__ __forceinline__ void rb_publish(double* __restrict__ hout, unsigned* __restrict__ done, const int slot, const double v,
                                           const unsigned seq) {
    hout[slot] = v;
    __threadfence_system();
    __hip_atomic_store(done + slot, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(256) void fb_reduce_stats_kernel(const float* __restrict__ statp, int n_tiles,
                                                             int ny, double* __restrict__ out, double* __restrict__ hout = nullptr,
                                                             unsigned* __restrict__ done = nullptr, unsigned seq = 0u) {
    __shared__ double sh[4];
    const int slot = blockIdx.x;          // (y, field)
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int t = tid; t < n_tiles; t += 256) s += (double)statp[(size_t)t * ny * FB_NSTAT + slot];
    s = wave_sum_d(s);
    if ((tid & 63) == 0) sh[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        const double v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        out[slot] = v;
        if (hout) rb_publish(hout, done, slot, v, seq);
    }
}

// ---- conjugate gradients on the device (cpo.py:184-204, trpo_lag.py:261-283).  Vectors live in the
// actor's device layout (length n = md.net[0].end, inter-tensor padding stays zero); p is the tangent
// vector the HVP kernels read (with its W2 mirror), z arrives in `hz` from fb_sum_parts_kernel.
// One 1024-thread block per step: both dot products and the three axpys, reductions in a fixed order.
struct CgScal { float rs[2]; int done; int iters; };     // rs[it & 1] = r.r entering iteration `it`
#define CG_NB 96                                          // blocks of the CG kernels (256 threads, float4 each, grid-stride)

// The dot products of conjugate gradients (p.z, r.r over ~8e4 entries) accumulate in float64 and are rounded to fp32 once:
// fp32 CG on the damped KL Hessian amplifies summation-order noise (DESIGN.md "conditioning note"), and these sums are its
// cheapest source to remove.  Fixed order throughout (lane tree, waves 0..3, blocks 0..CG_NB-1): deterministic.
__device__ __forceinline__ double cg_block_sum256(double v, double* sh, int tid) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// every block adds the CG_NB partials in the same fixed order (cheap, and no extra launch)
__device__ __forceinline__ float cg_sum_parts(const double* __restrict__ part, double* sh, int tid) {
    double v = (tid < CG_NB) ? part[tid] : 0.0;
    return (float)cg_block_sum256(v, sh, tid);
}

// r = g, p = g (with the W2 mirror), x = 0 ; partial r.r
__global__ __launch_bounds__(256) void cg_init_kernel(const float* __restrict__ g, float* __restrict__ r,
                                                     float* __restrict__ p, float* __restrict__ x,
                                                     CgScal* __restrict__ sc, double* __restrict__ part, int n,
                                                     const ModelDesc md) {
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int i4 = (blockIdx.x * 256 + tid) * 4; i4 < n; i4 += CG_NB * 1024) {
        const f32x4 gi = *reinterpret_cast<const f32x4*>(g + i4);
        *reinterpret_cast<f32x4*>(r + i4) = gi;
        *reinterpret_cast<f32x4*>(p + i4) = gi;
        *reinterpret_cast<f32x4*>(x + i4) = f32x4{0.f, 0.f, 0.f, 0.f};
        const int mi = w2f_mirror_of(md, i4);
        if (mi >= 0) *reinterpret_cast<f32x4*>(p + mi) = gi;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += (double)gi[e] * (double)gi[e];
    }
    const double t = cg_block_sum256(acc, sh, tid);
    if (tid == 0) part[blockIdx.x] = t;
    if (blockIdx.x == 0 && tid == 0) { sc->done = 0; sc->iters = 0; }
}

// z = H p + damping p (in place in hz) ; partial p.z
// parts != null: H p arrives as the split-K partials of fb_wgrad_kernel and is summed here exactly as fb_sum_parts_kernel would
// (float64, z ascending, rounded once): one launch less per conjugate-gradient iteration
__global__ __launch_bounds__(256) void cg_pz_kernel(float* __restrict__ hz, const float* __restrict__ p,
                                                   const CgScal* __restrict__ sc, double* __restrict__ part, int n,
                                                   float damping, const float* __restrict__ parts = nullptr, int nparts = 0,
                                                   int stride = 0) {
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    if (sc->done) return;
    double acc = 0.0;
    for (int i4 = (blockIdx.x * 256 + tid) * 4; i4 < n; i4 += CG_NB * 1024) {
        const f32x4 pi = *reinterpret_cast<const f32x4*>(p + i4);
        f32x4 z;
        if (parts) {
            double a[4];
            sum_parts4(parts, (size_t)stride, nparts, (size_t)i4, a);
            z = f32x4{(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
        } else
            z = *reinterpret_cast<const f32x4*>(hz + i4);
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = z[e] + pi[e] * damping;
        *reinterpret_cast<f32x4*>(hz + i4) = z;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += (double)pi[e] * (double)z[e];
    }
    const double t = cg_block_sum256(acc, sh, tid);
    if (tid == 0) part[blockIdx.x] = t;
}

// alpha = rs_old / p.z ; x += alpha p ; r -= alpha z ; partial r.r       (rs_old = sum of part_rr when it == 0)
__global__ __launch_bounds__(256) void cg_xr_kernel(const float* __restrict__ hz, float* __restrict__ r,
                                                   const float* __restrict__ p, float* __restrict__ x,
                                                   CgScal* __restrict__ sc, const double* __restrict__ part_pz,
                                                   double* __restrict__ part_rr, int n, int it) {
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    if (sc->done) return;
    float rs_old;
    if (it == 0) rs_old = cg_sum_parts(part_rr, sh, tid);         // r.r of cg_init_kernel
    else rs_old = sc->rs[it & 1];
    __syncthreads();
    const float pAp = cg_sum_parts(part_pz, sh, tid);
    __syncthreads();
    const float alpha = rs_old / pAp;
    double acc = 0.0;
    for (int i4 = (blockIdx.x * 256 + tid) * 4; i4 < n; i4 += CG_NB * 1024) {
        const f32x4 pi = *reinterpret_cast<const f32x4*>(p + i4);
        const f32x4 z = *reinterpret_cast<const f32x4*>(hz + i4);
        f32x4 xi = *reinterpret_cast<const f32x4*>(x + i4), ri = *reinterpret_cast<const f32x4*>(r + i4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { xi[e] = xi[e] + alpha * pi[e]; ri[e] = ri[e] - alpha * z[e]; }
        *reinterpret_cast<f32x4*>(x + i4) = xi;
        *reinterpret_cast<f32x4*>(r + i4) = ri;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += (double)ri[e] * (double)ri[e];
    }
    __syncthreads();
    const double t = cg_block_sum256(acc, sh, tid);
    // the new partials must not overwrite part_rr while other blocks still read it at it == 0:
    // they go to the second half of the buffer on even iterations, the first half on odd ones
    if (tid == 0) part_rr[((it & 1) ? 0 : CG_NB) + blockIdx.x] = t;
    if (blockIdx.x == 0 && tid == 0 && it == 0) sc->rs[0] = rs_old;
}

// rs_new = sum partials ; converged -> done ; else p = r + (rs_new / rs_old) p (with the W2 mirror)
__global__ __launch_bounds__(256) void cg_p_kernel(const float* __restrict__ r, float* __restrict__ p,
                                                  CgScal* __restrict__ sc, const double* __restrict__ part_rr, int n,
                                                  int it, float tol, const ModelDesc md) {
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    if (sc->done) return;
    const float rs_new = cg_sum_parts(part_rr + ((it & 1) ? 0 : CG_NB), sh, tid);
    __syncthreads();
    float rs_old = sc->rs[it & 1];
    if (it == 0) rs_old = cg_sum_parts(part_rr, sh, tid);          // sc->rs[0] is written by a sibling launch's block 0 only
    if (rs_new < tol) {                                            // every block takes the same branch
        if (blockIdx.x == 0 && tid == 0) { sc->rs[(it + 1) & 1] = rs_new; sc->iters = it + 1; }
        // `done` is raised by the NEXT launch's view: write it last, nobody in this launch reads it again
        if (blockIdx.x == 0 && tid == 0) { __threadfence(); sc->done = 1; }
        return;
    }
    const float beta = rs_new / rs_old;
    for (int i4 = (blockIdx.x * 256 + tid) * 4; i4 < n; i4 += CG_NB * 1024) {
        const f32x4 ri = *reinterpret_cast<const f32x4*>(r + i4);
        f32x4 pi = *reinterpret_cast<const f32x4*>(p + i4);
#pragma unroll
        for (int e = 0; e < 4; ++e) pi[e] = ri[e] + beta * pi[e];
        *reinterpret_cast<f32x4*>(p + i4) = pi;
        const int mi = w2f_mirror_of(md, i4);
        if (mi >= 0) *reinterpret_cast<f32x4*>(p + mi) = pi;
    }
    if (blockIdx.x == 0 && tid == 0) { sc->rs[(it + 1) & 1] = rs_new; sc->iters = it + 1; }
}

// ---- flat-vector algebra of the trust-region updates on the device (actor layout, length n = md.net[0].end, padding zero).
// Round 2 kept g, b, H^-1 g, H^-1 b, the step direction and the line-search parameters on the host: every HVP outside CG, every
// line-search evaluation and every gradient cost a stream drain, a repack of 8e4 floats and two synchronous copies -- 6 ms of
// a 45 ms CPO update were GPU idle time.  The expressions are the host's (float32 element-wise, no contraction; float64
// accumulation of the dot products, rounded once), so results move only by the order of the float64 sums.
struct TrDotArgs { const float* a[4]; const float* b[4]; int n_pairs; };
// partial sums of up to four dot products: part[pair * CG_NB + block]
__global__ __launch_bounds__(256) void tr_dots_kernel(const TrDotArgs da, double* __restrict__ part, int n,
                                                     double* __restrict__ hpart = nullptr, unsigned* __restrict__ done = nullptr,
                                                     unsigned seq = 0u) {
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    for (int k = 0; k < da.n_pairs; ++k) {
        const float* __restrict__ a = da.a[k];
        const float* __restrict__ b = da.b[k];
        double acc = 0.0;
        for (int i4 = (blockIdx.x * 256 + tid) * 4; i4 < n; i4 += CG_NB * 1024) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(a + i4), y = *reinterpret_cast<const f32x4*>(b + i4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += (double)x[e] * (double)y[e];
        }
        const double t = cg_block_sum256(acc, sh, tid);
        if (tid == 0) {
            part[k * CG_NB + blockIdx.x] = t;
            if (hpart) hpart[k * CG_NB + blockIdx.x] = t;
        }
        __syncthreads();
    }
    if (hpart && tid == 0) {              // the block's partials of every pair are on their way: its completion word behind them
        __threadfence_system();
        __hip_atomic_store(done + blockIdx.x, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// dst = src (+ the W2 mirror when `mirror`): a gradient / solution vector becomes the tangent the HVP kernels read
__global__ __launch_bounds__(256) void tr_copy_kernel(float* __restrict__ dst, const float* __restrict__ src, int n, int mirror,
                                                     float scale, const ModelDesc md) {
    for (int i4 = (blockIdx.x * 256 + threadIdx.x) * 4; i4 < n; i4 += CG_NB * 1024) {
        f32x4 v = *reinterpret_cast<const f32x4*>(src + i4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * scale;
        *reinterpret_cast<f32x4*>(dst + i4) = v;
        if (mirror) {
            const int mi = w2f_mirror_of(md, i4);
            if (mi >= 0) *reinterpret_cast<f32x4*>(dst + mi) = v;
        }
    }
}
// out = hz + v * damping    (cpo.py:182: _MVP + damping_coeff * v)
__global__ __launch_bounds__(256) void tr_damp_kernel(float* __restrict__ out, const float* __restrict__ hz,
                                                     const float* __restrict__ v, float damping, int n) {
#pragma clang fp contract(off)
    for (int i4 = (blockIdx.x * 256 + threadIdx.x) * 4; i4 < n; i4 += CG_NB * 1024) {
        const f32x4 h = *reinterpret_cast<const f32x4*>(hz + i4), x = *reinterpret_cast<const f32x4*>(v + i4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = h[e] + x[e] * damping;
        *reinterpret_cast<f32x4*>(out + i4) = o;
    }
}
// CPO step direction (cpo.py:306-310): dir = combined ? inv_lam * (Hg + nu * Hb) : nu * Hb ; partial dir.dir
__global__ __launch_bounds__(256) void tr_dir_kernel(float* __restrict__ dir, const float* __restrict__ Hg,
                                                    const float* __restrict__ Hb, float inv_lam, float nu, int combined,
                                                    double* __restrict__ part, int n) {
#pragma clang fp contract(off)
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int i4 = (blockIdx.x * 256 + tid) * 4; i4 < n; i4 += CG_NB * 1024) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(Hg + i4), b = *reinterpret_cast<const f32x4*>(Hb + i4);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d[e] = combined ? inv_lam * (g[e] + nu * b[e]) : nu * b[e];
            acc += (double)d[e] * (double)d[e];
        }
        *reinterpret_cast<f32x4*>(dir + i4) = d;
    }
    const double t = cg_block_sum256(acc, sh, tid);
    if (tid == 0) part[blockIdx.x] = t;
}
// dir /= sqrt(dir.dir)   (cpo.py:310: the unit-norm step direction)
__global__ __launch_bounds__(256) void tr_unit_kernel(float* __restrict__ dir, const double* __restrict__ part, int n) {
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    const float nrm = sqrtf(cg_sum_parts(part, sh, tid));
    for (int i4 = (blockIdx.x * 256 + tid) * 4; i4 < n; i4 += CG_NB * 1024) {
        f32x4 d = *reinterpret_cast<f32x4*>(dir + i4);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = d[e] / nrm;
        *reinterpret_cast<f32x4*>(dir + i4) = d;
    }
}
// line-search candidate written straight into the parameters (+ W2 mirror): theta = coef * dir + theta0
__global__ __launch_bounds__(256) void tr_step_kernel(float* __restrict__ P, const float* __restrict__ theta0,
                                                     const float* __restrict__ dir, float coef, int n, const ModelDesc md) {
#pragma clang fp contract(off)
    for (int i4 = (blockIdx.x * 256 + threadIdx.x) * 4; i4 < n; i4 += CG_NB * 1024) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(theta0 + i4), d = *reinterpret_cast<const f32x4*>(dir + i4);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = coef * d[e] + t0[e];
        *reinterpret_cast<f32x4*>(P + i4) = v;
        const int mi = w2f_mirror_of(md, i4);
        if (mi >= 0) *reinterpret_cast<f32x4*>(P + mi) = v;
    }
}

// ---- FOCOPS: logged statistics of one minibatch step (focops.py:161-215, 232-241) and the pass-level KL sum
#define FSRL_FOCOPS_NSTATS_K 8
struct FocopsFinalArgs {
    const float* statp_vf;   // [tiles][2][FB_NSTAT]   st0 = sum (ret - V)^2
    const float* statp_pi;   // [tiles][1][FB_NSTAT]   st0 = sum masked loss rows, st2 = sum KL(new||old)
    const float* psq0; const float* psq1;    // per-block sums of squares of the critics' pre-update parameters
    int n_psq0, n_psq1;
    const float* P; int sigma_off, Da;
    float* stats;            // row: nu_loss, nu_value, actor_loss, kl, entropy, vf0, vf1, vf_total
    CtrlBlock* ctrl;
    int n_tiles, n_tiles_pi, mb, first_in_pass, last_in_pass, iters_in_pass, pass;
    float l2, nu_loss, nu_value, delta;
    int vf_stride, pi_stride;   // networks interleaved per tile in statp_vf / statp_pi (one launch over all three: 3, 3)
};
__device__ __forceinline__ void focops_finalize_row(const FocopsFinalArgs& a, const int lane) {
    double s_vf0 = 0.0, s_vf1 = 0.0, s_loss = 0.0, s_kl = 0.0, s_p0 = 0.0, s_p1 = 0.0;
    for (int t = lane; t < a.n_tiles; t += 64) {
        s_vf0 += (double)a.statp_vf[((size_t)t * a.vf_stride + 0) * FB_NSTAT];
        s_vf1 += (double)a.statp_vf[((size_t)t * a.vf_stride + 1) * FB_NSTAT];
    }
    for (int t = lane; t < a.n_tiles_pi; t += 64) {
        s_loss += (double)a.statp_pi[(size_t)t * a.pi_stride * FB_NSTAT];
        s_kl += (double)a.statp_pi[(size_t)t * a.pi_stride * FB_NSTAT + 2];
    }
    for (int k = lane; k < a.n_psq0; k += 64) s_p0 += (double)a.psq0[k];
    for (int k = lane; k < a.n_psq1; k += 64) s_p1 += (double)a.psq1[k];
    s_vf0 = wave_sum_d(s_vf0); s_vf1 = wave_sum_d(s_vf1); s_loss = wave_sum_d(s_loss); s_kl = wave_sum_d(s_kl);
    s_p0 = wave_sum_d(s_p0); s_p1 = wave_sum_d(s_p1);
    if (lane == 0) {
        const float invB = 1.0f / (float)a.mb;
        float ent = 0.0f;
        for (int d = 0; d < a.Da; ++d) ent += 1.4189385332046727f + logf(expf(a.P[a.sigma_off + d]));   // P / sigma_off: the pre-step values
        const float vf0 = (float)s_vf0 * invB + (float)s_p0 * a.l2, vf1 = (float)s_vf1 * invB + (float)s_p1 * a.l2;
        const float kl = (float)s_kl * invB;
        float* o = a.stats;
        o[0] = a.nu_loss; o[1] = a.nu_value; o[2] = (float)s_loss * invB; o[3] = kl; o[4] = ent;
        o[5] = vf0; o[6] = vf1; o[7] = vf0 + vf1;
        const double ksum = (a.first_in_pass ? 0.0 : a.ctrl->kl_sum) + (double)kl;
        a.ctrl->kl_sum = ksum;
        if (a.last_in_pass && ksum / ((double)a.iters_in_pass + 1e-7) > (double)a.delta) a.ctrl->stopped_after = a.pass;
    }
}

// torch.optim.Adam single-tensor update of one element, operation order of torch (lerp_ / mul_ + addcmul_ / sqrt / div /
// add_(eps) / addcdiv_).  ONE definition for every kernel that steps parameters of the full-batch / replay paths, so that
// they round identically.
__device__ __forceinline__ void adam_element(float* __restrict__ P, float* __restrict__ M, float* __restrict__ V, const int i,
                                             const float p, const float gs, const float coef, const float l2,
                                             const float one_minus_b1, const float beta2, const float one_minus_b2,
                                             const float step_size, const float bc2_sqrt, const float eps,
                                             const ModelDesc& md, float* __restrict__ tgt, const float tau,
                                             const float one_minus_tau) {
    const float g = gs * coef + 2.0f * l2 * p;
    float m = M[i], v = V[i];
    m = m + one_minus_b1 * (g - m);
    v = v * beta2;
    v = v + (one_minus_b2 * g) * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    M[i] = m; V[i] = v;
    const float pn = p + (-step_size * m) / denom;
    P[i] = pn;
    const int mi = w2f_mirror_of(md, i);
    if (mi >= 0) P[mi] = pn;
    if (tgt) {                           // target <- tau * param + (1 - tau) * target (BasePolicy.soft_update) in the same pass:
        const float tv = tau * pn + one_minus_tau * tgt[i];   // the parameters do not change again before sync_weight
        tgt[i] = tv;
        if (mi >= 0) tgt[mi] = tv;
    }
}

// ---- FOCOPS minibatch step, parameter side, in two launches (focops.py:161-176, 205-213, 236-246).
//      Blocks are 256 parameters wide and laid out actor | critic 0 | critic 1 (nb_a, nb_c0, nb_c1 blocks).
struct FocopsStepArgs {
    float* P; float* M; float* V;
    const float* parts; int nparts, stride;  // split-K partial gradients of fb_wgrad_kernel (all three networks)
    float* G;                                // summed actor gradient (written by prep, read by the step)
    float* gsq;                              // [nb_a] per-block sums of squares of the actor gradient
    float* psq;                              // [nb_c0 + nb_c1] per-block sums of squares of the critics' PRE-update parameters
    float* sig_stash;                        // [Da] sigma_param before the step (entropy of the pre-update policy)
    // three-launch step (ppo_wgrad_kernel instead of fb_wgrad_kernel + focops_prep_kernel): G holds the final gradient of all
    // three networks, the actor's squared norm comes as ppo_wgrad's per-block sums + its extra block's per-network share, and
    // the step leaves sum(theta^2) of the critics / sigma_param AFTER the update for the next step's logged row
    const float* gsq_part; int n_gsq_part; const float* gsq_net;
    float* psq_next; float* sig_next;
    int nb_a, nb_c0, nb_c1;
    float max_norm, l2;
    float one_minus_b1, beta2, one_minus_b2, adam_eps;
    float step_a, bc2s_a, step_c, bc2s_c;    // lr / (1 - beta1^t), sqrt(1 - beta2^t) of the two optimisers
    FocopsFinalArgs fin;                     // the logged row / pass KL bookkeeping (done by the extra block of the step)
};
__device__ __forceinline__ void focops_block(const ModelDesc& md, const FocopsStepArgs& a, int& net, int& i) {
    int b = blockIdx.x;
    net = 0;
    if (b >= a.nb_a) { b -= a.nb_a; net = 1; if (b >= a.nb_c0) { b -= a.nb_c0; net = 2; } }
    i = md.net[net].begin + b * 256 + threadIdx.x;
}
// prep: actor blocks add the split-K partials in z order (-> G) and leave per-block sums of squares (clip_grad_norm_ over
// the actor); critic blocks leave per-block sums of squares of their parameters (the L2 term of the logged loss).
__global__ __launch_bounds__(256) void focops_prep_kernel(const ModelDesc md, const FocopsStepArgs a) {
    __shared__ float sh[4];
    int net, i;
    focops_block(md, a, net, i);
    float q = 0.0f;
    if (i < md.net[net].end) {
        if (net == 0) {
            if (a.nparts > 0) {                   // nparts == 0: called for the parameter sums only (start of a three-launch pass)
                float v = a.parts[i];
                for (int z = 1; z < a.nparts; ++z) v += a.parts[(size_t)z * a.stride + i];
                a.G[i] = v;
                q = v * v;
            }
        } else {
            const float p = a.P[i];
            q = p * p;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < md.Da) a.sig_stash[threadIdx.x] = a.P[md.net[0].sigma + threadIdx.x];
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        if (net == 0) a.gsq[blockIdx.x] = t; else a.psq[blockIdx.x - a.nb_a] = t;
    }
}
// step: Adam of the actor (clipped to max_norm) and of both critics (L2 inside the gradient) in one launch; one extra
// block writes the logged row from the per-tile statistics, the parameter sums and the stashed sigma_param.
__global__ __launch_bounds__(256) void focops_step_kernel(const ModelDesc md, const FocopsStepArgs a) {
    __shared__ double shd[4];
    __shared__ float coef_s;
    __shared__ float shq[4];
    if ((int)blockIdx.x == a.nb_a + a.nb_c0 + a.nb_c1) {
        if (threadIdx.x < 64) focops_finalize_row(a.fin, threadIdx.x);
        return;
    }
    int net, i;
    focops_block(md, a, net, i);
    float coef = 1.0f;
    if (net == 0 && a.max_norm > 0.0f) {     // same reduction order as adam_range_kernel's clip
        double sq = 0.0;
        if (a.gsq_part) {                    // the actor's blocks of ppo_wgrad_kernel come first, then its share of the extra block
            for (int k = threadIdx.x; k < a.n_gsq_part; k += 256) sq += (double)a.gsq_part[k];
            if (threadIdx.x == 0) sq += (double)a.gsq_net[0];
        } else
        for (int k = threadIdx.x; k < a.nb_a; k += 256) sq += (double)a.gsq[k];
        sq = wave_sum_d(sq);
        if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) coef_s = fminf(a.max_norm / (sqrtf((float)((shd[0] + shd[1]) + (shd[2] + shd[3]))) + 1e-6f), 1.0f);
        __syncthreads();
        coef = coef_s;
    }
    float pn = 0.0f;
    if (i < md.net[net].end) {
        const float p = a.P[i];
        float gs;
        if (net == 0 || a.nparts == 0) gs = a.G[i];
        else {
            gs = a.parts[i];
            for (int z = 1; z < a.nparts; ++z) gs += a.parts[(size_t)z * a.stride + i];
        }
        if (net == 0) adam_element(a.P, a.M, a.V, i, p, gs, coef, 0.0f, a.one_minus_b1, a.beta2, a.one_minus_b2, a.step_a, a.bc2s_a,
                                   a.adam_eps, md, nullptr, 0.0f, 0.0f);
        else adam_element(a.P, a.M, a.V, i, p, gs, coef, a.l2, a.one_minus_b1, a.beta2, a.one_minus_b2, a.step_c, a.bc2s_c, a.adam_eps,
                          md, nullptr, 0.0f, 0.0f);
        pn = a.P[i];                          // what this thread just wrote
        if (net == 0 && a.sig_next && (unsigned)(i - md.net[0].sigma) < (unsigned)md.Da) a.sig_next[i - md.net[0].sigma] = pn;
    }
    if (net != 0 && a.psq_next) {             // sum(theta^2) of this block AFTER the step: the next step's pre-update value
        float q = wave_sum(pn * pn);
        if ((threadIdx.x & 63) == 0) shq[threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0) a.psq_next[blockIdx.x - a.nb_a] = (shq[0] + shq[1]) + (shq[2] + shq[3]);
    }
}

// full-batch advantage normalisation (CPO cpo.py:127-131, TRPO trpo_lag.py:129-133): per critic
// (a - mean) / std with the unbiased std, float64 accumulate.  grid = C blocks of 1024 threads.
__global__ __launch_bounds__(1024) void fb_advnorm_kernel(float* __restrict__ advs, int N) {
    __shared__ double sh[16];
    __shared__ double mean_s, sd_s;
    float* a = advs + (size_t)blockIdx.x * N;
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < N; i += 1024) s += (double)a[i];
    s = wave_sum_d(s);
    if ((tid & 63) == 0) sh[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) { double t = 0; for (int w = 0; w < 16; ++w) t += sh[w]; mean_s = t / (double)N; }
    __syncthreads();
    const double mean = mean_s;
    double qv = 0.0;
    for (int i = tid; i < N; i += 1024) { const double d = (double)a[i] - mean; qv += d * d; }
    qv = wave_sum_d(qv);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = qv;
    __syncthreads();
    if (tid == 0) { double t = 0; for (int w = 0; w < 16; ++w) t += sh[w]; sd_s = sqrt(t / (double)(N - 1)); }
    __syncthreads();
    const float mf = (float)mean_s, sf = (float)sd_s;
    for (int i = tid; i < N; i += 1024) a[i] = (a[i] - mf) / sf;
}

// row data in store order for the full-batch kernels (identity permutation)
struct FbRowArgs {
    const float* act; const float* advs; const float* rets; const float* logp_old;
    const float* mean_old;   // [N][Da] (may be null: filled later by fb_tile EVAL? no: by infer)
    const float* sigma;      // sigma_param[Da] at process time (std_old = exp)
    float* rd;
    int N, C, Da;
};
__global__ void fb_rowdata_kernel(const FbRowArgs a) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < a.N * FSRL_RD; e += gridDim.x * blockDim.x) {
        const int r = e / FSRL_RD, f = e - r * FSRL_RD;
        float v = 0.0f;
        if (f < a.Da) v = a.act[(size_t)r * a.Da + f];
        else if (f == FSRL_RD_LOGP) v = a.logp_old[r];
        else if (f >= FSRL_RD_ADV && f < FSRL_RD_ADV + a.C) v = a.advs[(size_t)(f - FSRL_RD_ADV) * a.N + r];
        else if (f >= FSRL_RD_RET && f < FSRL_RD_RET + a.C) v = a.rets[(size_t)(f - FSRL_RD_RET) * a.N + r];
        else if (f >= FSRL_RD_MEAN && f < FSRL_RD_MEAN + a.Da) v = a.mean_old[(size_t)r * a.Da + f - FSRL_RD_MEAN];
        else if (f >= FSRL_RD_STD && f < FSRL_RD_STD + a.Da) v = expf(a.sigma[f - FSRL_RD_STD]);
        a.rd[e] = v;
    }
}

// mean_old / std_old columns of n rows of row data := the given means and exp(sigma_param)
__global__ void fb_rd_old_kernel(float* __restrict__ rd, const float* __restrict__ mean, const float* __restrict__ sigma,
                                 int n, int Da) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * 32; e += gridDim.x * blockDim.x) {
        const int r = e >> 5, f = e & 31;
        float v = 0.0f;
        if (f < Da) v = mean[(size_t)r * Da + f];
        else if (f >= 16 && f < 16 + Da) v = expf(sigma[f - 16]);
        rd[(size_t)r * FSRL_RD + FSRL_RD_MEAN + f] = v;         // MEAN [32, 48) and STD [48, 64) are adjacent
    }
}
// Batch.split(shuffle=True) of the trust-region learn loops: row j of the permuted copy = row perm[j] of the batch
__global__ void fb_gather_rows_kernel(float* __restrict__ obs_p, float* __restrict__ rd_p, const float* __restrict__ obs,
                                      const float* __restrict__ rd, const int* __restrict__ perm, int n, int Do) {
    const int per = Do + FSRL_RD;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < (long long)n * per; e += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(e / per), f = (int)(e - (long long)j * per);
        const int r = perm[j];
        if (f < Do) obs_p[(size_t)j * Do + f] = obs[(size_t)r * Do + f];
        else rd_p[(size_t)j * FSRL_RD + f - Do] = rd[(size_t)r * FSRL_RD + f - Do];
    }
}

// Adam on a parameter range with optional L2 term (CPO critics: loss += l2 * sum(theta^2)),
// and the per-network sum of squares of the PRE-update parameters (for the logged vf loss).
__global__ __launch_bounds__(256) void adam_range_kernel(float* __restrict__ P, float* __restrict__ M,
                                                        float* __restrict__ V, const float* __restrict__ G,
                                                        int begin, int end, float l2, float one_minus_b1,
                                                        float beta2, float one_minus_b2, float step_size,
                                                        float bc2_sqrt, float eps, int nparts, int stride,
                                                        const ModelDesc md, const float* __restrict__ gsq_part = nullptr,
                                                        int n_gsq = 0, float max_norm = 0.0f,
                                                        float* __restrict__ psq_part = nullptr,
                                                        float* __restrict__ tgt = nullptr, float tau = 0.0f,
                                                        float one_minus_tau = 0.0f, int sum64 = 0) {
    __shared__ double shd[4];
    __shared__ float coef_s, shf[4];
    const int i = begin + blockIdx.x * 256 + threadIdx.x;
    float coef = 1.0f;
    if (gsq_part && max_norm > 0.0f) {       // clip_grad_norm_(max_norm) over the range: same rule as adam_clip_kernel
        double sq = 0.0;
        for (int k = threadIdx.x; k < n_gsq; k += 256) sq += (double)gsq_part[k];
        sq = wave_sum_d(sq);
        if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) coef_s = fminf(max_norm / (sqrtf((float)((shd[0] + shd[1]) + (shd[2] + shd[3]))) + 1e-6f), 1.0f);
        __syncthreads();
        coef = coef_s;
    }
    float psq = 0.0f;
    if (i < end) {
        const float p = P[i];
        psq = p * p;
        float gs = G[i];                                   // split-K partials of fb_wgrad_kernel, z order
        if (sum64) {                                       // ... combined in float64 and rounded once, as fb_sum_parts_kernel does
            gs = (float)sum_parts1(G, (size_t)stride, nparts, (size_t)i);
        } else
        for (int z = 1; z < nparts; ++z) gs += G[(size_t)z * stride + i];
        adam_element(P, M, V, i, p, gs, coef, l2, one_minus_b1, beta2, one_minus_b2, step_size, bc2_sqrt, eps, md, tgt, tau,
                     one_minus_tau);
    }
    if (psq_part) {                          // per-block sum of squares of the PRE-update parameters (L2 term of the logged loss)
        psq = wave_sum(psq);
        if ((threadIdx.x & 63) == 0) shf[threadIdx.x >> 6] = psq;
        __syncthreads();
        if (threadIdx.x == 0) psq_part[blockIdx.x] = (shf[0] + shf[1]) + (shf[2] + shf[3]);
    }
}

// sum of squares of P[begin, end): blockIdx.x strides over the range, out[blockIdx.y * gridDim.x + blockIdx.x] = this block's
// partial (float64); the caller adds the gridDim.x partials in index order.  grid.y selects one of `ranges` (begin, end) pairs.
struct SumsqRanges { int begin[FSRL_MAX_NETS], end[FSRL_MAX_NETS]; };
__global__ __launch_bounds__(256) void sumsq_range_kernel(const float* __restrict__ P, const SumsqRanges rg,
                                                         double* __restrict__ out) {
    __shared__ double sh[4];
    const int tid = threadIdx.x;
    const int begin = rg.begin[blockIdx.y], end = rg.end[blockIdx.y];
    double s = 0.0;
    for (int i = begin + blockIdx.x * 256 + tid; i < end; i += 256 * gridDim.x) s += (double)P[i] * (double)P[i];
    s = wave_sum_d(s);
    if ((tid & 63) == 0) sh[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
