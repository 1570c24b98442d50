// kernels_cvpo.hpp -- CVPO update kernels (fsrl/policy/cvpo.py:206-430).  The replay gather, the
// float64 n-step target, the Q-network launches, the weight-gradient / Adam / Polyak kernels are the
// SAC path's (kernels_sac.hpp, kernels_fb.hpp); this file holds what is CVPO's own:
//   * the Gaussian actor tile (mu = max_action * tanh(head), sigma = exp(clamp(head))) in four modes:
//     target action, the K particles of actor_old, the M-step statistics, the M-step backward;
//   * the E-step: Adam on the two duals (eta, lambda) of the logsumexp loss and the softmax weights;
//   * the M-step dual step (Adam on the two KL multipliers) and the logged statistics.
#pragma once
#include "kernels_sac.hpp"

#define FSRL_CVPO_NSTATS_K 17
#define CVPO_EPS10 1.1920928955078125e-06f      // np.finfo(np.float32).eps * 10   (cvpo.py:163)

// scalars that live on the device between updates
struct CvpoScalars {
    float eta, lam;              // estep_dual[0], estep_dual[1]                  (cvpo.py:150-155)
    float em[2], ev[2]; int et;  // Adam moments / step count of the E-step duals
    float mdual[2];              // mstep_dual_mu, mstep_dual_std (stored unclipped)  (cvpo.py:178-188)
    float mm[2], mv[2]; int mt;  // their Adam state (reset by pre_update_fn)
    float dual_mu, dual_std;     // clipped to [0, mstep_dual_max]: what the current M-step backward uses
    float estep_loss;            // logged value of the first E-step iteration
    float mstats[8];             // kl_mu kl_std loss_kl loss_mle loss_total dual_mu dual_std entropy (first M iteration)
};

#define CVPO_A_TARGET 0      // a' = mu + sigma * eps at s_{t+n}  -> action columns of X
#define CVPO_A_PARTICLES 1   // actor_old at s_t: mu_old, std_old, K particles -> all columns of XK
#define CVPO_A_MFWD 2        // actor at s_t: per-tile sums of w*loglik, KL_mu, KL_std, entropy
#define CVPO_A_MBWD 3        // same forward + gradient of loss_mle + dual_mu*KL_mu + dual_std*KL_std
struct CvpoActorArgs {
    const float* obs;        // [B][Do]
    const float* eps;        // TARGET: [B][Da] ; PARTICLES: [K][B][Da]
    float* X;                // TARGET: [B][Do+Da] ; PARTICLES: [K*B][Do+Da], row k*B + b
    float* mu_old; float* std_old;   // [B][Da]   written by PARTICLES, read by the M modes
    const float* W;          // [K][B] E-step weights
    const float* XK;         // [K*B][Do+Da] particles (M modes)
    const CvpoScalars* sc;
    float* A1; float* A2; float* D1; float* D2; float* DO;   // side buffers (MBWD)
    float* statp;            // [n_tiles][FB_NSTAT]
    int B, K, mode;
    float max_action;
    // a second batch in the same launch (tiles_half > 0): workgroups [tiles_half, 2 * tiles_half) run mode2 with the actor
    // P2 on obs2 / eps2 into X2.  Used for TARGET (current actor at s_{t+n}) + PARTICLES (actor_old at s_t): neither
    // depends on the critic step in between.
    const float* P2; const float* obs2; const float* eps2; float* X2; int mode2, tiles_half;
};

template <int H, int R>
__global__ __launch_bounds__(4 * H) void cvpo_actor_tile_kernel(const float* __restrict__ P_,
                                                               const ModelDesc md, const CvpoActorArgs a) {
    __shared__ TileSmem<H> sm;
    constexpr int NT = TileGeom<H>::NT;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
    const bool second = a.tiles_half > 0 && (int)blockIdx.x >= a.tiles_half;
    const float* __restrict__ P = second ? a.P2 : P_;
    const float* __restrict__ obs_ = second ? a.obs2 : a.obs;
    const float* __restrict__ eps_ = second ? a.eps2 : a.eps;
    float* __restrict__ X_ = second ? a.X2 : a.X;
    const int mode = second ? a.mode2 : a.mode;
    const int row0 = (second ? (int)blockIdx.x - a.tiles_half : (int)blockIdx.x) * R;
    const NetOff no = md.net[0];
    const int Do = md.Do, Da = md.Da, Din = Do + Da;
    const int n_valid = max(0, min(R, a.B - row0));

    TileStage<H> stg;
    stg.issue(P, no, Do, 0, obs_ + (size_t)row0 * Do, nullptr, n_valid, tid);
    FwdW2Frag<H> wf;
    wf.load(P + no.W2f, wave, lane);
    for (int e = tid; e < 16 * FSRL_DOW; e += NT) sm.dout[e] = 0.0f;
    stg.commit(sm, no, Do, tid);
    __syncthreads();
    tile_forward<H, R, true>(sm, P, no, Do, tid, wf);     // sm.out[i][0..Da) = mean head, [Da..2Da) = raw log sigma

    float wb[H / 16][4];
    if (mode == CVPO_A_MBWD) {
        const float* __restrict__ W2c = P + no.W2 + wave * 16 + li;
#pragma unroll
        for (int jc = 0; jc < H / 16; ++jc) {
#pragma unroll
            for (int s = 0; s < 4; ++s) wb[jc][s] = W2c[(size_t)(16 * jc + 4 * q + s) * H];
        }
    }
    if (mode == CVPO_A_PARTICLES) {               // observation columns of the K replicated rows
        const int per = n_valid * Do;
        for (int e = tid; e < a.K * per; e += NT) {
            const int k = e / per, w = e - k * per;
            const int i = w / Do, f = w - i * Do;
            X_[((size_t)k * a.B + row0 + i) * Din + f] = obs_[(size_t)(row0 + i) * Do + f];
        }
    }
    if (tid < 16 * R) {
        const int i = tid >> 4, d = tid & 15;
        const int r = row0 + i;
        const bool valid = i < n_valid, on = valid && d < Da;
        float th = 0.0f, mu = 0.0f, sig = 1.0f, pass = 0.0f;
        if (on) {
            th = tanhf(sm.out[i * FSRL_MAX_ACT + d]);
            mu = a.max_action * th;
            const float lraw = sm.out[i * FSRL_MAX_ACT + Da + d];
            pass = (lraw >= SAC_LOG_SIG_MIN && lraw <= SAC_LOG_SIG_MAX) ? 1.0f : 0.0f;
            sig = expf(fminf(fmaxf(lraw, SAC_LOG_SIG_MIN), SAC_LOG_SIG_MAX));
        }
        if (mode == CVPO_A_TARGET) {
            if (on) X_[(size_t)r * Din + Do + d] = eps_[(size_t)r * Da + d] * sig + mu;
        } else if (mode == CVPO_A_PARTICLES) {
            if (on) {
                a.mu_old[(size_t)r * Da + d] = mu;
                a.std_old[(size_t)r * Da + d] = sig;
                for (int k = 0; k < a.K; ++k) {
                    const size_t rk = (size_t)k * a.B + r;
                    X_[rk * Din + Do + d] = eps_[rk * Da + d] * sig + mu;     // Normal.sample: eps * std + mean
                }
            }
        } else {
            // ---- M-step row terms (cvpo.py:378-417), per action dimension then summed over d
            float mle = 0.0f, klm = 0.0f, kls = 0.0f, ent = 0.0f;
            if (on) {
                const float mu_o = a.mu_old[(size_t)r * Da + d], sd_o = a.std_old[(size_t)r * Da + d];
                const float var_o = sd_o * sd_o, var = sig * sig;
                const float lso = logf(sd_o), ls = logf(sig);
                float s_w = 0.0f, s_wdm = 0.0f, s_wdo2 = 0.0f;     // sum_k w, w*(a-mu), w*(a-mu_old)^2
                for (int k = 0; k < a.K; ++k) {
                    const size_t rk = (size_t)k * a.B + r;
                    const float w = a.W[rk], ak = a.XK[rk * Din + Do + d];
                    const float dm = ak - mu, dmo = ak - mu_o;
                    // Normal(mu, std_old).log_prob(a) + Normal(mu_old, std).log_prob(a)
                    const float ll = (-(dm * dm) / (2.0f * var_o) - lso - LOG_SQRT_2PI) +
                                     (-(dmo * dmo) / (2.0f * var) - ls - LOG_SQRT_2PI);
                    mle = fmaf(w, ll, mle);
                    s_w += w; s_wdm = fmaf(w, dm, s_wdm); s_wdo2 = fmaf(w, dmo * dmo, s_wdo2);
                }
                const float var_oc = fmaxf(var_o, 1e-6f), var_c = fmaxf(var, 1e-6f);     // gaussian_kl clamps
                const float dmu = mu_o - mu;
                klm = 0.5f * (dmu * dmu) / var_oc;
                kls = 0.5f * (logf(var_c / var_oc) + var_oc / var_c - 1.0f);
                ent = (0.5f + 0.5f * 1.8378770664093453f + lso) + (0.5f + 0.5f * 1.8378770664093453f + ls);
                if (mode == CVPO_A_MBWD) {
                    const float invB = 1.0f / (float)a.B, invKB = invB / (float)a.K;
                    const float dual_mu = a.sc->dual_mu, dual_std = a.sc->dual_std;
                    // d loss_mle / d mu, / d sigma   (loss_mle = -mean_{k,b} w * loglik)
                    float g_mu = -invKB * (s_wdm / var_o);
                    float g_sg = -invKB * (s_wdo2 / (var * sig) - s_w / sig);
                    // KL penalties: d kl_mu / d mu = (mu - mu_old) / var_old_c / B ; d kl_std / d sigma (0 where var is clamped)
                    g_mu += dual_mu * invB * (-dmu / var_oc);
                    if (var > 1e-6f) g_sg += dual_std * invB * (1.0f / sig - var_oc / (var * sig));
                    sm.dout[i * FSRL_DOW + d] = g_mu * a.max_action * (1.0f - th * th);
                    sm.dout[i * FSRL_DOW + Da + d] = g_sg * sig * pass;
                }
            }
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            for (int dd = 0; dd < Da; ++dd) {
                const int src = (lane & 48) + dd;
                s0 += __shfl(mle, src, 64); s1 += __shfl(klm, src, 64);
                s2 += __shfl(kls, src, 64); s3 += __shfl(ent, src, 64);
            }
            if (d == 0) {
                sm.w1[i * FB_NSTAT + 0] = valid ? s0 : 0.0f; sm.w1[i * FB_NSTAT + 1] = valid ? s1 : 0.0f;
                sm.w1[i * FB_NSTAT + 2] = valid ? s2 : 0.0f; sm.w1[i * FB_NSTAT + 3] = valid ? s3 : 0.0f;
            }
        }
    }
    if (mode < CVPO_A_MFWD) return;
    __syncthreads();
    if (mode == CVPO_A_MFWD) {
        if (tid < 4) {
            float t = 0.0f;
            for (int i = 0; i < R; ++i) t += sm.w1[i * FB_NSTAT + tid];
            a.statp[(size_t)blockIdx.x * FB_NSTAT + tid] = t;
        }
        return;
    }
    tile_backward<H, R>(sm, no, wb, a.A1 + (size_t)row0 * H, a.A2 + (size_t)row0 * H, a.D1 + (size_t)row0 * H,
                        a.D2 + (size_t)row0 * H, a.DO + (size_t)row0 * FSRL_DOW, tid, false);
}

// ---- E-step (cvpo.py:278-288, 341-371): one workgroup.  Q values arrive as [n_q][K*B] (row k*B + b);
//      q0 / q1 are [B][K] scratch.  The reference's in-place aliasing is kept: every dual-loss evaluation leaves
//      q0 lowered by (pre-step lambda) * q1, and the weights subtract (post-step, clamped lambda) * q1 again.
struct CvpoEstepArgs {
    const float* QK; float* q0; float* q1; float* W;
    CvpoScalars* sc;
    int B, K, n_q, iters;
    float kl, thres, lr, dual_max, beta1, beta2, adam_eps;
};
// Adam step of the two E-step duals from the batch means (shared by both work distributions below)
__device__ __forceinline__ void cvpo_estep_adam(const CvpoEstepArgs& a, const int it, const float eta, const float lam,
                                                const float m_lse, const float m_pc, const float m_pq, float* duals) {
    CvpoScalars sc = *a.sc;
    const float loss = eta * a.kl + lam * a.thres + eta * m_lse;
    if (it == 0) sc.estep_loss = loss;
    const float g[2] = {a.kl + m_lse - m_pc / eta, a.thres - m_pq};
    sc.et += 1;
    const double bc1 = 1.0 - pow((double)a.beta1, (double)sc.et), bc2 = 1.0 - pow((double)a.beta2, (double)sc.et);
    const float step_size = (float)((double)a.lr / bc1);
    float nd[2] = {eta, lam};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        sc.em[j] = sc.em[j] + (float)(1.0 - (double)a.beta1) * (g[j] - sc.em[j]);
        sc.ev[j] = sc.ev[j] * a.beta2;
        sc.ev[j] = sc.ev[j] + ((float)(1.0 - (double)a.beta2) * g[j]) * g[j];
        const float denom = sqrtf(sc.ev[j]) / (float)sqrt(bc2) + a.adam_eps;
        nd[j] = nd[j] + (-step_size * sc.em[j]) / denom;
    }
    if (it == a.iters - 1) {                          // estep_dual.data.clamp_(eps, max) after the loop
        nd[0] = fminf(fmaxf(nd[0], CVPO_EPS10), a.dual_max);
        nd[1] = fminf(fmaxf(nd[1], CVPO_EPS10), a.dual_max);
    }
    sc.eta = nd[0]; sc.lam = nd[1];
    *a.sc = sc;
    duals[0] = nd[0]; duals[1] = nd[1];
}

// sum / max over the KP adjacent lanes that hold one state's particles (KP a power of two <= 64)
__device__ __forceinline__ float row_sum(float v, const int KP) {
    for (int w = 1; w < KP; w <<= 1) v += __shfl_xor(v, w, 64);
    return v;
}
__device__ __forceinline__ float row_max(float v, const int KP) {
    for (int w = 1; w < KP; w <<= 1) v = fmaxf(v, __shfl_xor(v, w, 64));
    return v;
}

__global__ __launch_bounds__(1024) void cvpo_estep_kernel(const CvpoEstepArgs a) {
    __shared__ double red[3][1024];
    __shared__ float duals[2];
    const int tid = threadIdx.x;
    const size_t KB = (size_t)a.K * a.B;
    const float logK = logf((float)a.K);
    if (tid == 0) { duals[0] = a.sc->eta; duals[1] = a.sc->lam; }
    auto q_of = [&](const size_t rk, float& v0, float& v1) {
        if (a.n_q == 2) { v0 = a.QK[rk]; v1 = a.QK[KB + rk]; }
        else { v0 = fminf(a.QK[rk], a.QK[KB + rk]); v1 = fminf(a.QK[2 * KB + rk], a.QK[3 * KB + rk]); }
    };
    auto block_sums = [&](const double s0, const double s1, const double s2) {
        red[0][tid] = s0; red[1][tid] = s1; red[2][tid] = s2;
        __syncthreads();
        for (int w = 512; w > 0; w >>= 1) {           // fixed tree: order independent of scheduling
            if (tid < w) { red[0][tid] += red[0][tid + w]; red[1][tid] += red[1][tid + w]; red[2][tid] += red[2][tid + w]; }
            __syncthreads();
        }
    };
    if ((a.K & (a.K - 1)) == 0 && a.K <= 64) {
        // ---- one thread per (state b, particle k): element e = b*K + k, a state's particles in K adjacent lanes
        const int K = a.K, n = (int)KB;
        for (int e = tid; e < n; e += 1024) {
            const int b = e / K, k = e - b * K;
            float v0, v1;
            q_of((size_t)k * a.B + b, v0, v1);
            a.q0[e] = v0; a.q1[e] = v1;
        }
        __syncthreads();
        for (int it = 0; it < a.iters; ++it) {
            const float eta = duals[0], lam = duals[1];
            double s_lse = 0.0, s_pc = 0.0, s_pq = 0.0;
            for (int e0 = 0; e0 < n; e0 += 1024) {    // n is a multiple of K and 1024 of K: rows never straddle
                const int e = e0 + tid;
                const bool on = e < n;
                const float q1v = on ? a.q1[e] : 0.0f;
                const float cq = on ? a.q0[e] - lam * q1v : -INFINITY;
                if (on) a.q0[e] = cq;                 // combined_q aliases q_values[0]
                const float z = cq / eta;
                const float mx = row_max(z, K);
                const float ex = on ? expf(z - mx) : 0.0f;
                const float se = row_sum(ex, K);
                const float p = ex / se;
                const float pc = row_sum(on ? p * cq : 0.0f, K), pq = row_sum(p * q1v, K);
                if (on && (e & (K - 1)) == 0) { s_lse += (double)((mx + logf(se)) - logK); s_pc += (double)pc; s_pq += (double)pq; }
            }
            block_sums(s_lse, s_pc, s_pq);
            if (tid == 0)
                cvpo_estep_adam(a, it, eta, lam, (float)(red[0][0] / a.B), (float)(red[1][0] / a.B), (float)(red[2][0] / a.B), duals);
            __syncthreads();
        }
        const float eta = duals[0], lam = duals[1];
        for (int e0 = 0; e0 < n; e0 += 1024) {
            const int e = e0 + tid;
            const bool on = e < n;
            const float z = on ? (a.q0[e] - lam * a.q1[e]) / eta : -INFINITY;
            const float mx = row_max(z, K);
            const float ex = on ? expf(z - mx) : 0.0f;
            const float se = row_sum(ex, K);
            if (on) { const int b = e / K, k = e - b * K; a.W[(size_t)k * a.B + b] = ex / se; }
        }
        return;
    }
    // ---- any K: one thread per state, particles in a sequential loop
    for (int b = tid; b < a.B; b += 1024)
        for (int k = 0; k < a.K; ++k) {
            float v0, v1;
            q_of((size_t)k * a.B + b, v0, v1);
            a.q0[(size_t)b * a.K + k] = v0; a.q1[(size_t)b * a.K + k] = v1;
        }
    __syncthreads();
    for (int it = 0; it < a.iters; ++it) {
        const float eta = duals[0], lam = duals[1];
        double s_lse = 0.0, s_pc = 0.0, s_pq = 0.0;
        for (int b = tid; b < a.B; b += 1024) {
            float* r0 = a.q0 + (size_t)b * a.K; const float* r1 = a.q1 + (size_t)b * a.K;
            float mx = -INFINITY;
            for (int k = 0; k < a.K; ++k) {
                const float cq = r0[k] - lam * r1[k];
                r0[k] = cq;                                    // combined_q aliases q_values[0]
                mx = fmaxf(mx, cq / eta);
            }
            float se = 0.0f;
            for (int k = 0; k < a.K; ++k) se += expf(r0[k] / eta - mx);
            float pc = 0.0f, pq = 0.0f;
            for (int k = 0; k < a.K; ++k) {
                const float p = expf(r0[k] / eta - mx) / se;
                pc = fmaf(p, r0[k], pc); pq = fmaf(p, r1[k], pq);
            }
            s_lse += (double)((mx + logf(se)) - logK); s_pc += (double)pc; s_pq += (double)pq;
        }
        block_sums(s_lse, s_pc, s_pq);
        if (tid == 0)
            cvpo_estep_adam(a, it, eta, lam, (float)(red[0][0] / a.B), (float)(red[1][0] / a.B), (float)(red[2][0] / a.B), duals);
        __syncthreads();
    }
    // optimal non-parametric distribution: softmax over the K particles of (q0 - lambda * q1) / eta
    const float eta = duals[0], lam = duals[1];
    for (int b = tid; b < a.B; b += 1024) {
        const float* r0 = a.q0 + (size_t)b * a.K; const float* r1 = a.q1 + (size_t)b * a.K;
        float mx = -INFINITY;
        for (int k = 0; k < a.K; ++k) mx = fmaxf(mx, (r0[k] - lam * r1[k]) / eta);
        float se = 0.0f;
        for (int k = 0; k < a.K; ++k) se += expf((r0[k] - lam * r1[k]) / eta - mx);
        for (int k = 0; k < a.K; ++k) a.W[(size_t)k * a.B + b] = expf((r0[k] - lam * r1[k]) / eta - mx) / se;
    }
}

// ---- M-step dual step (cvpo.py:392-405): KL means over the tiles, Adam on (mstep_dual_mu, mstep_dual_std),
//      the clipped multipliers for the backward, and the logged values of the first iteration.
struct CvpoMdualArgs {
    const float* statp; int n_tiles, B, K;
    CvpoScalars* sc;
    float kl_mu_eps, kl_std_eps, dual_max, lr, beta1, beta2, adam_eps;
    int log_it;
};
__global__ __launch_bounds__(64) void cvpo_mdual_kernel(const CvpoMdualArgs a) {
    const int lane = threadIdx.x;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t = lane; t < a.n_tiles; t += 64) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += (double)a.statp[(size_t)t * FB_NSTAT + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = wave_sum_d(s[k]);
    if (lane != 0) return;
    CvpoScalars sc = *a.sc;
    const float loss_mle = -(float)(s[0] / ((double)a.B * a.K));
    const float kl_mu = (float)(s[1] / a.B), kl_std = (float)(s[2] / a.B), ent = (float)(s[3] / a.B);
    const float g[2] = {a.kl_mu_eps - kl_mu, a.kl_std_eps - kl_std};
    sc.mt += 1;
    const double bc1 = 1.0 - pow((double)a.beta1, (double)sc.mt), bc2 = 1.0 - pow((double)a.beta2, (double)sc.mt);
    const float step_size = (float)((double)a.lr / bc1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        sc.mm[j] = sc.mm[j] + (float)(1.0 - (double)a.beta1) * (g[j] - sc.mm[j]);
        sc.mv[j] = sc.mv[j] * a.beta2;
        sc.mv[j] = sc.mv[j] + ((float)(1.0 - (double)a.beta2) * g[j]) * g[j];
        const float denom = sqrtf(sc.mv[j]) / (float)sqrt(bc2) + a.adam_eps;
        sc.mdual[j] = sc.mdual[j] + (-step_size * sc.mm[j]) / denom;
    }
    sc.dual_mu = fminf(fmaxf(sc.mdual[0], 0.0f), a.dual_max);
    sc.dual_std = fminf(fmaxf(sc.mdual[1], 0.0f), a.dual_max);
    if (a.log_it) {
        const float loss_kl = sc.dual_mu * (kl_mu - a.kl_mu_eps) + sc.dual_std * (kl_std - a.kl_std_eps);
        sc.mstats[0] = kl_mu; sc.mstats[1] = kl_std; sc.mstats[2] = loss_kl; sc.mstats[3] = loss_mle;
        sc.mstats[4] = loss_mle + loss_kl; sc.mstats[5] = sc.dual_mu; sc.mstats[6] = sc.dual_std; sc.mstats[7] = ent;
    }
    *a.sc = sc;
}

// ---- one row of logged statistics, in the order the reference's logger receives them (cvpo.py:248-276, 341-417)
struct CvpoFinalArgs {
    const float* statp_q; const float* Y; const CvpoScalars* sc; float* stats;
    int n_tiles_q, n_q, B;
    float thres;
};
__device__ __forceinline__ void cvpo_finalize_row(const CvpoFinalArgs& a, const int lane) {
    double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};     // td^2 of up to four Q-nets, sum y_r, sum y_c
    for (int t = lane; t < a.n_tiles_q; t += 64) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < a.n_q) s[k] += (double)a.statp_q[((size_t)t * a.n_q + k) * FB_NSTAT];
    }
    for (int b = lane; b < a.B; b += 64) { s[4] += (double)a.Y[b]; s[5] += (double)a.Y[(size_t)a.B + b]; }
    float m[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) m[k] = (float)(wave_sum_d(s[k]) / (double)a.B);
    if (lane != 0) return;
    const bool single = a.n_q == 2;
    const float lq0 = single ? m[0] : m[0] + m[1], lq1 = single ? m[1] : m[2] + m[3];
    const CvpoScalars sc = *a.sc;
    float* o = a.stats;
    o[0] = sc.estep_loss; o[1] = sc.eta; o[2] = sc.lam;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[3 + k] = sc.mstats[k];
    o[11] = lq0; o[12] = m[4]; o[13] = lq1; o[14] = m[5]; o[15] = a.thres; o[16] = lq0 + lq1;
}
