// kernels_layered_sac.hpp -- heads of the replay agents (SAC-Lag, DDPG-Lag) on LAYERED networks (hidden_sizes of any depth / width,
// kernels_layered.hpp): the head arithmetic of sac_actor_tile_kernel (kernels_sac.hpp) and of fb_tile_body's Q modes
// (kernels_fb.hpp) on head outputs that lin_kernel<LIN_F> launches left in a LayWork's `out` buffer.  Included by host_sac.inc.
#pragma once
#include "kernels_layered.hpp"
#include "kernels_sac.hpp"
#include "kernels_cvpo.hpp"

// ---- actor head.  FWD: a = tanh(mu + sigma eps) (DDPG: max_action tanh(out)) into the action columns of X, log pi into lp.
//      BWD: dL/d(head outputs) of rescale (alpha mean log pi + <dL/da, a>) with dL/da from the Q-networks' input gradients, and the
//      logged sums.  grid = ceil(B / 16), 256 threads = (row, action dim).
struct LaySacActorArgs {
    const float* out; float* dout;          // actor head outputs [B][16] / their gradients [B][FSRL_DOW]
    const float* eps; float* X; float* lp;  // FWD
    const float* DXQ;                       // BWD: [n_q][Bq][Din] input gradients of the Q-networks (unit seed); action columns at Do
    const float* QP;                        // [n_q][B]
    const SacScalars* sc;
    float* statp;                           // [tiles][FB_NSTAT]
    int B, Bq, Do, Da, mode, deterministic, auto_alpha;
    float max_action, cr, cc, rescale, alpha_fixed;
};
__global__ __launch_bounds__(256) void lay_sac_actor_head_kernel(const LaySacActorArgs a) {
    __shared__ float stl[16 * 4];
    const int tid = threadIdx.x, i = tid >> 4, d = tid & 15, lane = tid & 63;
    const int r = blockIdx.x * 16 + i, Da = a.Da, Din = a.Do + Da;
    const bool valid = r < a.B;
    const float invB = 1.0f / (float)a.B;
    const float* o = a.out + (size_t)(valid ? r : 0) * FSRL_MAX_ACT;
    const float alpha = a.auto_alpha ? a.sc->alpha : a.alpha_fixed;
    float lpd = 0.0f, act = 0.0f, sig = 1.0f, ep = 0.0f, one_m = 1.0f, pass = 0.0f, logp = 0.0f;
    float g0 = 0.0f, g1 = 0.0f;
    if (a.deterministic) {
        float th = 0.0f;
        if (valid && d < Da) th = tanhf(o[d]);
        if (a.mode == SAC_A_FWD) {
            if (valid && d < Da) a.X[(size_t)r * Din + a.Do + d] = a.max_action * th;
        } else if (valid && d < Da) {
            const float ga = (a.cr * invB) * a.DXQ[((size_t)0 * a.Bq + r) * Din + a.Do + d] +
                             (a.cc * invB) * a.DXQ[((size_t)1 * a.Bq + r) * Din + a.Do + d];
            g0 = ga * a.max_action * (1.0f - th * th);
        }
    } else {
        if (valid && d < Da) {
            const float mu = o[d], lraw = o[Da + d];
            pass = (lraw >= SAC_LOG_SIG_MIN && lraw <= SAC_LOG_SIG_MAX) ? 1.0f : 0.0f;
            sig = expf(fminf(fmaxf(lraw, SAC_LOG_SIG_MIN), SAC_LOG_SIG_MAX));
            ep = a.eps[(size_t)r * Da + d];
            const float u = mu + ep * sig;
            const float dv = u - mu;
            act = tanhf(u);
            one_m = 1.0f - act * act;
            lpd = (-(dv * dv) / (2.0f * (sig * sig)) - logf(sig) - LOG_SQRT_2PI) - logf(one_m + SAC_F32_EPS);
        }
        for (int dd = 0; dd < Da; ++dd) logp += __shfl(lpd, (lane & 48) + dd, 64);
        if (a.mode == SAC_A_FWD) {
            if (valid && d < Da) a.X[(size_t)r * Din + a.Do + d] = act;
            if (valid && d == 0) a.lp[r] = logp;
        } else if (valid && d < Da) {
            float ga = 0.0f;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {       // torch's tie rule for min: the smaller one gets the gradient, equal values share it
                const float q1 = a.QP[(size_t)(2 * pr) * a.B + r], q2 = a.QP[(size_t)(2 * pr + 1) * a.B + r];
                const float w1 = (q1 < q2) ? 1.0f : (q1 == q2 ? 0.5f : 0.0f), w2 = 1.0f - w1;
                const float cw = (pr == 0 ? a.cr : a.cc) * invB;
                ga += (w1 * cw) * a.DXQ[((size_t)(2 * pr) * a.Bq + r) * Din + a.Do + d];
                ga += (w2 * cw) * a.DXQ[((size_t)(2 * pr + 1) * a.Bq + r) * Din + a.Do + d];
            }
            const float c = a.rescale * alpha * invB;
            const float sq = 2.0f * act * one_m / (one_m + SAC_F32_EPS);
            const float dLdu = c * sq + ga * one_m;
            g0 = dLdu;                                   // d / d mu
            g1 = (dLdu * ep * sig - c) * pass;           // d / d (raw log sigma), head column Da + d
        }
    }
    if (a.mode == SAC_A_BWD && valid) {                  // head gradient rows: [mu (Da) | raw log sigma (Da) | zeros]
        float* dO = a.dout + (size_t)r * FSRL_DOW;
        dO[d] = 0.0f; dO[16 + d] = 0.0f;
    }
    __syncthreads();
    if (a.mode == SAC_A_BWD && valid && d < Da) {
        float* dO = a.dout + (size_t)r * FSRL_DOW;
        dO[d] = g0;
        if (!a.deterministic) dO[Da + d] = g1;
    }
    if (d == 0) {
        stl[i * 4] = valid ? logp : 0.0f;
        stl[i * 4 + 1] = 0.0f; stl[i * 4 + 2] = 0.0f;
        if (a.mode == SAC_A_BWD && valid) {
            if (a.deterministic) { stl[i * 4 + 1] = a.QP[r]; stl[i * 4 + 2] = a.QP[(size_t)a.B + r]; }
            else { stl[i * 4 + 1] = fminf(a.QP[r], a.QP[(size_t)a.B + r]); stl[i * 4 + 2] = fminf(a.QP[(size_t)2 * a.B + r], a.QP[(size_t)3 * a.B + r]); }
        }
    }
    __syncthreads();
    if (tid < 3) {
        float t = 0.0f;
        for (int rr = 0; rr < 16; ++rr) t += stl[rr * 4 + tid];
        a.statp[(size_t)blockIdx.x * FB_NSTAT + tid] = t;
    }
}

// ---- Q heads: FB_MODE_Q_FWD (write Q), _Q_TRAIN (td = Q - y, dout = 2 td / B, sum td^2), _Q_DIN (write Q, unit seed).
//      grid = (ceil(B / 16), n_q), 64 threads: lane r < 16 = row.
struct LaySacQArgs {
    const float* out; float* dout;          // [n_q][mbp][16] / [n_q][mbp][FSRL_DOW]
    const float* tgt; float* qout; float* statp;
    int B, mbp, n_q, mode, pair_shift;
};
__global__ __launch_bounds__(64) void lay_sac_q_head_kernel(const LaySacQArgs a) {
    const int net = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const int r = tile * 16 + tid;
    float st0 = 0.0f;
    if (tid < 16 && r < a.B) {
        const float qv = a.out[((size_t)net * a.mbp + r) * FSRL_MAX_ACT];
        float* dO = a.dout + ((size_t)net * a.mbp + r) * FSRL_DOW;
        if (a.mode == FB_MODE_Q_TRAIN) {
            const float td = qv - a.tgt[(size_t)(net >> a.pair_shift) * a.B + r];
            for (int e = 0; e < FSRL_DOW; ++e) dO[e] = 0.0f;
            dO[0] = 2.0f * td * (1.0f / (float)a.B);
            st0 = td * td;
            a.qout[(size_t)net * a.B + r] = qv;
        } else if (a.mode == FB_MODE_Q_FWD) {
            a.qout[(size_t)net * a.B + r] = qv;
        } else {
            for (int e = 0; e < FSRL_DOW; ++e) dO[e] = 0.0f;
            dO[0] = 1.0f;
            a.qout[(size_t)net * a.B + r] = qv;
        }
    }
    // rows of the tile summed in ascending order
    float t = 0.0f;
    for (int rr = 0; rr < 16; ++rr) t += __shfl(st0, rr, 64);
    if (tid < FB_NSTAT) a.statp[((size_t)tile * a.n_q + net) * FB_NSTAT + tid] = (tid == 0) ? t : 0.0f;
}

// ---- the collector's actor on a layered replay context: raw head outputs to pinned host memory + completion words
__global__ __launch_bounds__(64) void lay_raw_out_kernel(const float* __restrict__ out, float* __restrict__ raw_out, const int cols,
                                                        const int N, unsigned* done, const unsigned seq) {
    const int r = blockIdx.x * 16 + (threadIdx.x >> 2);
    if (r < N)
        for (int o = threadIdx.x & 3; o < cols; o += 4) raw_out[(size_t)r * cols + o] = out[(size_t)r * FSRL_MAX_ACT + o];
    if (done) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done + blockIdx.x, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- CVPO actor head (the head of cvpo_actor_tile_kernel, kernels_cvpo.hpp; cvpo.py:206-222, 319-417) on the head outputs of a
//      layered actor: TARGET (a' at s_{t+n}), PARTICLES (actor_old at s_t: mu_old, std_old, K particles and their observation
//      columns), MFWD (per-tile sums of w log-lik, KL_mu, KL_std, entropy), MBWD (the same forward + the head gradient rows).
//      grid = ceil(B / 16), 256 threads = (row, action dim).
struct LayCvpoActorArgs {
    const float* out; float* dout;          // [B][16] / [B][FSRL_DOW]
    const float* obs; const float* eps; float* X;
    float* mu_old; float* std_old;
    const float* W; const float* XK;
    const CvpoScalars* sc;
    float* statp;
    int B, K, Do, Da, mode;
    float max_action;
};
__global__ __launch_bounds__(256) void lay_cvpo_actor_head_kernel(const LayCvpoActorArgs a) {
    __shared__ float stl[16 * 4];
    const int tid = threadIdx.x, i = tid >> 4, d = tid & 15, lane = tid & 63;
    const int row0 = blockIdx.x * 16, r = row0 + i, Do = a.Do, Da = a.Da, Din = Do + Da;
    const int n_valid = max(0, min(16, a.B - row0));
    const bool valid = i < n_valid, on = valid && d < Da;
    if (a.mode == CVPO_A_PARTICLES) {               // observation columns of the K replicated rows
        const int per = n_valid * Do;
        for (int e = tid; e < a.K * per; e += 256) {
            const int k = e / per, w = e - k * per;
            const int ii = w / Do, f = w - ii * Do;
            a.X[((size_t)k * a.B + row0 + ii) * Din + f] = a.obs[(size_t)(row0 + ii) * Do + f];
        }
    }
    const float* o = a.out + (size_t)(valid ? r : 0) * FSRL_MAX_ACT;
    float th = 0.0f, mu = 0.0f, sig = 1.0f, pass = 0.0f;
    if (on) {
        th = tanhf(o[d]);
        mu = a.max_action * th;
        const float lraw = o[Da + d];
        pass = (lraw >= SAC_LOG_SIG_MIN && lraw <= SAC_LOG_SIG_MAX) ? 1.0f : 0.0f;
        sig = expf(fminf(fmaxf(lraw, SAC_LOG_SIG_MIN), SAC_LOG_SIG_MAX));
    }
    if (a.mode == CVPO_A_TARGET) {
        if (on) a.X[(size_t)r * Din + Do + d] = a.eps[(size_t)r * Da + d] * sig + mu;
        return;
    }
    if (a.mode == CVPO_A_PARTICLES) {
        if (on) {
            a.mu_old[(size_t)r * Da + d] = mu;
            a.std_old[(size_t)r * Da + d] = sig;
            for (int k = 0; k < a.K; ++k) {
                const size_t rk = (size_t)k * a.B + r;
                a.X[rk * Din + Do + d] = a.eps[rk * Da + d] * sig + mu;      // Normal.sample: eps * std + mean
            }
        }
        return;
    }
    // ---- M-step row terms (cvpo.py:378-417), per action dimension then summed over d
    float mle = 0.0f, klm = 0.0f, kls = 0.0f, ent = 0.0f, g0 = 0.0f, g1 = 0.0f;
    if (on) {
        const float mu_o = a.mu_old[(size_t)r * Da + d], sd_o = a.std_old[(size_t)r * Da + d];
        const float var_o = sd_o * sd_o, var = sig * sig;
        const float lso = logf(sd_o), ls = logf(sig);
        float s_w = 0.0f, s_wdm = 0.0f, s_wdo2 = 0.0f;
        for (int k = 0; k < a.K; ++k) {
            const size_t rk = (size_t)k * a.B + r;
            const float w = a.W[rk], ak = a.XK[rk * Din + Do + d];
            const float dm = ak - mu, dmo = ak - mu_o;
            const float ll = (-(dm * dm) / (2.0f * var_o) - lso - LOG_SQRT_2PI) + (-(dmo * dmo) / (2.0f * var) - ls - LOG_SQRT_2PI);
            mle = fmaf(w, ll, mle);
            s_w += w; s_wdm = fmaf(w, dm, s_wdm); s_wdo2 = fmaf(w, dmo * dmo, s_wdo2);
        }
        const float var_oc = fmaxf(var_o, 1e-6f), var_c = fmaxf(var, 1e-6f);
        const float dmu = mu_o - mu;
        klm = 0.5f * (dmu * dmu) / var_oc;
        kls = 0.5f * (logf(var_c / var_oc) + var_oc / var_c - 1.0f);
        ent = (0.5f + 0.5f * 1.8378770664093453f + lso) + (0.5f + 0.5f * 1.8378770664093453f + ls);
        if (a.mode == CVPO_A_MBWD) {
            const float invB = 1.0f / (float)a.B, invKB = invB / (float)a.K;
            const float dual_mu = a.sc->dual_mu, dual_std = a.sc->dual_std;
            float g_mu = -invKB * (s_wdm / var_o);
            float g_sg = -invKB * (s_wdo2 / (var * sig) - s_w / sig);
            g_mu += dual_mu * invB * (-dmu / var_oc);
            if (var > 1e-6f) g_sg += dual_std * invB * (1.0f / sig - var_oc / (var * sig));
            g0 = g_mu * a.max_action * (1.0f - th * th);
            g1 = g_sg * sig * pass;
        }
    }
    if (a.mode == CVPO_A_MBWD) {
        if (valid) { float* dO = a.dout + (size_t)r * FSRL_DOW; dO[d] = 0.0f; dO[16 + d] = 0.0f; }
        __syncthreads();
        if (on) { float* dO = a.dout + (size_t)r * FSRL_DOW; dO[d] = g0; dO[Da + d] = g1; }
        return;
    }
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    for (int dd = 0; dd < Da; ++dd) {
        const int src = (lane & 48) + dd;
        s0 += __shfl(mle, src, 64); s1 += __shfl(klm, src, 64);
        s2 += __shfl(kls, src, 64); s3 += __shfl(ent, src, 64);
    }
    if (d == 0) {
        stl[i * 4 + 0] = valid ? s0 : 0.0f; stl[i * 4 + 1] = valid ? s1 : 0.0f;
        stl[i * 4 + 2] = valid ? s2 : 0.0f; stl[i * 4 + 3] = valid ? s3 : 0.0f;
    }
    __syncthreads();
    if (tid < 4) {
        float t = 0.0f;
        for (int rr = 0; rr < 16; ++rr) t += stl[rr * 4 + tid];
        a.statp[(size_t)blockIdx.x * FB_NSTAT + tid] = t;
    }
}
