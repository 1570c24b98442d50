// kernels_sac.hpp -- SAC-Lagrangian update kernels (fsrl/policy/sac_lag.py:136-269,
// fsrl/policy/base_policy.py:453-512,543-567).  The Q-networks run through fb_tile_kernel's
// Q modes (kernels_fb.hpp); this file holds the replay gather, the tanh-Gaussian actor tile, the
// float64 n-step target, the scalar bookkeeping (alpha, logged stats) and the Polyak update.
#pragma once
#include "kernels_fb.hpp"
#include "kernels_sample.hpp"

#define SAC_LOG_SIG_MIN (-20.0f)
#define SAC_LOG_SIG_MAX (2.0f)
#define SAC_F32_EPS 1.1920928955078125e-07f   // np.finfo(np.float32).eps  (sac_lag.py:118)
#define FSRL_SAC_NSTATS_K 10

// (SacGatherArgs, the sampler's structs and device functions: kernels_sample.hpp)
__global__ void sac_gather_kernel(const SacGatherArgs a) {
    const int Din = a.Do + a.Da;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < a.B * Din; e += gridDim.x * blockDim.x) {
        const int r = e / Din, f = e - r * Din;
        const size_t s = (size_t)a.idx[r], t = (size_t)a.term[r];
        if (f < a.Do) {
            const float o = a.st.obs[s * a.Do + f], on = a.st.obs_next[t * a.Do + f];
            a.XQ[e] = o; a.XP[e] = o; a.XN[e] = on;
            a.OBS[(size_t)r * a.Do + f] = o; a.OBSN[(size_t)r * a.Do + f] = on;
        } else {
            a.XQ[e] = a.st.act[s * a.Da + (f - a.Do)];
        }
    }
}

__global__ __launch_bounds__(256) void sac_sample_kernel(const SacSampleArgs a) {
    // SAC / DDPG: one thread per sampled row.  CVPO (eps_k != NULL): B * K threads, thread (b, kp) also draws particle
    // kp's noise for row b; the row work is done by the kp == 0 threads.
    // the sub-buffers' bookkeeping goes to LDS first: the row's sub-buffer is found by a linear scan over it and the n-step chain
    // looks it up again per step -- from global memory every probe was a dependent round trip (7 us for this kernel)
    constexpr int BOOK_LDS = 512;
    __shared__ SacBook book_s[BOOK_LDS];
    const bool in_lds = a.env_num <= BOOK_LDS;
    if (in_lds) {
        for (int e = threadIdx.x; e < a.env_num; e += blockDim.x) book_s[e] = a.book[e];
        __syncthreads();
    }
    const SacBook* __restrict__ book = in_lds ? book_s : a.book;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = a.eps_k ? t % a.B : t, kp = a.eps_k ? t / a.B : 0;
    if (b >= a.B || kp >= max(a.K, 1)) return;
    const uint32_t k0 = (uint32_t)a.key, k1 = (uint32_t)(a.key >> 32);
    if (a.eps_k) {
        for (int d0 = 0; d0 < a.Da; d0 += 4) {             // draws 0x100.. : four normals per Philox block
            uint32_t r[4] = {(uint32_t)b, 0x100u + (uint32_t)(kp * 4 + (d0 >> 2)), (uint32_t)a.counter, (uint32_t)(a.counter >> 32)};
            philox4x32_10(r, k0, k1);
            float v[4];
            box_muller(r[0], r[1], v[0], v[1]);
            box_muller(r[2], r[3], v[2], v[3]);
            float* o = a.eps_k + ((size_t)kp * a.B + b) * a.Da + d0;
            for (int j = 0; j < 4 && d0 + j < a.Da; ++j) o[j] = v[j];
        }
        if (kp != 0) return;
    }
    sac_sample_row(a, book, b, k0, k1);
}

// ---- CVPO, library RNG: sample + gather + the K particles' noise in ONE launch (r6; sac_sample_kernel + sac_gather_kernel before).
//      Blocks [0, ceil(B / SG_ROWS)): sac_sample_gather_block; the blocks behind them: thread (b, kp) draws particle kp's noise for
//      row b -- the Philox counters of sac_sample_kernel's CVPO branch, so the same values.
__global__ __launch_bounds__(256) void cvpo_sample_gather_kernel(const SacSampleArgs a, const SacGatherArgs g) {
    const int nb = (a.B + SG_ROWS - 1) / SG_ROWS;
    if ((int)blockIdx.x < nb) { sac_sample_gather_block(a, g, blockIdx.x); return; }
    const int t = ((int)blockIdx.x - nb) * 256 + threadIdx.x;
    const int b = t % a.B, kp = t / a.B;
    if (kp >= a.K) return;
    const uint32_t k0 = (uint32_t)a.key, k1 = (uint32_t)(a.key >> 32);
    for (int d0 = 0; d0 < a.Da; d0 += 4) {             // draws 0x100.. : four normals per Philox block
        uint32_t r[4] = {(uint32_t)b, 0x100u + (uint32_t)(kp * 4 + (d0 >> 2)), (uint32_t)a.counter, (uint32_t)(a.counter >> 32)};
        philox4x32_10(r, k0, k1);
        float v[4];
        box_muller(r[0], r[1], v[0], v[1]);
        box_muller(r[2], r[3], v[2], v[3]);
        float* o = a.eps_k + ((size_t)kp * a.B + b) * a.Da + d0;
        for (int j = 0; j < 4 && d0 + j < a.Da; ++j) o[j] = v[j];
    }
}

// ---- sample + gather in ONE launch (SAC / DDPG-Lag, library RNG): sac_sample_gather_block (kernels_sample.hpp) per workgroup.
//      Two dependent ~5-7 us launches at their floors become one.
__global__ __launch_bounds__(256) void sac_sample_gather_kernel(const SacSampleArgs a, const SacGatherArgs g) {
    sac_sample_gather_block(a, g, blockIdx.x);
}

// ---- actor tile: a = tanh(mu + sigma*eps), log pi with the tanh correction; optional backward
#define SAC_A_FWD 0      // write action into X[:, Do:], log pi into lp_out
#define SAC_A_BWD 1      // forward again + gradient of rescale*(alpha*mean(log pi) + <dL/da, a>)
struct SacActorArgs {
    const float* obs;    // [B][Do]
    const float* eps;    // [B][Da] standard-normal draws (rsample)
    float* X;            // [B][Do+Da]: action columns written in FWD
    float* lp_out;       // [B]
    const float* DA;     // [4][B][Da] dQ_n/da of the four Q-nets, unit seed (BWD)
    const float* QP;     // [4][B] Q_n(s, a_pi): min routing + the logged min-Q sums (BWD)
    float cr, cc;        // loss weights of min(Qr1,Qr2) and min(Qc1,Qc2): -rescale, rescale*lambda
    int deterministic;   // DDPG-Lag actor: a = max_action * tanh(out), single critics, no entropy (ddpg_lag.py:189-213)
    float max_action;
    const SacScalars* sc;
    float* A1; float* A2; float* D1; float* D2; float* DO;   // side buffers (BWD)
    float* statp;        // [n_tiles][FB_NSTAT]  st[0] = sum log pi
    int B, mode;
    float rescale;       // lagrangian rescaling 1/(sum(lambda)+1)
    int auto_alpha; float alpha_fixed;
    // FWD only: a second batch in the same launch (tiles_half > 0): workgroups [tiles_half, 2 * tiles_half) evaluate the
    // actor P2 on obs2 / eps2 into X2 / lp2.  One launch then serves both a' ~ pi(s_{t+n}) for the targets and a ~ pi(s_t)
    // for the actor step -- neither depends on the critic update in between.
    const float* P2; const float* obs2; const float* eps2; float* X2; float* lp2; int tiles_half;
    // r5, FWD of an update with the library's RNG: every workgroup first DRAWS its rows (sac_sample_row: the Philox counters of
    // sac_sample_gather_kernel, so the same sample) and gathers them -- the first half obs_next at the chain's end into XN / OBSN,
    // the second half obs (+ the stored action) into XQ / XP / OBS -- then reads them back as its tile.  One launch less per update.
    int sg_on;
    SacSampleArgs sa;
    SacGatherArgs ga;
    int probe;           // -DFSRL_PROBES builds: FWD launches return early -- 1 at entry, 2 behind the prologue, 3 behind the forward pass
};

template <int H, int R>
__global__ __launch_bounds__(4 * H) void sac_actor_tile_kernel(const float* __restrict__ P_,
                                                              const ModelDesc md, const SacActorArgs a) {
    __shared__ TileSmem<H> sm;
    constexpr int NT = TileGeom<H>::NT;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
    const bool second = a.tiles_half > 0 && (int)blockIdx.x >= a.tiles_half;
    const float* __restrict__ Pn = second ? a.P2 : P_;
    const float* __restrict__ obs_ = second ? a.obs2 : a.obs;
    const float* __restrict__ eps_ = second ? a.eps2 : a.eps;
    float* __restrict__ X_ = second ? a.X2 : a.X;
    float* __restrict__ lp_ = second ? a.lp2 : a.lp_out;
    const int row0 = (second ? (int)blockIdx.x - a.tiles_half : (int)blockIdx.x) * R;
    const NetOff no = md.net[0];
    const int Do = md.Do, Da = md.Da;
    const int n_valid = max(0, min(R, a.B - row0));
    const float invB = 1.0f / (float)a.B;
#ifdef FSRL_PROBES
    if (a.probe == 1 && a.mode == SAC_A_FWD) return;
#endif

    TileStage<H> stg;
    if (a.sg_on) {
        constexpr int BOOK_LDS = 512;
        __shared__ SacBook book_s[BOOK_LDS];
        __shared__ int idx_s[16], term_s[16];
        const bool in_lds = a.sa.env_num <= BOOK_LDS;
        if (in_lds) {
            for (int e = tid; e < a.sa.env_num; e += NT) book_s[e] = a.sa.book[e];
            __syncthreads();
        }
        const SacBook* __restrict__ book = in_lds ? book_s : a.sa.book;
        if (tid < 16) { idx_s[tid] = 0; term_s[tid] = 0; }
        // both halves draw their rows: the same counters, the same values, written twice.  Index + chain: one thread per row in wave 0;
        // the noise: one thread per (row, pair of action dimensions) from wave 1 on
        const int npair = (Da + 1) >> 1;
        const bool sampler = tid < R && row0 + tid < a.B;
        unsigned char f_last = 0;
        bool head_last = false;
        if (sampler) {
            int idx, term;
            sac_sample_index(a.sa, book, row0 + tid, (uint32_t)a.sa.key, (uint32_t)(a.sa.key >> 32), idx, term, f_last, head_last);
            idx_s[tid] = idx; term_s[tid] = term;
        } else if (tid >= 64 && tid < 64 + R * npair) {
            const int t = tid - 64, rl = t / npair, pp = t - rl * npair;
            if (row0 + rl < a.B) sac_sample_noise(a.sa, row0 + rl, 2 * pp, (uint32_t)a.sa.key, (uint32_t)(a.sa.key >> 32));
        }
        __syncthreads();
        if (sampler) sac_sample_end_last(a.sa, row0 + tid, f_last, head_last);     // its flag byte may still be in flight at the barrier
        // r6: the gathered rows go STRAIGHT into the stage registers (element e = row * Do + k: the stage's own mapping) and to the
        // batch arrays the later launches read; until r5 they were stored, waited for and read back (two more round trips)
        const SacGatherArgs& g = a.ga;
        const int Din = Do + Da, nx = n_valid * Do;
        const unsigned magic = div_magic(Do);
        const float* __restrict__ src = second ? g.st.obs : g.st.obs_next;
        const int* __restrict__ row_s = second ? idx_s : term_s;
#pragma unroll
        for (int u = 0; u < TileStage<H>::NX; ++u) {
            const int e = tid + u * NT, ec = min(e, max(nx - 1, 0));
            const int i = div_by_magic((unsigned)ec, magic), k = ec - i * Do;
            const float v = src[(size_t)row_s[i] * Do + k];
            stg.xv[u] = (e < nx) ? v : 0.0f;
            if (e < nx) {
                const size_t r = (size_t)(row0 + i);
                if (second) { g.XQ[r * Din + k] = v; g.XP[r * Din + k] = v; g.OBS[r * Do + k] = v; }
                else { g.XN[r * Din + k] = v; g.OBSN[r * Do + k] = v; }
            }
        }
        if (second) {
            for (int e = tid; e < n_valid * Da; e += NT) {
                const int rl = e / Da, f = e - rl * Da;
                g.XQ[(size_t)(row0 + rl) * Din + Do + f] = g.st.act[(size_t)idx_s[rl] * Da + f];
            }
        }
#pragma unroll
        for (int u = 0; u < TileStage<H>::NR; ++u) stg.rdv[u] = 0.0f;
        stg.issue_params(Pn, no, Do, 0, tid);
    } else {
        stg.issue(Pn, no, Do, 0, obs_ + (size_t)row0 * Do, nullptr, n_valid, tid);
    }
    FwdW2Frag<H> wf;
    wf.load(Pn + no.W2f, wave, lane);
    for (int e = tid; e < 16 * FSRL_DOW; e += NT) sm.dout[e] = 0.0f;
    stg.commit(sm, no, Do, tid);
    __syncthreads();
#ifdef FSRL_PROBES
    if (a.probe == 2 && a.mode == SAC_A_FWD) return;
#endif
    tile_forward<H, R, true>(sm, Pn, no, Do, tid, wf);     // sm.out[i][0..Da) = mu, [Da..2Da) = raw log sigma
#ifdef FSRL_PROBES
    if (a.probe == 3 && a.mode == SAC_A_FWD) return;
#endif

    float wb[H / 16][4];
    if (a.mode == SAC_A_BWD) {
        const float* __restrict__ W2c = Pn + no.W2 + wave * 16 + li;
#pragma unroll
        for (int jc = 0; jc < H / 16; ++jc) {
#pragma unroll
            for (int s = 0; s < 4; ++s) wb[jc][s] = W2c[(size_t)(16 * jc + 4 * q + s) * H];
        }
    }
    const float alpha = a.auto_alpha ? a.sc->alpha : a.alpha_fixed;
    if (tid < 16 * R) {
        const int i = tid >> 4, d = tid & 15;
        const int r = row0 + i;
        const bool valid = i < n_valid;
        float lpd = 0.0f, act = 0.0f, sig = 1.0f, ep = 0.0f, one_m = 1.0f, pass = 0.0f;
        float logp = 0.0f;
        if (a.deterministic) {
            float th = 0.0f;
            if (valid && d < Da) th = tanhf(sm.out[i * FSRL_MAX_ACT + d]);
            if (a.mode == SAC_A_FWD) {
                if (valid && d < Da) X_[(size_t)r * (Do + Da) + Do + d] = a.max_action * th;
            } else if (valid && d < Da) {
                // dL/da_d = (cr * dQ_r/da_d + cc * dQ_c/da_d) / B ; a_d = max_action * tanh(out_d)
                const float ga = (a.cr * invB) * a.DA[((size_t)0 * a.B + r) * Da + d] +
                                 (a.cc * invB) * a.DA[((size_t)1 * a.B + r) * Da + d];
                sm.dout[i * FSRL_DOW + d] = ga * a.max_action * (1.0f - th * th);
            }
        } else {
        if (valid && d < Da) {
            const float mu = sm.out[i * FSRL_MAX_ACT + d];
            const float lraw = sm.out[i * FSRL_MAX_ACT + Da + d];
            pass = (lraw >= SAC_LOG_SIG_MIN && lraw <= SAC_LOG_SIG_MAX) ? 1.0f : 0.0f;
            sig = expf(fminf(fmaxf(lraw, SAC_LOG_SIG_MIN), SAC_LOG_SIG_MAX));
            ep = eps_[(size_t)r * Da + d];
            const float u = mu + ep * sig;
            const float dv = u - mu;
            act = tanhf(u);
            one_m = 1.0f - act * act;
            lpd = (-(dv * dv) / (2.0f * (sig * sig)) - logf(sig) - LOG_SQRT_2PI) - logf(one_m + SAC_F32_EPS);
        }
        // log pi = sum_d Normal.log_prob  -  sum_d log(1 - a^2 + eps): two separate sums in the
        // reference; summed per dim here (difference: fp32 rounding only)
        for (int dd = 0; dd < Da; ++dd) logp += __shfl(lpd, (lane & 48) + dd, 64);
        if (a.mode == SAC_A_FWD) {
            if (valid && d < Da) X_[(size_t)r * (Do + Da) + Do + d] = act;
            if (valid && d == 0) lp_[r] = logp;
        } else if (valid && d < Da) {
            // dL/da_d = sum over the two double critics of weight * d min(Q1,Q2)/da_d / B ; torch's
            // tie rule for min: the smaller one gets the gradient, equal values share it
            float ga = 0.0f;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const float q1 = a.QP[(size_t)(2 * pr) * a.B + r], q2 = a.QP[(size_t)(2 * pr + 1) * a.B + r];
                const float w1 = (q1 < q2) ? 1.0f : (q1 == q2 ? 0.5f : 0.0f), w2 = 1.0f - w1;
                const float cw = (pr == 0 ? a.cr : a.cc) * invB;
                ga += (w1 * cw) * a.DA[((size_t)(2 * pr) * a.B + r) * Da + d];
                ga += (w2 * cw) * a.DA[((size_t)(2 * pr + 1) * a.B + r) * Da + d];
            }
            const float c = a.rescale * alpha * invB;                 // weight of log pi in the loss
            const float sq = 2.0f * act * one_m / (one_m + SAC_F32_EPS);   // d(-log(1-a^2+eps))/du
            const float dLdu = c * sq + ga * one_m;                   // (+-(u-mu)/sigma^2 cancel)
            sm.dout[i * FSRL_DOW + d] = dLdu;                         // d/dmu
            sm.dout[i * FSRL_DOW + Da + d] = (dLdu * ep * sig - c) * pass;   // d/d(raw log sigma)
        }
        }
        if (d == 0) {
            sm.w1[i * FB_NSTAT] = valid ? logp : 0.0f;
            if (a.mode == SAC_A_BWD && a.deterministic) {          // sums of Q_r, Q_c
                sm.w1[i * FB_NSTAT + 1] = valid ? a.QP[r] : 0.0f;
                sm.w1[i * FB_NSTAT + 2] = valid ? a.QP[(size_t)a.B + r] : 0.0f;
            } else if (a.mode == SAC_A_BWD) {       // logged actor losses: sums of min(Q1,Q2) per double critic
                sm.w1[i * FB_NSTAT + 1] = valid ? fminf(a.QP[r], a.QP[(size_t)a.B + r]) : 0.0f;
                sm.w1[i * FB_NSTAT + 2] = valid ? fminf(a.QP[(size_t)2 * a.B + r], a.QP[(size_t)3 * a.B + r]) : 0.0f;
            }
        }
    }
    __syncthreads();
    if (tid < 3) {
        float t = 0.0f;
        if (tid == 0 || a.mode == SAC_A_BWD)
            for (int i = 0; i < R; ++i) t += sm.w1[i * FB_NSTAT + tid];
        a.statp[(size_t)blockIdx.x * FB_NSTAT + tid] = t;
    }
    if (a.mode != SAC_A_BWD) return;
    tile_backward<H, R>(sm, no, wb, a.A1 + (size_t)row0 * H, a.A2 + (size_t)row0 * H, a.D1 + (size_t)row0 * H,
                     a.D2 + (size_t)row0 * H, a.DO + (size_t)row0 * FSRL_DOW, tid, false);
}

// ---- n-step target (float64), base_policy.py:453-512 + nstep_return :543-567: the stand-alone launch (CVPO; SAC / DDPG-Lag fold
//      sac_nstep_target into the critics' tile launch, kernels_fb.hpp)
__global__ void sac_nstep_kernel(const SacNstepArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    a.Y[b] = sac_nstep_target(a, b, 0);
    a.Y[(size_t)a.B + b] = sac_nstep_target(a, b, 1);
}

// ---- scalar bookkeeping of one update: logged stats, alpha loss + Adam on log_alpha
struct SacFinalArgs {
    const float* statp_q;    // [n_tiles][4][FB_NSTAT]   st0 = sum td^2            (critic launch)
    const float* statp_pi;   // [n_tiles][FB_NSTAT]      st0 = sum log pi, st1 / st2 = sum min(Q1,Q2) reward / cost
    SacScalars* sc;
    float* stats;            // [FSRL_SAC_NSTATS_K]
    int n_tiles_q, n_tiles_pi, B;
    float rescale, lam, target_entropy, alpha_lr, beta1, beta2, adam_eps, alpha_fixed;
    int auto_alpha, use_lagrangian;
    int n_q;                 // Q-networks in statp_q: 4 (two double critics) or 2 (DDPG-Lag: single critics)
};
__device__ __forceinline__ void sac_finalize_row(const SacFinalArgs& a, const int lane) {
    // nine sums over the tiles (4 x td^2, 4 x min-Q, log pi): lanes stride the tiles, then a fixed
    // xor-tree adds the lanes (float64, order independent of scheduling)
    double s9[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) s9[k] = 0.0;
    for (int t = lane; t < a.n_tiles_q; t += 64) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < a.n_q) s9[k] += (double)a.statp_q[((size_t)t * a.n_q + k) * FB_NSTAT];
    }
    for (int t = lane; t < a.n_tiles_pi; t += 64) {
        s9[8] += (double)a.statp_pi[(size_t)t * FB_NSTAT];
        s9[4] += (double)a.statp_pi[(size_t)t * FB_NSTAT + 1];     // sum min(Qr1, Qr2)
        s9[6] += (double)a.statp_pi[(size_t)t * FB_NSTAT + 2];     // sum min(Qc1, Qc2)
    }
    float m9[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) m9[k] = (float)(wave_sum_d(s9[k]) / (double)a.B);
    const bool single = a.n_q == 2;
    const float q_r1 = m9[0], q_r2 = single ? 0.0f : m9[1], q_c1 = single ? m9[1] : m9[2], q_c2 = single ? 0.0f : m9[3];
    const float minqr = m9[4], minqc = m9[6], mlogp = m9[8];
    if (lane == 0) {
        SacScalars sc = *a.sc;
        const float alpha = a.auto_alpha ? sc.alpha : a.alpha_fixed;
        const float q0 = q_r1 + q_r2, q1 = q_c1 + q_c2;
        const float actor_rew = alpha * mlogp - minqr;
        const float actor_safety = a.use_lagrangian ? minqc * a.lam : 0.0f;
        const float actor_total = a.rescale * (actor_rew + actor_safety);
        float alpha_loss = 0.0f, alpha_value = alpha;
        if (a.auto_alpha) {
            const float lpm = mlogp + a.target_entropy;
            alpha_loss = -(sc.log_alpha * lpm);
            const float g = -lpm;                                   // d alpha_loss / d log_alpha
            sc.t += 1;
            sc.m = sc.m + (float)(1.0 - (double)a.beta1) * (g - sc.m);
            sc.v = sc.v * a.beta2;
            sc.v = sc.v + ((float)(1.0 - (double)a.beta2) * g) * g;
            const double bc1 = 1.0 - pow((double)a.beta1, (double)sc.t), bc2 = 1.0 - pow((double)a.beta2, (double)sc.t);
            const float step_size = (float)((double)a.alpha_lr / bc1);
            const float denom = sqrtf(sc.v) / (float)sqrt(bc2) + a.adam_eps;
            sc.log_alpha = sc.log_alpha + (-step_size * sc.m) / denom;
            sc.alpha = expf(sc.log_alpha);
            alpha_value = sc.alpha;
            *a.sc = sc;
        }
        float* o = a.stats;
        o[0] = a.rescale; o[1] = a.lam; o[2] = actor_safety; o[3] = alpha_loss; o[4] = alpha_value;
        o[5] = actor_rew; o[6] = actor_total; o[7] = q0; o[8] = q1; o[9] = q0 + q1;
    }
}

// The last Adam pass of an update with the update's bookkeeping riding along: blocks [0, gridDim.x - 1) step the
// parameters exactly as adam_range_kernel does (one shared adam_element), the extra block writes the logged row
// (FINAL = sac_finalize_row / cvpo_finalize_row) -- one kernel boundary less per update.
template <class FinalArgs, void (*FINAL)(const FinalArgs&, int)>
__global__ __launch_bounds__(256) void adam_final_kernel(float* __restrict__ P, float* __restrict__ M, float* __restrict__ V,
                                                        const float* __restrict__ G, int n, float one_minus_b1, float beta2,
                                                        float one_minus_b2, float step_size, float bc2_sqrt, float eps,
                                                        int nparts, int stride, const ModelDesc md, float* __restrict__ tgt,
                                                        float tau, float one_minus_tau, const FinalArgs fa) {
    if (blockIdx.x == gridDim.x - 1) {
        if (threadIdx.x < 64) FINAL(fa, threadIdx.x);
        return;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float p = P[i];
    float gs = G[i];                                       // split-K partials of fb_wgrad_kernel, z order
    for (int z = 1; z < nparts; ++z) gs += G[(size_t)z * stride + i];
    adam_element(P, M, V, i, p, gs, 1.0f, 0.0f, one_minus_b1, beta2, one_minus_b2, step_size, bc2_sqrt, eps, md, tgt, tau,
                 one_minus_tau);
}
