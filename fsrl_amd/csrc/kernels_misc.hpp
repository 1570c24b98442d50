// kernels_misc.hpp -- the HBM-bound / scalar parts of the update: store scatter + gather,
// float64 GAE scan, per-minibatch advantage statistics, stats finalisation, grad-clip + Adam.
#pragma once
#include "common.hpp"

// ---------------------------------------------------------------- store: staged rows -> slots
// Rows arrive packed (SoA) in a device staging area after one hipMemcpyAsync per array on
// the side stream; this kernel scatters them to their per-env sub-buffer slots
// (tianshou VectorReplayBuffer.add, as used at fsrl/data/fast_collector.py:333-335).
struct StorePtrs {
    float* obs;       // [maxsize][Do]
    float* obs_next;  // [maxsize][Do]
    float* act;       // [maxsize][Da]
    double* rew;      // [maxsize]
    double* cost;     // [maxsize]
    uint8_t* flags;   // [maxsize] bit0 terminated, bit1 truncated
};

// rows arrive as packed records (host: struct Staging): rew f64 | cost f64 | slot i32 | flags u32 | obs[Do] | obs_next[Do] | act[Da]
__global__ void store_scatter_kernel(StorePtrs st, const uint8_t* __restrict__ recs, const int rec, int k, int Do, int Da) {
    const int per = 2 * Do + Da + 1;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < k * per; e += gridDim.x * blockDim.x) {
        const int row = e / per, f = e - row * per;
        const uint8_t* r = recs + (size_t)row * rec;
        const size_t dst = (size_t)*reinterpret_cast<const int*>(r + 16);
        const float* x = reinterpret_cast<const float*>(r + 24);
        if (f < Do) st.obs[dst * Do + f] = x[f];
        else if (f < 2 * Do) st.obs_next[dst * Do + (f - Do)] = x[f];
        else if (f < 2 * Do + Da) st.act[dst * Da + (f - 2 * Do)] = x[f];
        else {
            st.rew[dst] = *reinterpret_cast<const double*>(r);
            st.cost[dst] = *reinterpret_cast<const double*>(r + 8);
            st.flags[dst] = (uint8_t)*reinterpret_cast<const uint32_t*>(r + 20);
        }
    }
}

// ---------------------------------------------------------------- batch = buffer.sample(0)
// Gather the store rows `indices` (env-major, chronological) into contiguous batch arrays;
// bit2 of the batch flags = end_flag (done | unfinished tail, base_policy.py:409-411),
// supplied by the host which owns the episode bookkeeping.
struct BatchPtrs {
    float* obs; float* obs_next; float* act; double* rew; double* cost; uint8_t* flags;
};

__global__ void batch_gather_kernel(StorePtrs st, BatchPtrs b, const int* __restrict__ indices,
                                    const uint8_t* __restrict__ end_flag, int n, int Do, int Da) {
    const int per = 2 * Do + Da + 1;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)n * per;
         e += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(e / per), f = (int)(e - (size_t)row * per);
        const size_t src = (size_t)indices[row];
        if (f < Do) b.obs[(size_t)row * Do + f] = st.obs[src * Do + f];
        else if (f < 2 * Do) b.obs_next[(size_t)row * Do + (f - Do)] = st.obs_next[src * Do + (f - Do)];
        else if (f < 2 * Do + Da) b.act[(size_t)row * Da + (f - 2 * Do)] = st.act[src * Da + (f - 2 * Do)];
        else {
            b.rew[row] = st.rew[src];
            b.cost[row] = st.cost[src];
            b.flags[row] = (uint8_t)((st.flags[src] & 3) | (end_flag[row] ? 4 : 0));
        }
    }
}

// ---------------------------------------------------------------- GAE(lambda), float64
// gae_return (fsrl/policy/base_policy.py:524-540).  The recurrence g_i = delta_i + disc_i*g_{i+1}
// is evaluated strictly sequentially inside one wave per episode segment (segments end where
// end_flag is set, where disc = 0 cuts the dependence), with the multiply and the add rounded
// separately -- bit-identical to the reference's scan.  Loads are coalesced 64 at a time.
struct GaeArgs {
    const float* values;   // [C][N]
    const float* vnext;    // [C][N]  already * ~terminated
    const double* rew;     // [N]
    const double* cost;    // [N]
    const uint8_t* flags;  // [N] bit2 = end_flag
    const int* seg_start;  // [n_seg+1]
    float* advs;           // [C][N] float32 (to_torch_as, base_policy.py:445-446)
    float* rets;           // [C][N]
    double* adv64;         // optional [C][N]
    int N;
    double gamma, gl;      // gl = gamma*lambda
    // reward_normalization (base_policy.py:430-444): rms = [C][3] running (mean, var, count) of the normalised returns.  Values
    // are un-normalised by sqrt(var + 1e-8) before the scan, returns divided by it after; ret64 keeps the float64 returns the
    // running statistics are then updated with (ret_rms_update_kernel).  Null = off.
    const double* rms;
    double* ret64;         // [C][N]
};
#define FSRL_RMS_EPS 1e-8   // BasePolicy._eps

__device__ __forceinline__ double readlane_f64(double x, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64) void gae_kernel(const GaeArgs a) {
#pragma clang fp contract(off)
    const int seg = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
    const int s = a.seg_start[seg], e = a.seg_start[seg + 1];
    const float* __restrict__ v = a.values + (size_t)c * a.N;
    const float* __restrict__ vn = a.vnext + (size_t)c * a.N;
    const double* __restrict__ met = (c == 0) ? a.rew : a.cost;
    // numpy: float32 array * float64 scalar (np.sqrt(var + eps)) -> float64 products (NumPy 2 promotion; NumPy 1's value-based
    // casting would round them to float32 -- a 6e-8 relative difference, inside every tolerance downstream)
    const double scale = a.rms ? sqrt(a.rms[3 * c + 1] + FSRL_RMS_EPS) : 1.0;
    double g = 0.0;
    for (int base = e - 64; base > s - 64; base -= 64) {
        const int idx = base + lane;
        const bool ok = idx >= s;
        double delta = 0.0, disc = 0.0, vv = 0.0;
        if (ok) {
            vv = (double)v[idx];
            double vnx = (double)vn[idx];
            if (a.rms) { vv = vv * scale; vnx = vnx * scale; }
            const double t0 = vnx * a.gamma;
            const double t1 = met[idx] + t0;
            delta = t1 - vv;
            disc = (1.0 - ((a.flags[idx] & 4) ? 1.0 : 0.0)) * a.gl;
        }
        double res = 0.0;
#pragma unroll
        for (int t = 63; t >= 0; --t) {
            const double dt = readlane_f64(delta, t);
            const double ct = readlane_f64(disc, t);
            const double prod = ct * g;
            g = dt + prod;
            if (lane == t) res = g;
        }
        if (ok) {
            const size_t o = (size_t)c * a.N + idx;
            a.advs[o] = (float)res;
            double ret = res + vv;
            if (a.rms) { ret = ret / scale; a.ret64[o] = ret; }
            a.rets[o] = (float)ret;
            if (a.adv64) a.adv64[o] = res;
        }
    }
}

// RunningMeanStd.update(ret) of tianshou 0.5 (utils/statistics.py:89-103) for every critic: batch mean and (biased) variance
// in float64, two passes like np.var, then the parallel-variance merge.  One block per critic; runs after gae_kernel on the
// same stream, so the scan above has already read the previous variance.
__global__ __launch_bounds__(1024) void ret_rms_update_kernel(const double* __restrict__ ret64, double* __restrict__ rms,
                                                              const int N) {
    __shared__ double sh[16];
    __shared__ double bc;
    const int c = blockIdx.x, tid = threadIdx.x;
    const double* __restrict__ x = ret64 + (size_t)c * N;
    auto block_sum = [&](double v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((tid & 63) == 0) sh[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) { double t = 0.0; for (int w = 0; w < 16; ++w) t += sh[w]; bc = t; }
        __syncthreads();
        const double r = bc;
        __syncthreads();
        return r;
    };
    double s = 0.0;
    for (int i = tid; i < N; i += 1024) s += x[i];
    const double mean = block_sum(s) / (double)N;
    double q = 0.0;
    for (int i = tid; i < N; i += 1024) { const double d = x[i] - mean; q += d * d; }
    const double var = block_sum(q) / (double)N;
    if (tid == 0) {
#pragma clang fp contract(off)
        const double m0 = rms[3 * c], v0 = rms[3 * c + 1], n0 = rms[3 * c + 2];
        const double nb = (double)N, delta = mean - m0, tot = n0 + nb;
        const double new_mean = m0 + delta * nb / tot;
        const double m_a = v0 * n0, m_b = var * nb;
        const double m_2 = m_a + m_b + delta * delta * n0 * nb / tot;
        rms[3 * c] = new_mean; rms[3 * c + 1] = m_2 / tot; rms[3 * c + 2] = tot;
    }
}

// ---------------------------------------------------------------- per-pass batch preparation
// Two launches per pass: (1) mean and unbiased std of each critic's advantages
// over the minibatch rows (the reference normalises the minibatch copy every pass,
// ppo_lag.py:178-182; float64 accumulate), (2) the minibatch's rows permuted into pass order:
// obs_p[pos] and the per-row loss inputs rd_p[pos] = act | logp_old | adv_n[c] | ret[c].
// After this the fused step kernel reads contiguous, coalesced tiles with no index chase.
struct PrepArgs {
    const float* obs; const float* act; const float* advs; const float* rets; const float* logp_old;
    const int* perm; const int* mb_start; const int* mb_size;
    float* obs_p; float* rd_p;
    int N, C, Do, Da, norm_adv;
    const float* values;     // optional [C][N]: value_clip's batch.values -> rd_p[FSRL_RD_VOLD + c]
    const float* mean_old;   // optional [N][Da] + sigma_old[Da] (log std at process time): FOCOPS needs the old distribution
    const float* sigma_old;
    float* mbstat;           // [n_minibatches][FSRL_MAX_CRITICS][2]: mean, std of the minibatch's advantages per critic
    int batch, nmb;          // minibatch k covers pass positions [k * batch, (k + 1) * batch), the last one to the end
};

// (1) one block per minibatch: mean and unbiased std of each critic's advantages over the minibatch rows -> mbstat[mb][c][2]
__global__ __launch_bounds__(1024) void ppo_adv_stats_kernel(const PrepArgs a) {
    constexpr int NT = 1024, NW = NT / 64;
    __shared__ double sh[NW];
    const int mb = blockIdx.x, tid = threadIdx.x;
    const int st = a.mb_start[mb], n = a.mb_size[mb];
    auto block_sum = [&](double v) -> double {      // fixed order: waves 0..15
        v = wave_sum_d(v);
        __syncthreads();
        if ((tid & 63) == 0) sh[tid >> 6] = v;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += sh[w];
        return t;
    };
    for (int c = 0; c < a.C; ++c) {
        const float* __restrict__ adv = a.advs + (size_t)c * a.N;
        double s = 0.0;
        for (int m = tid; m < n; m += NT) s += (double)adv[a.perm[st + m]];
        const double mean = block_sum(s) / (double)n;
        double q = 0.0;
        for (int m = tid; m < n; m += NT) {
            const double d = (double)adv[a.perm[st + m]] - mean;
            q += d * d;
        }
        const double var = block_sum(q) / (double)(n - 1);
        if (tid == 0) {
            a.mbstat[((size_t)mb * FSRL_MAX_CRITICS + c) * 2 + 0] = (float)mean;
            a.mbstat[((size_t)mb * FSRL_MAX_CRITICS + c) * 2 + 1] = (float)sqrt(var);
        }
    }
}

// (2) 64 pass positions per block: the rows permuted into pass order, advantages normalised with their minibatch's statistics.
// One block per minibatch did both steps and took 28 us at batch 256 (78 blocks) and 106 us at batch 1 024 (19 blocks of a
// 256-CU chip, each walking 1 024+ rows); as two launches the row work spreads over N / 64 blocks.
__global__ __launch_bounds__(256) void ppo_permute_rows_kernel(const PrepArgs a) {
    __shared__ int perm_s[64];
    __shared__ float mean_f[64][FSRL_MAX_CRITICS], sd_f[64][FSRL_MAX_CRITICS];      // per row: its minibatch's statistics
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * 64, nc = min(64, a.N - p0);
    if (tid < nc) {
        perm_s[tid] = a.perm[p0 + tid];
        const int mb = min((p0 + tid) / a.batch, a.nmb - 1);          // minibatches are [k * batch, ...), the last one merged
        for (int c = 0; c < a.C; ++c) {
            mean_f[tid][c] = a.norm_adv ? a.mbstat[((size_t)mb * FSRL_MAX_CRITICS + c) * 2 + 0] : 0.0f;
            sd_f[tid][c] = a.norm_adv ? a.mbstat[((size_t)mb * FSRL_MAX_CRITICS + c) * 2 + 1] : 1.0f;
        }
    }
    __syncthreads();
    {   // thread = (field f, row phase): the field decides the source ONCE, then the thread's 16 rows are 16 independent gathers
        // (a per-element if-chain inside the row loop made every one of them its own round trip: 19 us for 5 MB)
        const int f = tid & 63, mph = tid >> 6;
        const float* __restrict__ src = nullptr;
        int stride = 1, kind = 0;                     // kind 1: advantage of critic cn (normalised per minibatch)
        int cn = 0;
        float cst = 0.0f;
        if (f < a.Da) { src = a.act + f; stride = a.Da; }
        else if (f == FSRL_RD_LOGP) src = a.logp_old;
        else if (f >= FSRL_RD_ADV && f < FSRL_RD_ADV + a.C) { cn = f - FSRL_RD_ADV; src = a.advs + (size_t)cn * a.N; kind = 1; }
        else if (f >= FSRL_RD_RET && f < FSRL_RD_RET + a.C) src = a.rets + (size_t)(f - FSRL_RD_RET) * a.N;
        else if (a.values && f >= FSRL_RD_VOLD && f < FSRL_RD_VOLD + a.C) src = a.values + (size_t)(f - FSRL_RD_VOLD) * a.N;
        else if (a.mean_old && f >= FSRL_RD_MEAN && f < FSRL_RD_MEAN + a.Da) { src = a.mean_old + (f - FSRL_RD_MEAN); stride = a.Da; }
        else if (a.mean_old && f >= FSRL_RD_STD && f < FSRL_RD_STD + a.Da) cst = expf(a.sigma_old[f - FSRL_RD_STD]);
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int m = mph + 4 * u;
            v[u] = (src != nullptr && m < nc) ? src[(size_t)perm_s[min(m, nc - 1)] * stride] : cst;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int m = mph + 4 * u;
            if (m < nc) {
                float x = v[u];
                if (kind == 1 && a.norm_adv) x = (x - mean_f[m][cn]) / sd_f[m][cn];
                a.rd_p[(size_t)(p0 + m) * FSRL_RD + f] = x;
            }
        }
    }
    for (int e = tid; e < nc * a.Do; e += 256) {
        const int m = e / a.Do, k = e - m * a.Do;
        a.obs_p[(size_t)(p0 + m) * a.Do + k] = a.obs[(size_t)perm_s[m] * a.Do + k];
    }
}

// ---------------------------------------------------------------- grad clip + Adam
// rebuild the forward-fragment mirrors of every W2 from the row-major matrices (after a host upload)
__global__ __launch_bounds__(256) void w2f_sync_kernel(float* __restrict__ P, const ModelDesc md) {
    const int H = md.H, q4 = H * H / 4;                       // float4 groups per network
    for (int e = blockIdx.x * 256 + threadIdx.x; e < md.n_nets * q4; e += gridDim.x * 256) {
        const int net = e / q4, g = e - net * q4;
        const int n = (4 * g) / H, k = (4 * g) - n * H;       // 4 consecutive k of row n = one mirror float4
        const f32x4 v = *reinterpret_cast<const f32x4*>(P + md.net[net].W2 + (size_t)n * H + k);
        *reinterpret_cast<f32x4*>(P + md.net[net].W2f + w2f_index(H, n, k)) = v;
    }
}

#ifndef ADAM_NT
#define ADAM_NT 256
#endif
// clip_grad_norm_(max_norm) then torch.optim.Adam single-tensor update, same operation order
// as torch (lerp_ / mul_+addcmul_ / sqrt / div / add_(eps) / addcdiv_), ppo_lag.py:235-241.
__device__ __forceinline__ void adam_clip_body(float* __restrict__ P, float* __restrict__ M,
                                               float* __restrict__ V,
                                               const float* __restrict__ G,
                                               const float* __restrict__ gsq_part, int nparts,
                                               int n, const PpoStepArgs& sa,
                                               CtrlBlock* ctrl, const ModelDesc& md) {
    __shared__ double sh[ADAM_NT / 64];
    __shared__ float coef_s;
    const int tid = threadIdx.x;
    if (FSRL_PROBE(sa, 30)) return;                     // launch floor of this kernel (timing experiments only)
    const int i4 = (blockIdx.x * ADAM_NT + tid) * 4;   // float4 per thread
    // issue every load of this thread first (one cold round trip), then reduce the norm
    f32x4 g = {0, 0, 0, 0}, m = g, v = g, p = g;
    if (i4 < n) {
        g = *reinterpret_cast<const f32x4*>(G + i4); m = *reinterpret_cast<const f32x4*>(M + i4);
        v = *reinterpret_cast<const f32x4*>(V + i4); p = *reinterpret_cast<const f32x4*>(P + i4);
    }
    float coef = 1.0f;
    if (sa.max_grad_norm > 0.0f) {
        double s = 0.0;
        for (int k = tid; k < nparts; k += ADAM_NT) s += (double)gsq_part[k];
        s = wave_sum_d(s);
        if ((tid & 63) == 0) sh[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int w = 0; w < ADAM_NT / 64; ++w) tot += sh[w];
            const float norm = sqrtf((float)tot);
            coef_s = fminf(sa.max_grad_norm / (norm + 1e-6f), 1.0f);
            if (blockIdx.x == 0) ctrl->last_grad_norm = norm;
        }
        __syncthreads();
        coef = coef_s;
    }
    if (i4 < n) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float me = m[e], ve = v[e], pe = p[e];
            ppo_adam_math(g[e] * coef, me, ve, pe, sa);
            m[e] = me; v[e] = ve; p[e] = pe;
        }
        store4_next(M + i4, m);             // written through as well: nothing of this kernel is left dirty at its boundary
        store4_next(V + i4, v);             // (same-box A/B: 116.3 -> 116.9 updates/s, every one of three alternations)
        store4_next(P + i4, p);
        const int mi = w2f_mirror_of(md, i4);           // 4 consecutive k of one W2 row = one mirror float4
        if (mi >= 0) store4_next(P + mi, p);
    }
    // pass-level KL early stop (ppo_lag.py:251-255); only after the last minibatch of a pass
    if (sa.last_in_pass && blockIdx.x == 0 && tid == 0 && sa.target_kl > 0.0f) {
        const double mean_kl = ctrl->kl_sum / ((double)sa.iters_in_pass + 1e-7);
        if (mean_kl > sa.kl_thresh) ctrl->stopped_after = sa.pass;
    }
}

__global__ __launch_bounds__(ADAM_NT) void adam_clip_kernel(float* __restrict__ P, float* __restrict__ M,
                                                       float* __restrict__ V,
                                                       const float* __restrict__ G,
                                                       const float* __restrict__ gsq_part, int nparts,
                                                       int n, const PpoStepArgs sa,
                                                       CtrlBlock* ctrl, const ModelDesc md) {
    adam_clip_body(P, M, V, G, gsq_part, nparts, n, sa, ctrl, md);
}

// ---- the stand-alone n-step return (nstep_return, base_policy.py:543-567) behind fsrl_nstep_return: one lane per batch
// row, the reference's float64 operations in its order (gamma * returns rounded before the add, gamma_buffer built by
// repeated multiplication): bit-exact with the sequential numba loop for any n_step / q.
struct NstepArgs {
    const double* metric;    // [len] rew or info.cost of the WHOLE buffer (base_policy.py:481)
    const uint8_t* end_flag; // [len] done | unfinished
    const float* target_q;   // [bsz][q] value-masked targets (float32, as torch hands them over)
    const int64_t* indices;  // [n_step][bsz] the buffer.next chain
    double* out;             // [bsz][q]
    int64_t len;
    int bsz, q, n_step;
    double gamma;
};
__global__ __launch_bounds__(256) void nstep_return_kernel(const NstepArgs a) {
#pragma clang fp contract(off)
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.bsz) return;
    int gammas = a.n_step;
    double ret = 0.0;
    for (int n = a.n_step - 1; n >= 0; --n) {
        const int64_t now = a.indices[(size_t)n * a.bsz + b];
        if (a.end_flag[now]) { gammas = n + 1; ret = 0.0; }
        const double t = a.gamma * ret;
        ret = a.metric[now] + t;
    }
    double gpow = 1.0;
    for (int i = 0; i < gammas; ++i) gpow = gpow * a.gamma;          // gamma_buffer[gammas]
    for (int j = 0; j < a.q; ++j) {
        const double prod = (double)a.target_q[(size_t)b * a.q + j] * gpow;
        a.out[(size_t)b * a.q + j] = prod + ret;
    }
}

