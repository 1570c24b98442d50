// kernels_mlp.hpp -- the dense part of the policy update, hand-written for gfx950.
//
// Topology (tianshou-0.5 Net + ActorProb/Critic as FSRL builds them,
// fsrl/agent/ppo_lag_agent.py:136-153):  x[Do] -> Linear(H) -> ReLU -> Linear(H) -> ReLU ->
// Linear(out);  actor: mu = max_action*tanh(out), sigma = exp(sigma_param);  critic: V = out.
//
// Work decomposition (DESIGN.md "Kernels"):
//   * one workgroup (4 waves) owns ONE 16-row M-tile of ONE network and runs the whole
//     forward, the loss head and the activation backward for it (ppo_fwd_bwd_kernel).  The two
//     H x H GEMMs run on v_mfma_f32_16x16x4_f32 (exact fp32); the weight operand is streamed
//     straight from L2 into VGPRs (it is used by exactly one wave once per tile, so an LDS
//     round trip would be pure overhead), the activation operand comes from LDS as b128.
//   * weight gradients are a second kernel (ppo_wgrad_kernel): 32x32 output tiles per
//     workgroup with split-K over the 4 waves, so no per-tile partial gradients ever reach HBM.
#pragma once
#include "common.hpp"

template <int H>
struct TileSmem {
    static constexpr int LD = H + 4;  // +4 floats: rows land on different banks, b128-aligned
    float xT[FSRL_MAX_OBS * 16];      // obs tile transposed [k][i]
    float h1[16 * LD];
    float h2[16 * LD];
    float d2[16 * LD];                // dL/dz2 (after ReLU mask)
    float out[16 * FSRL_MAX_ACT];     // head pre-activation [i][o]
    float dout[16 * FSRL_DOW];        // [i][0..16) dL/dout, [i][16..32) dL/dsigma_param rows
    int rowidx[16];
};

// ---------------------------------------------------------------- forward of one 16-row tile
// Pre: sm.xT filled and __syncthreads() done.  Post: sm.h1, sm.h2, sm.out valid (synced).
template <int H>
__device__ __forceinline__ void tile_forward(TileSmem<H>& sm, const float* __restrict__ P,
                                             const NetOff no, const int Do, const int tid) {
    constexpr int LD = TileSmem<H>::LD;
    constexpr int NTW = H / 64;  // 16-wide N tiles per wave
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;

    // ---- layer 1 (K = Do is tiny: plain FMA, one output column per thread)
    for (int j = tid; j < H; j += 256) {
        float acc[16];
        const float b = P[no.b1 + j];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = b;
        const float* __restrict__ w = P + no.W1 + (size_t)j * Do;
        for (int k = 0; k < Do; ++k) {
            const float wk = w[k];
            const f32x4* xr = reinterpret_cast<const f32x4*>(&sm.xT[k * 16]);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 x = xr[v];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * v + e] = fmaf(x[e], wk, acc[4 * v + e]);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sm.h1[i * LD + j] = fmaxf(acc[i], 0.0f);
    }
    __syncthreads();

    // ---- layer 2: h2[16,H] = relu(h1[16,H] @ W2^T + b2) on MFMA 16x16x4 (fp32)
    {
        f32x4 acc[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* __restrict__ W2 = P + no.W2;
        const int n0 = wave * NTW * 16;
        // lane (li,q) feeds k-slot q; over 4 MFMAs it covers k = kc+4q+{0..3} (one float4)
        const float* __restrict__ wrow[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) wrow[t] = W2 + (size_t)(n0 + t * 16 + li) * H + 4 * q;
        const float* arow = &sm.h1[li * LD + 4 * q];
#pragma unroll 4
        for (int kc = 0; kc < H; kc += 16) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + kc);
            f32x4 b[NTW];
#pragma unroll
            for (int t = 0; t < NTW; ++t) b[t] = *reinterpret_cast<const f32x4*>(wrow[t] + kc);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[t] = mfma_16x16x4(a[s], b[t][s], acc[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int j = n0 + t * 16 + li;
            const float bias = P[no.b2 + j];
#pragma unroll
            for (int r = 0; r < 4; ++r) sm.h2[(4 * q + r) * LD + j] = fmaxf(acc[t][r] + bias, 0.0f);
        }
    }
    __syncthreads();

    // ---- head (out <= 16): 16 lanes per row, shuffle-reduce
    {
        const int i = tid >> 4, p = tid & 15;
        for (int o = 0; o < no.out; ++o) {
            const float* __restrict__ w3 = P + no.W3 + (size_t)o * H;
            float s = 0.0f;
#pragma unroll 4
            for (int k = p; k < H; k += 16) s = fmaf(sm.h2[i * LD + k], w3[k], s);
            s += __shfl_xor(s, 8, 64);
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 1, 64);
            if (p == 0) sm.out[i * FSRL_MAX_ACT + o] = s + P[no.b3 + o];
        }
    }
    __syncthreads();
}

// load 16 gathered observation rows (transposed) into LDS
template <int H>
__device__ __forceinline__ void tile_load_x(TileSmem<H>& sm, const float* __restrict__ obs,
                                            const int Do, const int tid) {
    for (int e = tid; e < 16 * Do; e += 256) {
        const int i = e / Do, k = e - i * Do;
        const int r = sm.rowidx[i];
        sm.xT[k * 16 + i] = (r >= 0) ? obs[(size_t)r * Do + k] : 0.0f;
    }
}

// ---------------------------------------------------------------- process_fn inference
// grid = (ceil(N/16), 2*C + 1).  job < C: V_job(obs) ; C <= job < 2C: V(obs_next)*~terminated ;
// job == 2C: logp_old = log N(act | mu(obs), sigma)      (fsrl/policy/base_policy.py:416-428,
// fsrl/policy/ppo_lag.py:142-149)
struct InferArgs {
    const float* obs;
    const float* obs_next;
    const float* act;
    const uint8_t* flags;   // bit0 terminated
    float* values;          // [C][N]
    float* vnext;           // [C][N]  (masked)
    float* logp_old;        // [N]
    float* mu_out;          // optional [N][Da] (actor_forward API)  may be null
    int N, C;
    float max_action;
};

#define LOG_SQRT_2PI 0.9189385332046727f

template <int H>
__global__ __launch_bounds__(256) void mlp_infer_kernel(const float* __restrict__ P,
                                                       const ModelDesc md, const InferArgs a) {
    __shared__ TileSmem<H> sm;
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * 16;
    const int job = blockIdx.y;
    const int C = a.C;
    const bool is_actor = (job == 2 * C);
    const int net = is_actor ? 0 : 1 + (C > 0 ? job % C : 0);
    const bool use_next = (!is_actor) && job >= C;
    const NetOff no = md.net[net];
    if (tid < 16) sm.rowidx[tid] = (row0 + tid < a.N) ? row0 + tid : -1;
    __syncthreads();
    tile_load_x(sm, use_next ? a.obs_next : a.obs, md.Do, tid);
    __syncthreads();
    tile_forward<H>(sm, P, no, md.Do, tid);
    if (tid < 16 && sm.rowidx[tid] >= 0) {
        const int r = sm.rowidx[tid];
        if (!is_actor) {
            float v = sm.out[tid * FSRL_MAX_ACT];
            const int c = (C > 0) ? job % C : 0;
            if (use_next) {
                if (a.flags[r] & 1) v = 0.0f;  // v_next * ~terminated
                a.vnext[(size_t)c * a.N + r] = v;
            } else {
                a.values[(size_t)c * a.N + r] = v;
            }
        } else {
            float logp = 0.0f;
            for (int d = 0; d < md.Da; ++d) {
                const float mu = a.max_action * tanhf(sm.out[tid * FSRL_MAX_ACT + d]);
                const float sig = expf(P[no.sigma + d]);
                if (a.mu_out) a.mu_out[(size_t)r * md.Da + d] = mu;
                if (a.act) {
                    const float diff = a.act[(size_t)r * md.Da + d] - mu;
                    logp += -(diff * diff) / (2.0f * sig * sig) - logf(sig) - LOG_SQRT_2PI;
                }
            }
            if (a.logp_old) a.logp_old[r] = logp;
        }
    }
}

// ---------------------------------------------------------------- fused fwd + loss + bwd
// One PPO minibatch step, activation side.  grid = (ceil(mb/16), n_nets).
// Implements, for its 16 rows: PPOLagrangian.policy_loss / critics_loss gradients
// (fsrl/policy/ppo_lag.py:152-212, lagrangian_base.py:145-166) analytically.
struct PpoBatchPtrs {
    const float* obs;        // [N][Do]  batch in sample(0) order
    const float* act;        // [N][Da]
    const float* advs;       // [C][N]
    const float* rets;       // [C][N]
    const float* logp_old;   // [N]
    const int* perm;         // [N] permutation of this pass
    const float* mbstats;    // [n_mb][C][2] = (mean, 1/std) of advs per minibatch of this pass
    // per-net activation side buffers, row = position inside the minibatch
    float* A1;               // [n_nets][mbp_max][H]   relu(z1)
    float* A2;               // [n_nets][mbp_max][H]   relu(z2)
    float* D1;               // [n_nets][mbp_max][H]   dL/dz1
    float* D2;               // [n_nets][mbp_max][H]   dL/dz2
    float* DO;               // [n_nets][mbp_max][FSRL_DOW]
    float* XB;               // [mbp_max][Do]          gathered obs rows
    float* statp;            // [n_tiles_max][n_nets][4] partial sums of the logged stats
    const CtrlBlock* ctrl;
    int mbp_max;
    int N;
};

template <int H>
__global__ __launch_bounds__(256) void ppo_fwd_bwd_kernel(const float* __restrict__ P,
                                                         const ModelDesc md,
                                                         const PpoBatchPtrs bp,
                                                         const PpoStepArgs sa) {
    __shared__ TileSmem<H> sm;
    constexpr int LD = TileSmem<H>::LD;
    constexpr int NTW = H / 64;
    if (sa.pass > bp.ctrl->stopped_after) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x, net = blockIdx.y;
    const int row0 = tile * 16;
    const NetOff no = md.net[net];
    const int Do = md.Do, Da = md.Da, C = md.n_nets - 1;

    if (tid < 16) {
        const int m = row0 + tid;
        sm.rowidx[tid] = (m < sa.mb_size) ? bp.perm[sa.mb_start + m] : -1;
    }
    __syncthreads();
    tile_load_x(sm, bp.obs, Do, tid);
    __syncthreads();
    tile_forward<H>(sm, P, no, Do, tid);

    // ---- loss head: dL/dout per row + partial sums of the logged statistics
    for (int e = tid; e < 16 * FSRL_DOW; e += 256) sm.dout[e] = 0.0f;
    __syncthreads();
    if (tid < 16) {
        const int i = tid;
        const int r = sm.rowidx[i];
        const float invB = 1.0f / (float)sa.mb_size;
        float st0 = 0.f, st1 = 0.f, st2 = 0.f, st3 = 0.f;
        if (r >= 0) {
            if (net == 0) {
                float logp = 0.0f;
                for (int d = 0; d < Da; ++d) {
                    const float mu_d = sa.max_action * tanhf(sm.out[i * FSRL_MAX_ACT + d]);
                    const float sig = expf(P[no.sigma + d]);
                    const float df = bp.act[(size_t)r * Da + d] - mu_d;
                    logp += -(df * df) / (2.0f * sig * sig) - logf(sig) - LOG_SQRT_2PI;
                }
                const float lpo = bp.logp_old[r];
                const float ratio = expf(logp - lpo);
                // advantages, normalised per minibatch copy (ppo_lag.py:178-182)
                float adv[FSRL_MAX_CRITICS];
#pragma unroll
                for (int c = 0; c < FSRL_MAX_CRITICS; ++c) {
                    float av = 0.0f;
                    if (c < C) {
                        av = bp.advs[(size_t)c * bp.N + r];
                        if (sa.norm_adv) {
                            const float mean = bp.mbstats[(sa.mb_index * C + c) * 2 + 0];
                            const float sd = bp.mbstats[(sa.mb_index * C + c) * 2 + 1];
                            av = (av - mean) / sd;
                        }
                    }
                    adv[c] = av;
                }
                const float ar = adv[0];
                const float s1 = ratio * ar;
                const float rc = fminf(fmaxf(ratio, 1.0f - sa.eps_clip), 1.0f + sa.eps_clip);
                const float s2 = rc * ar;
                const bool inrange = (ratio >= 1.0f - sa.eps_clip) && (ratio <= 1.0f + sa.eps_clip);
                // d min(s1,s2)/d ratio with torch's tie rule (equal => gradient shared)
                float g_c1 = inrange ? ar : (s1 < s2 ? ar : (s1 == s2 ? 0.5f * ar : 0.0f));
                float term = fminf(s1, s2);
                float g_term = g_c1;
                if (sa.dual_clip > 0.0f) {
                    const float c1 = term;
                    const float lim = sa.dual_clip * ar;
                    const float c2 = fmaxf(c1, lim);
                    if (ar < 0.0f) {
                        term = c2;
                        g_term = (c1 > lim) ? g_c1 : (c1 == lim ? 0.5f * g_c1 : 0.0f);
                    }
                }
                float dL_dratio = -g_term * invB;
                float safety_sum = 0.0f;
                if (sa.use_lagrangian) {
#pragma unroll
                    for (int c = 1; c < FSRL_MAX_CRITICS; ++c) {
                        if (c < C) {
                            dL_dratio += sa.lam[c - 1] * adv[c] * invB;
                            safety_sum += ratio * adv[c] * sa.lam[c - 1];
                        }
                    }
                }
                const float dL_dlogp = sa.rescale * dL_dratio * ratio;
                for (int d = 0; d < Da; ++d) {
                    const float th = tanhf(sm.out[i * FSRL_MAX_ACT + d]);
                    const float sig = expf(P[no.sigma + d]);
                    const float var = sig * sig;
                    const float df = bp.act[(size_t)r * Da + d] - sa.max_action * th;
                    sm.dout[i * FSRL_DOW + d] =
                        dL_dlogp * (df / var) * sa.max_action * (1.0f - th * th);
                    sm.dout[i * FSRL_DOW + 16 + d] = dL_dlogp * (df * df / var - 1.0f);
                }
                st0 = term;          // sum of min(surr1,surr2) (-> loss/actor_rew)
                st1 = safety_sum;    // sum of ratio*A_c*lambda  (-> loss/actor_safety)
                st2 = lpo - logp;    // approx KL
            } else {
                const int c = net - 1;
                const float v = sm.out[i * FSRL_MAX_ACT];
                const float d = bp.rets[(size_t)c * bp.N + r] - v;
                sm.dout[i * FSRL_DOW] = -2.0f * sa.vf_coef * d * invB;
                st0 = d * d;
            }
        }
        // 16-lane reduce (lanes 0..15 of wave 0)
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            st0 += __shfl_xor(st0, o, 64);
            st1 += __shfl_xor(st1, o, 64);
            st2 += __shfl_xor(st2, o, 64);
        }
        if (i == 0) {
            float* sp = bp.statp + ((size_t)tile * md.n_nets + net) * 4;
            sp[0] = st0; sp[1] = st1; sp[2] = st2; sp[3] = st3;
        }
    }
    __syncthreads();

    // ---- dL/dz2 = (dout @ W3) * relu'(z2)
    for (int k = tid; k < H; k += 256) {
        float g[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) g[i] = 0.0f;
        for (int o = 0; o < no.out; ++o) {
            const float w = P[no.W3 + (size_t)o * H + k];
#pragma unroll
            for (int i = 0; i < 16; ++i) g[i] = fmaf(sm.dout[i * FSRL_DOW + o], w, g[i]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sm.d2[i * LD + k] = (sm.h2[i * LD + k] > 0.0f) ? g[i] : 0.0f;
    }
    __syncthreads();

    // ---- dL/dz1 = (dz2 @ W2) * relu'(z1) on MFMA; result goes straight to HBM/L2
    const size_t nb = (size_t)net * bp.mbp_max;
    {
        f32x4 acc[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // MFMA t owns the interleaved columns col0 + t, col0 = wave*16*NTW + NTW*li
        const int col0 = wave * 16 * NTW + NTW * li;
        const float* __restrict__ W2c = P + no.W2 + col0;
        const float* arow = &sm.d2[li * LD + 4 * q];
#pragma unroll 2
        for (int jc = 0; jc < H; jc += 16) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + jc);
            float b[4][NTW];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float* src = W2c + (size_t)(jc + 4 * q + s) * H;
                if constexpr (NTW == 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src);
                    b[s][0] = v[0]; b[s][1] = v[1]; b[s][2] = v[2]; b[s][3] = v[3];
                } else if constexpr (NTW == 2) {
                    const f32x2 v = *reinterpret_cast<const f32x2*>(src);
                    b[s][0] = v[0]; b[s][1] = v[1];
                } else {
                    b[s][0] = *src;
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[t] = mfma_16x16x4(a[s], b[s][t], acc[t]);
            }
        }
        float* __restrict__ D1 = bp.D1 + (nb + row0) * H;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            float v[NTW];
#pragma unroll
            for (int t = 0; t < NTW; ++t)
                v[t] = (sm.h1[i * LD + col0 + t] > 0.0f) ? acc[t][r] : 0.0f;
            float* dst = D1 + (size_t)i * H + col0;
            if constexpr (NTW == 4) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            } else if constexpr (NTW == 2) {
                *reinterpret_cast<f32x2*>(dst) = f32x2{v[0], v[1]};
            } else {
                *dst = v[0];
            }
        }
    }

    // ---- spill the tile's activations for the weight-gradient kernel (coalesced float4)
    {
        float* __restrict__ A1 = bp.A1 + (nb + row0) * H;
        float* __restrict__ A2 = bp.A2 + (nb + row0) * H;
        float* __restrict__ D2 = bp.D2 + (nb + row0) * H;
        constexpr int H4 = H / 4;
        for (int e = tid; e < 16 * H4; e += 256) {
            const int i = e / H4, c4 = e - i * H4;
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(&sm.h1[i * LD + 4 * c4]);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(&sm.h2[i * LD + 4 * c4]);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(&sm.d2[i * LD + 4 * c4]);
            *reinterpret_cast<f32x4*>(&A1[(size_t)i * H + 4 * c4]) = v1;
            *reinterpret_cast<f32x4*>(&A2[(size_t)i * H + 4 * c4]) = v2;
            *reinterpret_cast<f32x4*>(&D2[(size_t)i * H + 4 * c4]) = v3;
        }
        float* __restrict__ DOb = bp.DO + (nb + row0) * FSRL_DOW;
        for (int e = tid; e < 16 * FSRL_DOW; e += 256) DOb[e] = sm.dout[e];
        if (net == 0) {
            float* __restrict__ XB = bp.XB + (size_t)row0 * Do;
            for (int e = tid; e < 16 * Do; e += 256) {
                const int i = e / Do, k = e - i * Do;
                XB[e] = sm.xT[k * 16 + i];
            }
        }
    }
}

// ---------------------------------------------------------------- weight gradients
// grid.x = n_nets * (NT2 + NA): NT2 = (H/32)^2 MFMA tile blocks (dW2) + NA = H/64 aux blocks
// (dW1, db1, db2, dW3, db3, dsigma) per network.  Each block also emits the sum of squares
// of the gradient entries it produced (for clip_grad_norm_, ppo_lag.py:237-240).
struct WgradPtrs {
    const float* A1; const float* A2; const float* D1; const float* D2; const float* DO;
    const float* XB;
    float* grad;       // flat, same layout as the parameters
    float* gsq_part;   // [gridDim.x]
    const CtrlBlock* ctrl;
    int mbp_max;
};

template <int H>
__global__ __launch_bounds__(256) void ppo_wgrad_kernel(const ModelDesc md, const WgradPtrs wp,
                                                       const int mbp, const int pass) {
    constexpr int TPD = H / 32;          // tiles per dimension
    constexpr int NT2 = TPD * TPD;
    constexpr int NA = H / 64;
    constexpr int PB = NT2 + NA;
    __shared__ float red[256 * 20];      // split-K partials of the 4 waves / aux reduce scratch
    __shared__ float xs[64 * 16];        // aux: 64-row x 16-col chunk of XB
    __shared__ float dos[64 * FSRL_DOW];
    __shared__ float wsum[4];
    if (pass > wp.ctrl->stopped_after) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int net = blockIdx.x / PB, rb = blockIdx.x % PB;
    const NetOff no = md.net[net];
    const size_t nb = (size_t)net * wp.mbp_max;
    const float* __restrict__ A1 = wp.A1 + nb * H;
    const float* __restrict__ A2 = wp.A2 + nb * H;
    const float* __restrict__ D1 = wp.D1 + nb * H;
    const float* __restrict__ D2 = wp.D2 + nb * H;
    const float* __restrict__ DOb = wp.DO + nb * FSRL_DOW;
    float sq = 0.0f;

    if (rb < NT2) {
        // ---- dW2[j][k] = sum_r D2[r][j] * A1[r][k], 32x32 tile, split-K over waves
        const int tj = rb / TPD, tk = rb % TPD;
        const int c = lane & 15, q = lane >> 4;
        f32x4 acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
        const float* __restrict__ pa = D2 + tj * 32 + 2 * c;
        const float* __restrict__ pb = A1 + tk * 32 + 2 * c;
        const int KS = mbp >> 2;
#pragma unroll 4
        for (int s = wave; s < KS; s += 4) {
            const size_t r = (size_t)(4 * s + q) * H;
            const f32x2 a = *reinterpret_cast<const f32x2*>(pa + r);
            const f32x2 b = *reinterpret_cast<const f32x2*>(pb + r);
            acc00 = mfma_16x16x4(a[0], b[0], acc00);
            acc01 = mfma_16x16x4(a[0], b[1], acc01);
            acc10 = mfma_16x16x4(a[1], b[0], acc10);
            acc11 = mfma_16x16x4(a[1], b[1], acc11);
        }
        // acc_tu[r]: j_local = 2*(4q+r)+t, k_local = 2c+u
        float* myred = red + wave * (32 * 33);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jl = 2 * (4 * q + r);
            myred[(jl + 0) * 33 + 2 * c + 0] = acc00[r];
            myred[(jl + 0) * 33 + 2 * c + 1] = acc01[r];
            myred[(jl + 1) * 33 + 2 * c + 0] = acc10[r];
            myred[(jl + 1) * 33 + 2 * c + 1] = acc11[r];
        }
        __syncthreads();
        float* __restrict__ g = wp.grad + no.W2;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int e = tid + 256 * m;
            const int jl = e >> 5, kl = e & 31;
            const float v = red[jl * 33 + kl] + red[(32 * 33) + jl * 33 + kl] +
                            red[2 * (32 * 33) + jl * 33 + kl] + red[3 * (32 * 33) + jl * 33 + kl];
            g[(size_t)(tj * 32 + jl) * H + tk * 32 + kl] = v;
            sq = fmaf(v, v, sq);
        }
    } else {
        // ---- aux: column j of this 64-wide chunk; rows split over the 4 waves
        const int ch = rb - NT2;
        const int cidx = tid & 63, ph = tid >> 6;
        const int j = ch * 64 + cidx;
        const int Do = md.Do, out = no.out;
        float db1 = 0.f, db2 = 0.f;
        float dw3[FSRL_MAX_ACT];
#pragma unroll
        for (int o = 0; o < FSRL_MAX_ACT; ++o) dw3[o] = 0.f;
        float dsum = 0.f;  // chunk 0, lanes < 32: column sums of DO (db3 / dsigma)
        for (int k0 = 0; k0 < Do; k0 += 16) {
            const int kn = min(16, Do - k0);
            float dw1[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) dw1[k] = 0.f;
            for (int r0 = 0; r0 < mbp; r0 += 64) {
                const int rn = min(64, mbp - r0);
                __syncthreads();
                for (int e = tid; e < rn * kn; e += 256) {
                    const int rr = e / kn, k = e - rr * kn;
                    xs[rr * 16 + k] = wp.XB[(size_t)(r0 + rr) * Do + k0 + k];
                }
                if (k0 == 0)
                    for (int e = tid; e < rn * FSRL_DOW; e += 256) dos[e] = DOb[(size_t)r0 * FSRL_DOW + e];
                __syncthreads();
                for (int rr = ph; rr < rn; rr += 4) {
                    const size_t r = (size_t)(r0 + rr) * H + j;
                    const float d1 = D1[r];
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k < kn) dw1[k] = fmaf(d1, xs[rr * 16 + k], dw1[k]);
                    if (k0 == 0) {
                        db1 += d1;
                        db2 += D2[r];
                        const float a2 = A2[r];
#pragma unroll
                        for (int o = 0; o < FSRL_MAX_ACT; ++o)
                            if (o < out) dw3[o] = fmaf(dos[rr * FSRL_DOW + o], a2, dw3[o]);
                        if (ch == 0 && cidx < FSRL_DOW) dsum += dos[rr * FSRL_DOW + cidx];
                    }
                }
            }
            // reduce dW1 chunk over the 4 row phases and write
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; ++k) red[(ph * 64 + cidx) * 17 + k] = dw1[k];
            __syncthreads();
            if (ph == 0) {
                for (int k = 0; k < kn; ++k) {
                    const float v = red[cidx * 17 + k] + red[(64 + cidx) * 17 + k] +
                                    red[(128 + cidx) * 17 + k] + red[(192 + cidx) * 17 + k];
                    wp.grad[no.W1 + (size_t)j * Do + k0 + k] = v;
                    sq = fmaf(v, v, sq);
                }
            }
        }
        // reduce the k0==0 quantities: db1, db2, dw3[out], dsum  -> layout [ph][cidx][20]
        __syncthreads();
        {
            float* rr_ = red + (ph * 64 + cidx) * 20;
            rr_[0] = db1; rr_[1] = db2; rr_[2] = dsum;
#pragma unroll
            for (int o = 0; o < FSRL_MAX_ACT; ++o)
                if (o < out) rr_[3 + o] = dw3[o];
        }
        __syncthreads();
        if (ph == 0) {
            auto tot = [&](int f) {
                return red[cidx * 20 + f] + red[(64 + cidx) * 20 + f] + red[(128 + cidx) * 20 + f] +
                       red[(192 + cidx) * 20 + f];
            };
            float v = tot(0);
            wp.grad[no.b1 + j] = v; sq = fmaf(v, v, sq);
            v = tot(1);
            wp.grad[no.b2 + j] = v; sq = fmaf(v, v, sq);
            for (int o = 0; o < out; ++o) {
                v = tot(3 + o);
                wp.grad[no.W3 + (size_t)o * H + j] = v; sq = fmaf(v, v, sq);
            }
            if (ch == 0 && cidx < FSRL_DOW) {
                v = tot(2);
                if (cidx < out) { wp.grad[no.b3 + cidx] = v; sq = fmaf(v, v, sq); }
                if (no.sigma >= 0 && cidx >= 16 && cidx < 16 + md.Da) {
                    wp.grad[no.sigma + cidx - 16] = v; sq = fmaf(v, v, sq);
                }
            }
        }
    }
    // ---- block sum of squares (deterministic order)
    sq = wave_sum(sq);
    __syncthreads();
    if (lane == 0) wsum[wave] = sq;
    __syncthreads();
    if (tid == 0) wp.gsq_part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
