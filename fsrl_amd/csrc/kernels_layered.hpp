// kernels_layered.hpp -- layer-by-layer kernels for networks the fused kernels do not cover (all seven agents; written first for
// the PPO-Lagrangian update, whose launch sequence is listed here):
// `hidden_sizes` of any depth (1 .. FSRL_MAX_HIDDEN hidden layers) and any width (fsrl/agent/ppo_lag_agent.py:91,136-145:
// tianshou `Net(hidden_sizes=...)` under ActorProb / Critic).  The fused path (kernels_mlp.hpp) keeps a whole two-layer
// network of at most 256 units in one workgroup; here every Linear is its own MFMA GEMM launch over the minibatch (or, at
// process_fn time, over the whole batch), all networks of the policy in one launch (grid.z), activations in HBM between
// launches:
//     forward   Z_l = relu(Z_{l-1} W_l^T + b_l)             lin_kernel<LIN_F>   one launch per layer (+ the head)
//     loss      the PPO-Lag / value heads on the head outputs lay_ppo_head_kernel (the arithmetic of the fused kernel's head)
//     backward  dZ_{l-1} = (dZ_l W_l) * relu'(Z_{l-1})        lin_kernel<LIN_X>   one launch per layer
//               dW_l = dZ_l^T Z_{l-1}, db_l = colsum(dZ_l)    lin_kernel<LIN_W>   ONE launch for every layer of every network
//     then the shared ppo_stats_kernel and adam_clip_kernel.
// 2 L + 5 launches per minibatch step for L hidden layers (the fused path: 3).  fp32 on v_mfma_f32_16x16x4_f32; every
// reduction has a fixed order (no atomics), so an update is reproducible run to run.
#pragma once
#include "common.hpp"
#include "kernels_mlp.hpp"
#include "kernels_fb.hpp"

#define LAY_MAX_JOBS 32
#define LIN_F 0     // C[m][n] = act(sum_k A[m][k] B[n][k] + bias[n])            A: M x K,  B: N x K   (both k-contiguous)
#define LIN_X 1     // C[m][n] = (sum_k A[m][k] B[k][n]) * (mask[m][n] > 0)      A: M x K,  B: K x N
#define LIN_W 2     // C[m][n] = sum_k A[k][m] B[k][n];  bias_out[m] = sum_k A[k][m]   A: K x M,  B: K x N   (k = batch row)

struct LinJob {
    const float* A; const float* B; float* C;
    const float* aux;     // LIN_F: bias [N] ; LIN_X: the activations whose relu' masks the result, M x N with row stride ldaux
    float* bias_out;      // LIN_W: column sums of A (the bias gradient), length M; may be null
    int lda, ldb, ldc, ldaux;
    int M, N, K;
    int relu;             // LIN_F
    // the R-operator products of the trust-region path (Hessian-vector products, host_trust.inc); null / 0 everywhere else
    const float* acc;     // LIN_F / LIN_X: added to the product before bias / relu / mask, M x N with row stride ldacc (may be C itself)
    const float* mask;    // LIN_F: result := (mask[m][n] > 0) ? result : 0 after the bias, M x N with row stride ldmask
    const float* A2; const float* B2;     // LIN_W: a second operand pair accumulated into the same output (same shapes and strides)
    int ldacc, ldmask;
    int a_len;            // 0, or the READABLE length of A's contiguous dimension (K of a k-minor A, M of a k-major one) when the
                          // buffer is zero-padded beyond the logical length (head-gradient rows: FSRL_DOW wide, zeros past the head
                          // width) -- lets a 2-wide head share the float4 instantiation with the wide layers of the same launch
};
// ksplit > 1 (LIN_W only): the reduction over K (the batch rows) is cut into ksplit contiguous ranges of kchunk rows, range s
// writing its own partial at C + s * part_stride / bias_out + s * part_stride; the consumer adds the partials in float64 in
// ascending order (fb_sum_parts_kernel, adam_range_kernel, cg_pz_kernel: the split-K convention of fb_wgrad_kernel).
struct LinJobs { int n; int ksplit; int kchunk; int part_stride; LinJob j[LAY_MAX_JOBS]; };
static_assert(sizeof(LinJobs) + 16 <= 4096, "LinJobs travels as a kernel argument: keep it under the 4 KB kernarg segment");

// offsets of one network inside the flat parameter vector (API layout == device layout: no padding, no mirrors)
struct LayLayer { int W, b, in, out; };
struct LayNet { int sigma; int nl; LayLayer l[FSRL_MAX_HIDDEN + 1]; };      // l[0 .. nl-2] hidden layers, l[nl-1] the head
struct LayModel { int Do, Da, n_nets, unbounded; LayNet net[FSRL_MAX_NETS]; };

#define LIN_KC 64       // k-chunk staged per round trip (a 16-deep chunk made every round trip ~0.7 us of load latency for 16 MFMAs)
#define LIN_LD 68       // floats per row of a staged 64 x 64 operand tile, either orientation:
                        //   k-minor [64 operand rows][64 k]: fragment = one ds_read_b128 at row i, k = 16 kc + 4 q
                        //   k-major [64 k][64 operand cols]: fragment = 4 x ds_read_b32 (banks 16 q + l, every bank exactly twice)

// One 64 x 64 operand tile of a k-chunk: 256 threads x 4 float4, rows of 256 contiguous bytes.  `rows` x `cols` is the extent
// of the matrix the tile is cut from (k-minor: operand rows x K; k-major: K x operand columns).  Out-of-range elements are
// zeros, so ragged M / N / K need no special case further down.
// The loads are STRAIGHT-LINE (addresses clamped into the matrix, no lane-divergent branch around a load: a predicated load
// costs its whole latency on the spot, kernels_wgrad2.hpp) and the zero-fill is applied when the tile is written to LDS, two
// chunks later; `ok` carries one bit per element.  VEC (chosen by the host per launch): every operand row is 16-byte aligned
// and 4 | its length, so a float4 is all in or all out; otherwise four dword loads per slot.
template <int SLOTS> struct LinTile { f32x4 v[SLOTS]; unsigned ok; };     // SLOTS float4 per thread: 1024 / threads of the workgroup
// One operand's view for this thread: slot j = tid + NT j (NT threads per workgroup) covers tile row tr = slot >> 4, tile columns tc .. tc + 3 (tc = 4 (slot & 15)).
//   k-minor: tile row = operand row (fixed), tile column = k (advances by 64 per chunk)
//   k-major: tile row = k (advances),        tile column = operand column (fixed)
template <bool KMAJOR, bool VEC, int SLOTS>
struct LinOperand {
    static constexpr int NT = 1024 / SLOTS;
    const float* base; int ld, fixed_n, K, fixed0;     // fixed_n: extent of the operand dimension (rows of a k-minor operand, columns of a k-major one)
    int tid;
    __device__ __forceinline__ LinTile<SLOTS> load(const int k0) const {
        LinTile<SLOTS> t;
        t.ok = 0u;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int slot = tid + NT * j, tr = slot >> 4, tc = 4 * (slot & 15);
            const int f = fixed0 + (KMAJOR ? tc : tr);        // operand index (fixed over the chunks)
            const int k = k0 + (KMAJOR ? tr : tc);
            if (VEC) {
                // (max(0, .): an operand the job does not have -- the sigma_param job's B -- has extent 0 and reads element 0, masked)
                const size_t off = KMAJOR ? (size_t)min(k, K - 1) * ld + max(0, min(f, fixed_n - 4)) : (size_t)max(0, min(f, fixed_n - 1)) * ld + min(k, K - 4);
                t.v[j] = *reinterpret_cast<const f32x4*>(base + off);
                if (f < fixed_n && k < K) t.ok |= 0xFu << (4 * j);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int fe = KMAJOR ? f + e : f, ke = KMAJOR ? k : k + e;
                    const size_t off = KMAJOR ? (size_t)min(ke, K - 1) * ld + max(0, min(fe, fixed_n - 1)) : (size_t)max(0, min(fe, fixed_n - 1)) * ld + min(ke, K - 1);
                    t.v[j][e] = base[off];
                    if (fe < fixed_n && ke < K) t.ok |= 1u << (4 * j + e);
                }
            }
        }
        return t;
    }
};
template <int SLOTS>
__device__ __forceinline__ void lin_stage(float* __restrict__ s, const LinTile<SLOTS>& t, const int tid) {
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int idx = tid + (1024 / SLOTS) * j;
        f32x4 v = t.v[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((t.ok >> (4 * j + e)) & 1u) ? v[e] : 0.0f;
        *reinterpret_cast<f32x4*>(&s[(idx >> 4) * LIN_LD + 4 * (idx & 15)]) = v;
    }
}
// the four k-values (k = 16 kc + 4 q + s) of operand row / column `i` of the staged tile, as the MFMA wants them
template <bool KMAJOR>
__device__ __forceinline__ f32x4 lin_frag(const float* __restrict__ s, const int i, const int kc, const int q) {
    if (KMAJOR) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = s[(16 * kc + 4 * q + e) * LIN_LD + i];
        return v;
    }
    return *reinterpret_cast<const f32x4*>(&s[i * LIN_LD + 16 * kc + 4 * q]);
}

// grid = (max column tiles, max row tiles, jobs); 256 threads = 4 waves; workgroup tile 64 x 64, wave w owns rows 16 w .. 16 w + 15
// of it (4 MFMA column tiles), k-chunks of 64 staged through LDS (34 KB) with the next two chunks' global loads in flight.
// LIN_W additionally: the workgroups of column tile 0 reduce the bias gradient (ascending batch row), and EVERY workgroup of
// the grid writes its share of the squared gradient norm to gsq_part[linear block index] (0 for idle ones).
// NW = waves along the 64 output columns (1, 2 or 4): 256 NW threads, wave (wm = w & 3, wn = w >> 2) owns rows 16 wm .. + 15 and
// the 4 / NW column tiles from 16 (4 / NW) wn.  A minibatch-sized product has only a few hundred workgroups: with NW = 1 that is
// less than one wave per SIMD and every load, LDS read and barrier is exposed; NW = 4 puts up to four waves on each SIMD of the
// workgroup's CU (the host picks NW from the number of workgroups of the launch).
template <int FORM, bool VEC, int NW>
__global__ __launch_bounds__(256 * NW) void lin_kernel(const LinJobs jobs, float* __restrict__ gsq_part) {
    constexpr bool AKJ = (FORM == LIN_W), BKJ = (FORM != LIN_F);
    constexpr int T = 4 / NW, SLOTS = 4 / NW;
    __shared__ __attribute__((aligned(16))) float sA[64 * LIN_LD];
    __shared__ __attribute__((aligned(16))) float sB[64 * LIN_LD];
    __shared__ float red[4 * NW];
    const int tid = threadIdx.x, wv = tid >> 6, wave = wv & 3, wn = wv >> 2, lane = tid & 63, li = lane & 15, q = lane >> 4;
    LinJob jb = jobs.j[blockIdx.z];              // by value: the fields live in SGPRs, not re-read from the kernel arguments per chunk
    int by = blockIdx.y;
    if (FORM == LIN_W && jobs.ksplit > 1) {      // this workgroup's range of batch rows and its partial
        const int row_tiles = gridDim.y / jobs.ksplit, sp = by / row_tiles;
        by -= sp * row_tiles;
        const int kb = min(sp * jobs.kchunk, jb.K);
        jb.K = min(jb.K - kb, jobs.kchunk);
        jb.A += (size_t)kb * jb.lda; jb.B += (size_t)kb * jb.ldb;
        if (jb.A2) { jb.A2 += (size_t)kb * jb.lda; jb.B2 += (size_t)kb * jb.ldb; }
        jb.C += (size_t)sp * jobs.part_stride;
        if (jb.bias_out) jb.bias_out += (size_t)sp * jobs.part_stride;
    }
    const int M = jb.M, N = jb.N, K = jb.K;
    const int m0 = by * 64, n0 = blockIdx.x * 64;
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const bool bias_role = (FORM == LIN_W) && blockIdx.x == 0 && jb.bias_out != nullptr;
    if (m0 >= M || (n0 >= N && !bias_role)) {
        if (FORM == LIN_W && gsq_part && tid == 0) gsq_part[blk] = 0.0f;
        return;
    }
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    // operand geometry: A rows are output rows (k-minor) or batch rows (k-major); the same for B and the output columns
    // one pass over K for an operand pair; LIN_W may run a second pair into the same accumulators (R{dW} = R{dz}^T a + dz^T R{a})
    auto k_pass = [&](const float* Ap, const float* Bp, const bool with_bias) __attribute__((always_inline)) {
    const LinOperand<AKJ, VEC, SLOTS> opA{Ap, jb.lda, (AKJ && jb.a_len) ? jb.a_len : M, (!AKJ && jb.a_len) ? jb.a_len : K, m0, tid};
    const LinOperand<BKJ, VEC, SLOTS> opB{Bp, jb.ldb, N, K, n0, tid};
    // register prefetch two chunks ahead: one chunk's MFMA work (~0.85 us) is shorter than a cold round trip to L2 / HBM.
    // Two named tile pairs (a dynamically indexed register array would go to scratch); a load past K is clamped and unused.
    LinTile<SLOTS> a0 = opA.load(0), b0 = opB.load(0), a1 = opA.load(LIN_KC), b1 = opB.load(LIN_KC);
    auto chunk = [&](const int k0, LinTile<SLOTS>& ta, LinTile<SLOTS>& tb) __attribute__((always_inline)) {
        __syncthreads();                       // everybody is done with the previous chunk's tiles
        lin_stage(sA, ta, tid);
        lin_stage(sB, tb, tid);
        __syncthreads();
        ta = opA.load(k0 + 2 * LIN_KC); tb = opB.load(k0 + 2 * LIN_KC);
        // fragments of sub-chunk kc + 1 are read while the 16 MFMAs of sub-chunk kc issue; the MFMAs of one k-step go round the
        // four accumulators, so consecutive ones are independent
        f32x4 fa = lin_frag<AKJ>(sA, 16 * wave + li, 0, q), fb[T];
#pragma unroll
        for (int t = 0; t < T; ++t) fb[t] = lin_frag<BKJ>(sB, 16 * (T * wn + t) + li, 0, q);
#pragma unroll
        for (int kc = 0; kc < LIN_KC / 16; ++kc) {
            const f32x4 a = fa;
            f32x4 b[T];
#pragma unroll
            for (int t = 0; t < T; ++t) b[t] = fb[t];
            if (kc + 1 < LIN_KC / 16) {
                fa = lin_frag<AKJ>(sA, 16 * wave + li, kc + 1, q);
#pragma unroll
                for (int t = 0; t < T; ++t) fb[t] = lin_frag<BKJ>(sB, 16 * (T * wn + t) + li, kc + 1, q);
            }
            if (k0 + 16 * kc < K) {            // uniform; the rest of the chunk is zeros
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[t] = mfma_16x16x4(a[s], b[t][s], acc[t]);
                if (FORM == LIN_W) {
                    if (with_bias && bias_role && tid < 64) {   // ascending batch row; the 16 LDS reads of a sub-chunk issued together
                        float v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[u] = sA[(16 * kc + u) * LIN_LD + tid];
#pragma unroll
                        for (int u = 0; u < 16; ++u) bsum += v[u];
                    }
                }
            }
        }
    };
    for (int k0 = 0; k0 < K; k0 += 2 * LIN_KC) {       // both halves unconditionally: a load inside a branch makes the compiler wait for
        chunk(k0, a0, b0);                             // every outstanding load at the join; a half past K stages zeros and skips
        chunk(k0 + LIN_KC, a1, b1);                    // its MFMAs
    }
    };
    k_pass(jb.A, jb.B, true);
    if (FORM == LIN_W) {
        if (jb.A2) k_pass(jb.A2, jb.B2, false);     // uniform
    }
    // ---- epilogue: acc[t][r] = C[m0 + 16 wave + 4 q + r][n0 + 16 t + li]
    float sq = 0.0f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int n = n0 + 16 * (T * wn + t) + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * wave + 4 * q + r;
            if (m < M && n < N) {
                float v = acc[t][r];
                if (FORM == LIN_F) {
                    if (jb.acc) v += jb.acc[(size_t)m * jb.ldacc + n];
                    if (jb.aux) v += jb.aux[n];
                    if (jb.relu) v = fmaxf(v, 0.0f);
                    if (jb.mask) v = (jb.mask[(size_t)m * jb.ldmask + n] > 0.0f) ? v : 0.0f;
                } else if (FORM == LIN_X) {
                    if (jb.acc) v += jb.acc[(size_t)m * jb.ldacc + n];
                    if (jb.aux) v = (jb.aux[(size_t)m * jb.ldaux + n] > 0.0f) ? v : 0.0f;
                } else {
                    sq = fmaf(v, v, sq);
                }
                jb.C[(size_t)m * jb.ldc + n] = v;
            }
        }
    }
    if (FORM == LIN_W) {
        if (bias_role && tid < 64 && m0 + tid < M) {
            jb.bias_out[m0 + tid] = bsum;
            sq = fmaf(bsum, bsum, sq);
        }
        if (gsq_part) {
            sq = wave_sum(sq);
            if (lane == 0) red[wv] = sq;
            __syncthreads();
            if (tid == 0) {
                float tot = 0.0f;
#pragma unroll
                for (int g = 0; g < NW; ++g) tot += (red[4 * g] + red[4 * g + 1]) + (red[4 * g + 2] + red[4 * g + 3]);
                gsq_part[blk] = tot;
            }
        }
    }
}

// ---------------------------------------------------------------- loss heads of one minibatch step
// grid = (ceil(mb_size / 16), n_nets), 256 threads = (row i = tid >> 4, action dim d = tid & 15): the head arithmetic of
// ppo_fwd_bwd_body (kernels_mlp.hpp; fsrl/policy/ppo_lag.py:152-212, lagrangian_base.py:145-166) on head outputs that a
// lin_kernel<LIN_F> launch left in `out` ([net][mbp][16]).  Writes dL/d(head) | dL/d(log sigma) rows to `dout`
// ([net][mbp][FSRL_DOW]) and the per-16-row partial sums of the logged quantities to statp, like the fused kernel.
struct LayHeadArgs {
    const float* out; float* dout; const float* rd; float* statp;
    const float* P; int sigma;      // the actor's sigma_param inside P
    int mbp, n_nets, Da, unbounded;
};
__global__ __launch_bounds__(256) void lay_ppo_head_kernel(const LayHeadArgs h, const PpoStepArgs sa) {
    __shared__ float st[16 * 4];
    const int tid = threadIdx.x, i = tid >> 4, d = tid & 15, lane = tid & 63;
    const int net = blockIdx.y, tile = blockIdx.x, C = h.n_nets - 1, Da = h.Da;
    const int row = tile * 16 + i;
    const bool valid = row < sa.mb_size;
    const float* rd = h.rd + (size_t)(sa.mb_start + (valid ? row : 0)) * FSRL_RD;
    const float* o = h.out + ((size_t)net * h.mbp + (valid ? row : 0)) * FSRL_MAX_ACT;
    float* dO = h.dout + ((size_t)net * h.mbp + row) * FSRL_DOW;
    const float invB = 1.0f / (float)sa.mb_size;
    float st0 = 0.f, st1 = 0.f, st2 = 0.f;
    if (net == 0) {
        float th = 0.f, var = 1.f, df = 0.f, lp = 0.f;
        float hs = sa.max_action;
        if (d < Da) {
            const float x = o[d];
            th = tanhf(x);
            const float sig = expf(h.P[h.sigma + d]);
            var = sig * sig;
            df = rd[d] - sa.max_action * th;
            if (h.unbounded) { df = rd[d] - x; th = 0.0f; hs = 1.0f; }
            lp = -(df * df) / (2.0f * var) - logf(sig) - LOG_SQRT_2PI;
        }
        float logp = 0.0f;
        for (int dd = 0; dd < Da; ++dd) logp += __shfl(lp, (lane & 48) + dd, 64);
        const float lpo = rd[FSRL_RD_LOGP];
        const float ratio = expf(logp - lpo);
        const float ar = rd[FSRL_RD_ADV];
        const float s1 = ratio * ar;
        const float rc = fminf(fmaxf(ratio, 1.0f - sa.eps_clip), 1.0f + sa.eps_clip);
        const float s2 = rc * ar;
        const bool inrange = (ratio >= 1.0f - sa.eps_clip) && (ratio <= 1.0f + sa.eps_clip);
        const float g_c1 = inrange ? ar : (s1 < s2 ? ar : (s1 == s2 ? 0.5f * ar : 0.0f));
        float term = fminf(s1, s2);
        float g_term = g_c1;
        if (sa.dual_clip > 0.0f) {
            const float c1 = term;
            const float lim = sa.dual_clip * ar;
            if (ar < 0.0f) {
                term = fmaxf(c1, lim);
                g_term = (c1 > lim) ? g_c1 : (c1 == lim ? 0.5f * g_c1 : 0.0f);
            }
        }
        float dL_dratio = -g_term * invB;
        float safety_sum = 0.0f;
        if (sa.use_lagrangian) {
#pragma unroll
            for (int c = 1; c < FSRL_MAX_CRITICS; ++c) {
                if (c < C) {
                    const float ac = rd[FSRL_RD_ADV + c];
                    dL_dratio += sa.lam[c - 1] * ac * invB;
                    safety_sum += ratio * ac * sa.lam[c - 1];
                }
            }
        }
        const float dL_dlogp = sa.rescale * dL_dratio * ratio;
        if (valid) {
            dO[d] = (d < Da) ? dL_dlogp * (df / var) * hs * (1.0f - th * th) : 0.0f;
            dO[16 + d] = (d < Da) ? dL_dlogp * (df * df / var - 1.0f) : 0.0f;
            st0 = term; st1 = safety_sum; st2 = lpo - logp;
        }
    } else {
        const int c = net - 1;
        const float v = o[0];
        const float dd = rd[FSRL_RD_RET + c] - v;
        float g = -2.0f * dd, vf = dd * dd;
        if (sa.value_clip) {
            const float vo = rd[FSRL_RD_VOLD + c];
            const float dv = v - vo;
            const float vc = vo + fminf(fmaxf(dv, -sa.eps_clip), sa.eps_clip);
            const float d2 = rd[FSRL_RD_RET + c] - vc;
            const float vf2 = d2 * d2;
            const float g2 = (dv >= -sa.eps_clip && dv <= sa.eps_clip) ? -2.0f * d2 : 0.0f;
            g = (vf > vf2) ? g : (vf == vf2 ? 0.5f * g + 0.5f * g2 : g2);
            vf = fmaxf(vf, vf2);
        }
        if (valid) {
            dO[d] = (d == 0) ? sa.vf_coef * g * invB : 0.0f;
            dO[16 + d] = 0.0f;
            st0 = vf;
        }
    }
    if (d == 0) { st[i * 4 + 0] = st0; st[i * 4 + 1] = st1; st[i * 4 + 2] = st2; }
    __syncthreads();
    if (tid < 4) {       // rows summed in ascending order
        float t = 0.0f;
        if (tid < 3)
            for (int r = 0; r < 16; ++r) t += st[r * 4 + tid];
        h.statp[((size_t)tile * h.n_nets + net) * 4 + tid] = t;
    }
}

// ---------------------------------------------------------------- heads of the full-batch family (FOCOPS step, CPO / TRPO-Lag)
// The head arithmetic of fb_tile_body (kernels_fb.hpp) on head outputs a lin_kernel<LIN_F> launch left in `out`: actor modes
// FB_MODE_SUR / _KL / _FOCOPS / _EVAL (cpo.py:184-214, trpo_lag.py:134-171, focops.py:161-203), the plain value regression on
// the critics.  grid = (ceil(N / 16), ny): network net0 + blockIdx.y; statistics slots [tile][ny][FB_NSTAT] like fb_tile_kernel.
struct LayFbHeadArgs {
    const float* out; float* dout; const float* rd; float* statp;      // rd: row data of THESE rows (already offset)
    const float* P; int sigma;
    int mbp, net0, ny, Da, unbounded, N, mode;
    float max_action, cr, cc, eta;
};
__global__ __launch_bounds__(256) void lay_fb_head_kernel(const LayFbHeadArgs h) {
    __shared__ float stl[16 * FB_NSTAT];
    const int tid = threadIdx.x, i = tid >> 4, d = tid & 15, lane = tid & 63;
    const int y = blockIdx.y, net = h.net0 + y, tile = blockIdx.x, Da = h.Da;
    const int row = tile * 16 + i;
    const bool valid = row < h.N;
    const bool backward = h.mode != FB_MODE_EVAL;
    const float* rd = h.rd + (size_t)(valid ? row : 0) * FSRL_RD;
    const float* o = h.out + ((size_t)net * h.mbp + (valid ? row : 0)) * FSRL_MAX_ACT;
    float* dO = h.dout + ((size_t)net * h.mbp + row) * FSRL_DOW;
    const float invN = 1.0f / (float)h.N;
    float st[FB_NSTAT];
#pragma unroll
    for (int k = 0; k < FB_NSTAT; ++k) st[k] = 0.0f;
    if (net == 0) {
        float th = 0.f, var = 1.f, df = 0.f, lp = 0.f, klp = 0.f, dmu = 0.f, so2 = 0.f;
        float hs = h.max_action;
        if (d < Da) {
            const float x = o[d];
            th = tanhf(x);
            const float sig = expf(h.P[h.sigma + d]);
            var = sig * sig;
            const float mu = h.max_action * th;
            df = rd[d] - mu;
            dmu = mu - rd[FSRL_RD_MEAN + d];
            if (h.unbounded) { df = rd[d] - x; dmu = x - rd[FSRL_RD_MEAN + d]; th = 0.0f; hs = 1.0f; }
            lp = -(df * df) / (2.0f * var) - logf(sig) - LOG_SQRT_2PI;
            // KL(old || new), torch.distributions.kl._kl_normal_normal
            const float so = rd[FSRL_RD_STD + d];
            so2 = so * so;
            const float var_ratio = (so / sig) * (so / sig);
            const float t1 = (dmu / sig) * (dmu / sig);
            klp = 0.5f * (var_ratio + t1 - 1.0f - logf(var_ratio));
            if (h.mode == FB_MODE_FOCOPS) {          // KL(new || old): _kl_normal_normal(p = new, q = old)
                const float vr = (sig / so) * (sig / so);
                const float t1n = (dmu / so) * (dmu / so);
                klp = 0.5f * (vr + t1n - 1.0f - logf(vr));
                so2 = vr;                             // d KL / d log sigma_new = vr - 1
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
        if (valid && backward) {
            float g0 = 0.0f, g1 = 0.0f;
            if (d < Da) {
                if (h.mode == FB_MODE_SUR) {
                    const float dL_dlogp = (h.cr * ar + h.cc * ac) * ratio * invN;
                    g0 = dL_dlogp * (df / var) * hs * (1.0f - th * th);
                    g1 = dL_dlogp * (df * df / var - 1.0f);
                } else if (h.mode == FB_MODE_KL) {
                    g0 = (dmu / var) * invN * hs * (1.0f - th * th);
                    g1 = (1.0f - (so2 + dmu * dmu) / var) * invN;
                } else {                              // FB_MODE_FOCOPS: (KL - cr * ratio * (A_r - cc * A_c)) * mask ; cr = 1 / lambda, cc = nu
                    const float mask = (klrow <= h.eta) ? invN : 0.0f;
                    const float dL_dlogp = -h.cr * (ar - h.cc * ac) * ratio;
                    g0 = (dL_dlogp * (df / var) + dmu) * hs * (1.0f - th * th) * mask;
                    g1 = (dL_dlogp * (df * df / var - 1.0f) + (so2 - 1.0f)) * mask;
                }
            }
            dO[d] = g0; dO[16 + d] = g1;
        }
        if (valid) {
            st[0] = ratio * ar; st[1] = ratio * ac; st[2] = klrow; st[3] = lpo - logp; st[4] = ar; st[5] = ac;
            if (h.mode == FB_MODE_FOCOPS) st[0] = (klrow <= h.eta) ? (klrow - h.cr * ratio * (ar - h.cc * ac)) : 0.0f;
        }
    } else {
        const int c = net - 1;
        const float dd = rd[FSRL_RD_RET + c] - o[0];
        if (valid) {
            if (backward) { dO[d] = (d == 0) ? -2.0f * dd * invN : 0.0f; dO[16 + d] = 0.0f; }
            st[0] = dd * dd;
        }
    }
    if (d == 0) {
#pragma unroll
        for (int k = 0; k < FB_NSTAT; ++k) stl[i * FB_NSTAT + k] = st[k];
    }
    __syncthreads();
    if (tid < FB_NSTAT) {        // rows summed in ascending order
        float t = 0.0f;
        for (int r = 0; r < 16; ++r) t += stl[r * FB_NSTAT + tid];
        h.statp[((size_t)tile * h.ny + y) * FB_NSTAT + tid] = t;
    }
}

// KL head of a Hessian-vector product (the head of fb_hvp_body, kernels_fb.hpp; cpo.py:169-182): from the head outputs x and their
// tangent R{x} (`rout`), the tangent of log sigma (v's sigma_param block) -> dout = dKLbar / d(head | log sigma) and R{dout}.
// grid = ceil(N / 16), 256 threads = (row, action dim).
struct LayHvpHeadArgs {
    const float* out; const float* rout; float* dout; float* rdout; const float* rd;
    const float* P; const float* V; int sigma;
    int Da, unbounded, N;
    float max_action;
};
__global__ __launch_bounds__(256) void lay_hvp_head_kernel(const LayHvpHeadArgs h) {
    const int tid = threadIdx.x, i = tid >> 4, d = tid & 15;
    const int row = blockIdx.x * 16 + i;
    if (row >= h.N) return;
    float g0 = 0.f, g1 = 0.f, r0 = 0.f, r1 = 0.f;
    if (d < h.Da) {
        const float invN = 1.0f / (float)h.N;
        const float* rd = h.rd + (size_t)row * FSRL_RD;
        const float x = h.out[(size_t)row * FSRL_MAX_ACT + d];
        const float t = h.unbounded ? 0.0f : tanhf(x);
        const float hs = h.unbounded ? 1.0f : h.max_action;
        const float ro = h.rout[(size_t)row * FSRL_MAX_ACT + d];
        const float sp = h.P[h.sigma + d], rls = h.V[h.sigma + d];      // R{log sigma} = v_sigma
        const float sig = expf(sp), var = sig * sig;
        const float dt = hs * (1.0f - t * t);                           // dmu / dout
        const float rmu = dt * ro;
        const float dmu_b = h.max_action * t - rd[FSRL_RD_MEAN + d];
        const float dmu = h.unbounded ? x - rd[FSRL_RD_MEAN + d] : dmu_b;
        const float so = rd[FSRL_RD_STD + d], so2 = so * so;
        const float gmu = dmu / var;                                    // dKL / dmu
        const float rgmu = rmu / var - 2.0f * gmu * rls;
        const float rgls = -2.0f * dmu * rmu / var + 2.0f * (so2 + dmu * dmu) / var * rls;
        const float rdt = hs * (-2.0f * t) * (1.0f - t * t) * ro;       // R{dmu / dout}
        g0 = invN * gmu * dt;
        r0 = invN * (rgmu * dt + gmu * rdt);
        g1 = invN * (1.0f - (so2 + dmu * dmu) / var);
        r1 = invN * rgls;
    }
    h.dout[(size_t)row * FSRL_DOW + d] = g0; h.dout[(size_t)row * FSRL_DOW + 16 + d] = g1;
    h.rdout[(size_t)row * FSRL_DOW + d] = r0; h.rdout[(size_t)row * FSRL_DOW + 16 + d] = r1;
}

// ---------------------------------------------------------------- process_fn / collector inference: the tail of mlp_infer_kernel
// grid = (ceil(N / 16), jobs), 64 threads: lane r < 16 finishes row 16 blockIdx.x + r of the job from its head outputs
// (`out`: [job][N][16]).  Job numbering and semantics are InferArgs' (kernels_mlp.hpp).
__global__ __launch_bounds__(64) void lay_infer_out_kernel(const float* __restrict__ out, const float* __restrict__ P,
                                                          const int sigma, const int Da, const int unbounded,
                                                          const InferArgs a) {
    const int job = blockIdx.y, C = a.C, tid = threadIdx.x;
    const bool is_actor = (job == 2 * C);
    const bool use_next = (!is_actor) && job >= C;
    const int r = blockIdx.x * 16 + tid;
    if (a.sigma_param_out && is_actor && blockIdx.x == 0 && tid < Da) a.sigma_param_out[tid] = P[sigma + tid];
    if (tid < 16 && r < a.N) {
        const float* o = out + ((size_t)job * a.N + r) * FSRL_MAX_ACT;
        if (!is_actor) {
            float v = o[0];
            const int c = (C > 0) ? job % C : 0;
            if (use_next) {
                if (a.flags[r] & 1) v = 0.0f;
                a.vnext[(size_t)c * a.N + r] = v;
            } else {
                a.values[(size_t)c * a.N + r] = v;
            }
        } else {
            float logp = 0.0f;
            for (int d = 0; d < Da; ++d) {
                const float x = o[d];
                const float mu = unbounded ? x : a.max_action * tanhf(x);
                const float sig = expf(P[sigma + d]);
                if (a.mu_out) a.mu_out[(size_t)r * Da + d] = mu;
                if (a.act) {
                    const float diff = a.act[(size_t)r * Da + d] - mu;
                    logp += -(diff * diff) / (2.0f * sig * sig) - logf(sig) - LOG_SQRT_2PI;
                }
            }
            if (a.logp_old) a.logp_old[r] = logp;
        }
    }
    if (a.done) {
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.done + blockIdx.x, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
