// kernels_wgrad3.hpp -- the weight-side products of the full-batch passes, third form (round 6, the default at 256 wide):
// every workgroup is a 64 x 64 MFMA tile job, 512 threads, two workgroups per CU.
//
// Reference: the gradients autograd forms for cpo.py:147-162 (critics), :206-220 (_get_flat_grad), :177-182 (_MVP) and
// trpo_lag.py:234-259 -- dW2 = dz2^T h1, dW1 = dz1^T x, dW3 = dout^T h2, the bias sums, and their R-op twins.
//
// What round 5's counters said about fb_wgrad_kernel (kernels_fb.hpp; 43 % MFMA-busy, 28.5 % of a trust-region update): a
// launch is ONE round of 250 1024-thread workgroups, 160 of them dW2 tiles (16 MFMAs per k-step and wave: the matrix pipe of
// their CU is theirs, 53 us of MFMA time at N = 20 000) and 90 "aux" workgroups (dW1 / dW3 / bias sums: 1/16 of the FLOPs,
// bound by scalar observation loads, as long as a tile workgroup) -- 35 % of the CUs carry almost no matrix work, and nothing
// shares a CU with a 1024-thread, > 100-VGPR workgroup.  Here
//   * dW1 is a tile job like dW2: the observations are re-laid ONCE per batch view into `obs_pad` ([rows][64 per column group],
//     zero-filled, column 16 u + c of a group at position 4 c + u), so a lane loads one float4 of them per k-step and feeds
//     MFMAs per 16-column chunk -- a narrow observation (8 columns: one chunk) puts all four tile rows of hidden units in ONE
//     job, so a dW1 job is 12-16 MFMAs per k-step like a dW2 job; db1 rides with the dW1 jobs, db2 with the dW2 jobs of tile
//     column 0; dW3 (two jobs of 128 hidden units, 8 MFMAs per k-step), db3 and dsigma ride together.  The second operand
//     pair of an R-op launch is 16 + 2 MORE jobs with
//     partial slots of their own, not twice the work per job: every workgroup of a launch carries about the same matrix work
//     (first version, jobs of 32 / 16 / 8 MFMAs per k-step: the CUs holding two 32-MFMA jobs set the launch time, 123 k MFMA
//     cycles per SIMD against 96 k on average -- same time as the kernel it replaced);
//   * 512 threads (8-way split-K over the waves, the operands of the next TWO k-steps in flight per wave) and <= 128 VGPRs: two
//     workgroups per CU, so the critic lane's launches and the actor's share CUs throughout (the lane of round 5 overlapped only
//     with the tile kernels), and twice the row splits fill the 2 x 256 slots in one round;
//   * the workgroups of one (network, split) -- which stream the same rows -- sit behind ONE L2 (hardware block L runs on XCD
//     L % 8: observed placement, used for speed only), so each 20 MB operand array crosses the fabric once, not four times.
// Partial gradients per split at out + z * split_stride, added by the consumers in z order (float64, fixed order), as before.
#pragma once
#include "kernels_fb.hpp"

#define WG3_SLOT (64 * 65)            // one 64 x 64 partial tile (+1 column of padding)
#define WG3_EXT 1536                  // floats behind the four slots: [8 waves][64] db2 or dout column sums | [8][128] db1

// observations -> obs_pad: out[r][64 g + 4 c + u] = obs[r][64 g + 16 u + c] (0 beyond N rows / Do columns), r < rows
__global__ __launch_bounds__(256) void fb_obs_pad_kernel(const float* __restrict__ obs, float* __restrict__ out, const int N,
                                                        const int Do, const int rows, const int KO) {
    const int W = 64 * KO;
    const size_t total = (size_t)rows * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / W), p = (int)(i % W);
        const int g = p >> 6, cc = (p >> 2) & 15, u = p & 3;
        const int col = 64 * g + 16 * u + cc;
        out[i] = (r < N && col < Do) ? obs[(size_t)r * Do + col] : 0.0f;
    }
}

// partial 64 x 64 tiles of the 8 waves -> four LDS slots in two rounds (waves 0-3 store, waves 4-7 add); the caller sums the slots.
// ORIG: column index 16 u + c (the dW1 jobs: obs_pad's order back to the observation's), else 4 c + u.
template <bool ORIG>
__device__ __forceinline__ void wg3_tile_to_lds(const f32x4 (&acc)[4][4], float* red, const int wave, const int c, const int q) {
    float* myred = red + (wave & 3) * WG3_SLOT;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if ((wave >> 2) == round) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* p = &myred[(4 * (4 * q + r) + t) * 65];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = ORIG ? 16 * u + c : 4 * c + u;
                        if (round == 0) p[k] = acc[t][u][r];
                        else p[k] += acc[t][u][r];
                    }
                }
        }
        __syncthreads();
    }
}

// The k-step loop of every job: wave w of the workgroup owns k-steps KS0 + w, + 8, + 16 ... (ascending: the accumulation order
// does not depend on the pipelining); three operand sets in rotation, so the loads of the wave's next TWO k-steps are in flight
// while the MFMAs of the current one issue.  fetch(k-step, set) / fma(set) are the job's.  Every fetch is UNCONDITIONAL (the
// k-step index is clamped to the wave's last one: the tail re-reads a row it already has): with loads behind branches the
// compiler cannot count them and drains the queue (`s_waitcnt vmcnt(0)`) before every use -- the first version waited for the
// loads it had just issued, once per k-step.
template <class Set, class Fetch, class Fma>
__device__ __forceinline__ void wg3_pipeline(const int KS0, const int KS, const int wave, Set& A, Set& B, Set& C, Fetch&& fetch, Fma&& fma,
                                             const int probe) {
    constexpr int NW = 8;
    const int first = KS0 + wave;
    const int n = first < KS ? (KS - first + NW - 1) / NW : 0;           // this wave's k-steps
    if (n <= 0) return;
#ifdef FSRL_PROBES
    if (probe & 64) return;                                              // timing probe: prologue + epilogue only
    const int last = (probe & 8) ? 0 : n - 1;                            // timing probe: every fetch re-reads the wave's FIRST k-step (results invalid)
#else
    const int last = n - 1;
#endif
    fetch(first, A);
    fetch(first + min(1, last) * NW, B);
    int i = 0;
    for (; i + 3 <= n; i += 3) {
        fetch(first + min(i + 2, last) * NW, C);
        fma(A);
        fetch(first + min(i + 3, last) * NW, A);
        fma(B);
        fetch(first + min(i + 4, last) * NW, B);
        fma(C);
    }
    if (i < n) fma(A);
    if (i + 1 < n) fma(B);
}

// the same with two operand sets (one k-step ahead): jobs whose accumulators + three sets do not fit 128 VGPRs
template <class Set, class Fetch, class Fma>
__device__ __forceinline__ void wg3_pipeline2(const int KS0, const int KS, const int wave, Set& A, Set& B, Fetch&& fetch, Fma&& fma,
                                              const int probe) {
    constexpr int NW = 8;
    const int first = KS0 + wave;
    const int n = first < KS ? (KS - first + NW - 1) / NW : 0;
    if (n <= 0) return;
#ifdef FSRL_PROBES
    if (probe & 64) return;
    const int last = (probe & 8) ? 0 : n - 1;
#else
    const int last = n - 1;
#endif
    fetch(first, A);
    int i = 0;
    for (; i + 2 <= n; i += 2) {
        fetch(first + min(i + 1, last) * NW, B);
        fma(A);
        fetch(first + min(i + 2, last) * NW, A);
        fma(B);
    }
    if (i < n) fma(A);
}

// sum of the four LDS slots of element e (fixed order)
__device__ __forceinline__ float wg3_sum4(const float* red, const int e) {
    return (red[e] + red[WG3_SLOT + e]) + (red[2 * WG3_SLOT + e] + red[3 * WG3_SLOT + e]);
}

// NP = 1 + (the launch has a second operand pair).  Jobs of one (network, split), all 8-way split-K over the waves and all of
// (about) 16 MFMAs per k-step, so that any two of them share a CU evenly:
//   T(p, tj, tk)      16 x NP   dW2 tile of pair p: out[j][k] = sum_r Y_p[r][64 tj + j] X_p[r][64 tk + k]
//   U(ko, part)    <= 4 x KO    dW1 for the 64-column group ko of the observations (NU = its 16-column chunks that exist):
//                               TJ = 4 / NU' tile rows of hidden units per job (NU' = 1, 2, 4 >= NU), 4 / TJ jobs
//   V(p, half)         2 x NP   dW3 of pair p for 128 hidden units: h2_p[:, j] x dout_p[:, o] (8 MFMAs per k-step); the jobs of pair a also db1 of their
//                               hidden units, the first db3 / dsigma
// A second pair gets partial slots of its own (z + nsplit): a T job is the same work whichever pair it belongs to, and the
// consumers simply add NP x nsplit partials.  What pair b does not produce (W1, b1, b2, b3, sigma) is written as zeros into its
// slots by the jobs that own those outputs.
struct Wg3Ctx {
    const FbWgradNet* wn; const NetOff* no; const FbWgradArgs* wa;
    float* red; float* ext; float* gout; float* gout_b;
    int KS0, KS, tid, wave, c, q, NP, Do;
};

template <int H>
__device__ __forceinline__ void wg3_job_T(const Wg3Ctx& k, const int pr, const int tj, const int tk) {
    constexpr int NW = 8;
    const FbWgradNet& wn = *k.wn; const NetOff& no = *k.no;
    const int c = k.c, q = k.q, wave = k.wave, tid = k.tid;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // lane (c, q) holds four columns of row 4 s + q of each operand; acc[t][u][r] = output (j = 4 (4 q + r) + t, k = 4 c + u)
    const Wg3Buf by = wg3_buf((pr ? wn.w2_yb : wn.w2_ya) + tj * 64), bx = wg3_buf((pr ? wn.w2_xb : wn.w2_xa) + tk * 64);
    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = zero4;
    f32x4 sy = zero4;                                        // column sums of Y: db2 (pair a, tile column 0: b2_src == w2_ya)
    struct Set { f32x4 y, x; };
    auto fetch = [&](const int s, Set& o) {
        const unsigned r = (unsigned)(4 * s + q) * (unsigned)H + 4u * (unsigned)c;   // 32-bit offsets: rows * 256 floats stays far below 2^32
        o.y = wg3_ld4(by, r);
        o.x = wg3_ld4(bx, r);
    };
    auto fma = [&](const Set& o) {
#ifdef FSRL_PROBES
        if (k.wa->dbg_skip & 16) { sy += o.y + o.x; return; }           // timing probe: no matrix work (results invalid)
#endif
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t][u] = mfma_16x16x4(o.y[t], o.x[u], acc[t][u]);
        sy += o.y;
        // pin the sum HERE: left alone, the scheduler sinks these adds to the end of the loop body, which keeps every set's y alive
        // past its re-fetch -- the re-fetched sets land in other registers and are copied back behind `s_waitcnt vmcnt(1)`
        // once per three k-steps (the whole prefetch drained: 69.5 -> 63.2 us per launch)
        asm volatile("" : "+v"(sy));
    };
    {
        Set A, B, C;
        A.y = A.x = B.y = B.x = C.y = C.x = zero4;
        wg3_pipeline(k.KS0, k.KS, wave, A, B, C, fetch, fma, k.wa->dbg_skip);
    }
    if (tk == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sy[t] += __shfl_xor(sy[t], 16, 64);
            sy[t] += __shfl_xor(sy[t], 32, 64);
        }
        if (q == 0) *reinterpret_cast<f32x4*>(k.ext + wave * 64 + 4 * c) = sy;
    }
    wg3_tile_to_lds<false>(acc, k.red, wave, c, q);
    float* __restrict__ go = pr ? k.gout_b : k.gout;
#pragma unroll
    for (int e0 = 0; e0 < 4096; e0 += 512) {
        const int e = e0 + tid, jl = e >> 6, kl = e & 63;
        go[no.W2 + (size_t)(tj * 64 + jl) * H + tk * 64 + kl] = wg3_sum4(k.red, jl * 65 + kl);
    }
    if (tk == 0 && tid < 64) {
        float b = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) b += k.ext[w * 64 + tid];
        go[no.b2 + tj * 64 + tid] = pr ? 0.0f : b;           // pair b has no bias term
    }
}

// dW1 job: TJ tile rows of hidden units (tj0 ..) x the NU (<= 4 / TJ) 16-column chunks of observation group ko.
// acc[t][w][r], w = tjl * (4 / TJ) + u: output (j = 64 (tj0 + tjl) + 4 (4 q + r) + t, k = 64 ko + 16 u + c)
template <int H, int TJ>
__device__ __forceinline__ void wg3_job_U(const Wg3Ctx& k, const int tj0, const int ko, const int NU) {
    constexpr int NW = 8, NUP = 4 / TJ;
    const FbWgradNet& wn = *k.wn; const NetOff& no = *k.no;
    const int c = k.c, q = k.q, wave = k.wave, tid = k.tid, Do = k.Do;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const Wg3Buf by = wg3_buf(wn.w1_y + tj0 * 64), bx = wg3_buf(k.wa->obs_pad + (size_t)64 * ko);
    const unsigned ldx = 64u * (unsigned)k.wa->obs_ko;
    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int w = 0; w < 4; ++w) acc[t][w] = zero4;
    struct Set { f32x4 y[TJ]; f32x4 x; };
    auto fetch = [&](const int s, Set& o) {
        const unsigned r = (unsigned)(4 * s + q);
#pragma unroll
        for (int j = 0; j < TJ; ++j) o.y[j] = wg3_ld4(by, r * (unsigned)H + 64u * j + 4u * (unsigned)c);
        o.x = wg3_ld4(bx, r * ldx + 4u * (unsigned)c);
    };
    auto fma = [&](const Set& o) {
#ifdef FSRL_PROBES
        if (k.wa->dbg_skip & 16) { acc[0][0] += o.y[0] + o.x; return; }
#endif
#pragma unroll
        for (int u = 0; u < NUP; ++u) {
            if (u < NU) {                                    // block-uniform (NU = 3 of 4)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t][j * NUP + u] = mfma_16x16x4(o.y[j][t], o.x[u], acc[t][j * NUP + u]);
            }
        }
    };
    {
        Set A, B, C;
#pragma unroll
        for (int j = 0; j < TJ; ++j) A.y[j] = B.y[j] = C.y[j] = zero4;
        A.x = B.x = C.x = zero4;
        if constexpr (TJ == 4) wg3_pipeline2(k.KS0, k.KS, wave, A, B, fetch, fma, k.wa->dbg_skip);      // 64 + 4 x 20 registers do not fit
        else wg3_pipeline(k.KS0, k.KS, wave, A, B, C, fetch, fma, k.wa->dbg_skip);
    }
    wg3_tile_to_lds<true>(acc, k.red, wave, c, q);
#pragma unroll
    for (int e0 = 0; e0 < 4096; e0 += 512) {
        const int e = e0 + tid, jl = e >> 6, kl = e & 63;
        const int w = kl >> 4, cc = kl & 15, tjl = w / NUP, u = w % NUP;
        const int kcol = 64 * ko + 16 * u + cc;
        const float v = wg3_sum4(k.red, jl * 65 + kl);
        if (u < NU && kcol < Do) {
            const size_t o1 = no.W1 + (size_t)((tj0 + tjl) * 64 + jl) * Do + kcol;
            k.gout[o1] = v;
            if (k.NP == 2) k.gout_b[o1] = 0.0f;
        }
    }
}

// dW3 job of pair pr for 128 hidden units (half = 0 / 1): ad[g][t][r] = output (j = 128 half + 64 g + 4 (4 q + r) + t, o = c); the
// first job of pair a also carries the column sums of its dout-like rows: db3 (columns 0 .. 15) and dsigma (16 .. 31); do_src ==
// w3_ya in every caller.  (One job for all 256 hidden units: 64 accumulators + its operand sets + six load addresses spill.)
template <int H>
__device__ __forceinline__ void wg3_job_V(const Wg3Ctx& k, const int pr, const int half, const int Da) {
    constexpr int NW = 8;
    const FbWgradNet& wn = *k.wn; const NetOff& no = *k.no;
    const int c = k.c, q = k.q, wave = k.wave, tid = k.tid, out = no.out;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bool head = pr == 0 && half == 0;
    f32x4 ad[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) ad[g][t] = zero4;
    float s3a = 0.0f, s3b = 0.0f;
    f32x4 s1[2] = {zero4, zero4};                            // db1: column sums of the dW1 operand (b1_src == w1_y in every caller)
    const Wg3Buf bx3 = wg3_buf((pr ? wn.w3_xb : wn.w3_xa) + 128 * half), bd3 = wg3_buf(pr ? wn.w3_yb : wn.w3_ya),
                 by1 = wg3_buf(wn.w1_y + 128 * half);
    struct VSet { f32x4 x[2], y1[2]; float d, d2; };
    auto vfetch = [&](const int s, VSet& o) {
        const unsigned r = (unsigned)(4 * s + q);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            o.x[g] = wg3_ld4(bx3, r * (unsigned)H + 64u * g + 4u * (unsigned)c);
            o.y1[g] = wg3_ld4(by1, r * (unsigned)H + 64u * g + 4u * (unsigned)c);
        }
        o.d = wg3_ld1(bd3, r * (unsigned)FSRL_DOW + (unsigned)c);
        o.d2 = wg3_ld1(bd3, r * (unsigned)FSRL_DOW + 16u + (unsigned)c);
    };
    auto vfma = [&](const VSet& o) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) ad[g][t] = mfma_16x16x4(o.x[g][t], o.d, ad[g][t]);
        s3a += o.d;
        s3b += o.d2;
        s1[0] += o.y1[0]; s1[1] += o.y1[1];
        asm volatile("" : "+v"(s3a), "+v"(s3b), "+v"(s1[0]), "+v"(s1[1]));       // see wg3_job_T
    };
    {
        VSet A, B, C;
#pragma unroll
        for (int g = 0; g < 2; ++g) A.x[g] = B.x[g] = C.x[g] = A.y1[g] = B.y1[g] = C.y1[g] = zero4;
        A.d = A.d2 = B.d = B.d2 = C.d = C.d2 = 0.0f;
        wg3_pipeline(k.KS0, k.KS, wave, A, B, C, vfetch, vfma, k.wa->dbg_skip);
    }
    if (head) {
        s3a += __shfl_xor(s3a, 16, 64); s3a += __shfl_xor(s3a, 32, 64);
        s3b += __shfl_xor(s3b, 16, 64); s3b += __shfl_xor(s3b, 32, 64);
        if (q == 0) { k.ext[wave * 64 + c] = s3a; k.ext[wave * 64 + 16 + c] = s3b; }
    }
    if (pr == 0) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s1[g][t] += __shfl_xor(s1[g][t], 16, 64);
                s1[g][t] += __shfl_xor(s1[g][t], 32, 64);
            }
            if (q == 0) *reinterpret_cast<f32x4*>(k.ext + 512 + wave * 128 + 64 * g + 4 * c) = s1[g];
        }
    }
    // [128 hidden units][16 outputs] per slot (17 floats a row)
    float* myred = k.red + (wave & 3) * WG3_SLOT;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if ((wave >> 2) == round) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* p = &myred[(64 * g + 4 * (4 * q + r) + t) * 17 + c];
                        if (round == 0) *p = ad[g][t][r];
                        else *p += ad[g][t][r];
                    }
        }
        __syncthreads();
    }
    float* __restrict__ go = pr ? k.gout_b : k.gout;
#pragma unroll
    for (int e0 = 0; e0 < 128 * 16; e0 += 512) {
        const int e = e0 + tid, j = e >> 4, o = e & 15;
        if (o < out) go[no.W3 + (size_t)o * H + 128 * half + j] = wg3_sum4(k.red, j * 17 + o);
    }
    if (tid >= 64 && tid < 192) {                            // db1 of this job's 128 hidden units (zeros into pair b's slot)
        const int j = tid - 64;
        float b = 0.0f;
        if (pr == 0) {
#pragma unroll
            for (int w = 0; w < NW; ++w) b += k.ext[512 + w * 128 + j];
        }
        go[no.b1 + 128 * half + j] = b;
    }
    if (half == 0 && tid < 32) {
        float tot = 0.0f;
        if (head) {
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += k.ext[w * 64 + tid];
        }
        if (tid < out) go[no.b3 + tid] = tot;
        if (no.sigma >= 0 && tid >= 16 && tid < 16 + Da) go[no.sigma + tid - 16] = tot;
    }
}

// jobs of one (network, split) in block order: T jobs | per observation group its U jobs | V jobs
__host__ __device__ __forceinline__ int wg3_ujobs(const int Do, const int ko) {       // U jobs of observation group ko
    const int NU = min(4, (Do - 64 * ko + 15) >> 4);
    return NU <= 1 ? 1 : NU == 2 ? 2 : 4;
}

template <int H>
__global__ __launch_bounds__(512, 4) void fb_wgrad3_kernel(const ModelDesc md, const FbWgradArgs wa, const int NP, const int nsplit,
                                                          const int NB) {
    static_assert(H == 256, "fb_wgrad3_kernel is written for 256-wide layers (other widths keep fb_wgrad_kernel)");
    constexpr int TPD = H / 64, NT2 = TPD * TPD;
    __shared__ float red[4 * WG3_SLOT + WG3_EXT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int Lp = blockIdx.x;
    if (wa.wg3_flags & 1) {                                  // XCD-aware order (see the header)
        const int per = gridDim.x >> 3;
        Lp = (Lp & 7) * per + (Lp >> 3);
    }
    if (Lp >= wa.remap_total) return;
#ifdef FSRL_PROBES
    if (wa.dbg_skip & 32) return;                            // timing probe: launch + dispatch only
#endif
    int rb = Lp % NB;
    const int grp = Lp / NB;
    const int by = grp % wa.remap_ny, bz = grp / wa.remap_ny;
    const FbWgradNet wn = wa.nets[by];
    const NetOff no = md.net[wn.net];
    Wg3Ctx k;
    k.wn = &wn; k.no = &no; k.wa = &wa; k.red = red; k.ext = red + 4 * WG3_SLOT;
    k.KS0 = bz * wa.ks_per_split;
    k.KS = min(wa.rows >> 2, k.KS0 + wa.ks_per_split);      // this split's k-step range [KS0, KS): 4 rows per k-step
    k.gout = wa.out + (size_t)bz * wa.split_stride;
    k.gout_b = NP == 2 ? wa.out + (size_t)(bz + nsplit) * wa.split_stride : nullptr;      // pair b's slot
    k.tid = tid; k.wave = wave; k.c = lane & 15; k.q = lane >> 4; k.NP = NP; k.Do = md.Do;
    if (rb < NT2 * NP) {
        wg3_job_T<H>(k, rb / NT2, (rb % NT2) / TPD, rb % TPD);
        return;
    }
    rb -= NT2 * NP;
    int ko = 0, nj = 0;                                      // which observation group's dW1 job (the job itself runs OUTSIDE this loop)
    for (; ko < wa.obs_ko; ++ko) {
        nj = wg3_ujobs(md.Do, ko);
        if (rb < nj) break;
        rb -= nj;
    }
    if (ko < wa.obs_ko) {
        const int NU = min(4, (md.Do - 64 * ko + 15) >> 4);
        if (nj == 1) wg3_job_U<H, 4>(k, 0, ko, NU);
        else if (nj == 2) wg3_job_U<H, 2>(k, 2 * rb, ko, NU);
        else wg3_job_U<H, 1>(k, rb, ko, NU);
        return;
    }
    wg3_job_V<H>(k, rb >> 1, rb & 1, md.Da);
}
