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
//     4 x ceil(Do / 16) MFMAs; db1 rides with the dW1 jobs, db2 with the dW2 jobs of tile column 0; dW3 (two jobs of 128
//     hidden units, 8 MFMAs per k-step), db3 and dsigma ride together.  The second operand pair of an R-op launch is 16 + 2 MORE jobs with
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
#define WG3_EXT 1024                  // floats behind the four slots: bias partials [8 waves][64] | dout column sums [8][64]

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

// NP = 1 + (the launch has a second operand pair).  Jobs of one (network, split), all 8-way split-K over the waves with the
// operands of the next two k-steps in flight:
//   T(p, tj, tk)  16 x NP   dW2 tile of pair p: out[j][k] = sum_r Y_p[r][64 tj + j] X_p[r][64 tk + k]       16 MFMAs per k-step
//   U(tj, ko)      4 x KO   dW1 tile: Y1[:, 64 tj ..] x obs_pad[:, group ko]; ko = 0 also db1               4 x ceil(Do / 16)
//   V(p, half)     2 x NP   dW3 of pair p for 128 hidden units: h2_p[:, j] x dout_p[:, o]; the first also db3 / dsigma    8
// A second pair gets partial slots of its own (z + nsplit): a T job is the same work whichever pair it belongs to, so every job
// of a launch is about one unit and the consumers simply add NP x nsplit partials.  What pair b does not produce (W1, b1, b2,
// b3, sigma) is written as zeros into its slots by the jobs that own those outputs.
template <int H>
__global__ __launch_bounds__(512, 4) void fb_wgrad3_kernel(const ModelDesc md, const FbWgradArgs wa, const int NP, const int nsplit) {
    static_assert(H == 256, "fb_wgrad3_kernel is written for 256-wide layers (other widths keep fb_wgrad_kernel)");
    constexpr int TPD = H / 64, NT2 = TPD * TPD, SLOT = WG3_SLOT, NW = 8;
    __shared__ float red[4 * SLOT + WG3_EXT];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KO = wa.obs_ko;
    const int NTJ = NT2 * NP, NUJ = TPD * KO;
    const int NB = NTJ + NUJ + 2 * NP;
    int Lp = blockIdx.x;
    if (wa.wg3_flags & 1) {                                  // XCD-aware order (see the header)
        const int per = gridDim.x >> 3;
        Lp = (Lp & 7) * per + (Lp >> 3);
    }
    if (Lp >= wa.remap_total) return;
#ifdef FSRL_PROBES
    if (wa.dbg_skip & 32) return;                            // timing probe: launch + dispatch only
#endif
    const int rb = Lp % NB, grp = Lp / NB;
    const int by = grp % wa.remap_ny, bz = grp / wa.remap_ny;
    const FbWgradNet wn = wa.nets[by];
    const NetOff no = md.net[wn.net];
    const int KS0 = bz * wa.ks_per_split;
    const int KS = min(wa.rows >> 2, KS0 + wa.ks_per_split);     // this split's k-step range [KS0, KS): 4 rows per k-step
    float* __restrict__ gout = wa.out + (size_t)bz * wa.split_stride;
    float* __restrict__ gout_b = NP == 2 ? wa.out + (size_t)(bz + nsplit) * wa.split_stride : nullptr;   // pair b's slot
    float* ext = red + 4 * SLOT;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int Do = md.Do, out = no.out;

    if (rb < NTJ + NUJ) {
        // ---- a 64 x 64 tile job: lane (c, q) holds four columns of row 4 s + q of each operand;
        //      acc[t][u][r] = output (j = 4 (4 q + r) + t, k = 4 c + u) of the tile (dW1: k = 64 ko + 16 u + c, obs_pad's order)
        const bool isT = rb < NTJ;
        int pr = 0, tj, tk, NU = 4;
        const float* py; const float* px; int ldx = H;
        if (isT) {
            pr = rb / NT2; tj = (rb % NT2) / TPD; tk = rb % TPD;
            py = (pr ? wn.w2_yb : wn.w2_ya) + tj * 64 + 4 * c;
            px = (pr ? wn.w2_xb : wn.w2_xa) + tk * 64 + 4 * c;
        } else {
            const int ui = rb - NTJ;
            tj = ui % TPD; tk = ui / TPD;                    // tk = the observations' 64-column group
            py = wn.w1_y + tj * 64 + 4 * c;
            px = wa.obs_pad + (size_t)64 * tk + 4 * c;
            ldx = 64 * KO;
            NU = min(4, (Do - 64 * tk + 15) >> 4);           // 16-column chunks of this group that exist
        }
        f32x4 acc[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[t][u] = zero4;
        f32x4 sy = zero4;                                    // column sums of Y: db2 (T jobs of pair a, tile column 0: b2_src == w2_ya)
                                                             // or db1 (U jobs of group 0: b1_src == w1_y)
        struct Set { f32x4 y, x; };
        auto fetch = [&](const int s, Set& o) {
            const unsigned r = (unsigned)(4 * s + q);        // 32-bit offsets: rows * 256 floats stays far below 2^32
            o.y = *reinterpret_cast<const f32x4*>(py + r * (unsigned)H);
            o.x = *reinterpret_cast<const f32x4*>(px + r * (unsigned)ldx);
        };
        auto fma = [&](const Set& o) {
#ifdef FSRL_PROBES
            if (wa.dbg_skip & 16) { sy += o.y + o.x; return; }   // timing probe: no matrix work (results invalid)
#endif
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u < NU) {                                // block-uniform (dW1 jobs of a narrow observation group)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t][u] = mfma_16x16x4(o.y[t], o.x[u], acc[t][u]);
                }
            }
            sy += o.y;
            // pin the sum HERE: left alone, the scheduler sinks these adds to the end of the loop body, which keeps every set's y alive
            // past its re-fetch -- the re-fetched sets land in other registers and are copied back behind `s_waitcnt vmcnt(1)`
            // once per three k-steps (the whole prefetch drained)
            asm volatile("" : "+v"(sy));
        };
        {
            Set A, B, C;
            A.y = A.x = B.y = B.x = C.y = C.x = zero4;
            wg3_pipeline(KS0, KS, wave, A, B, C, fetch, fma, wa.dbg_skip);
        }
        const bool bias = tk == 0;                           // T: db2 | U: db1
        if (bias) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                sy[t] += __shfl_xor(sy[t], 16, 64);
                sy[t] += __shfl_xor(sy[t], 32, 64);
            }
            if (q == 0) *reinterpret_cast<f32x4*>(ext + wave * 64 + 4 * c) = sy;
        }
        if (isT) wg3_tile_to_lds<false>(acc, red, wave, c, q);
        else wg3_tile_to_lds<true>(acc, red, wave, c, q);
        float* __restrict__ go = pr ? gout_b : gout;
#pragma unroll
        for (int e0 = 0; e0 < 4096; e0 += 512) {
            const int e = e0 + tid, jl = e >> 6, kl = e & 63;
            const float v = (red[jl * 65 + kl] + red[SLOT + jl * 65 + kl]) + (red[2 * SLOT + jl * 65 + kl] + red[3 * SLOT + jl * 65 + kl]);
            if (isT) go[no.W2 + (size_t)(tj * 64 + jl) * H + tk * 64 + kl] = v;
            else if (64 * tk + kl < Do) {
                const size_t o1 = no.W1 + (size_t)(tj * 64 + jl) * Do + 64 * tk + kl;
                gout[o1] = v;
                if (NP == 2) gout_b[o1] = 0.0f;
            }
        }
        if (bias && tid < 64) {
            float b = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) b += ext[w * 64 + tid];
            const int ob = (isT ? no.b2 : no.b1) + tj * 64 + tid;
            if (isT) go[ob] = pr ? 0.0f : b;                 // pair b has no bias term
            else { gout[ob] = b; if (NP == 2) gout_b[ob] = 0.0f; }
        }
        return;
    }

    // ---- V(p, half): dW3[o][j] = sum_r dout_p[r][o] h2_p[r][j] for 128 hidden units j (half = 0 / 1); (pair a, half 0) also carries
    //      the column sums of its dout-like rows: db3 (columns 0 .. 15) and dsigma (16 .. 31); do_src == w3_ya in every caller.
    //      ad[g][t][r] = output (j = 128 half + 64 g + 4 (4 q + r) + t, o = c)
    {
        const int vi = rb - NTJ - NUJ, pr = vi >> 1, half = vi & 1;
        const bool head = vi == 0;
        f32x4 ad[2][4];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) ad[g][t] = zero4;
        float s3a = 0.0f, s3b = 0.0f;
        const float* px3 = (pr ? wn.w3_xb : wn.w3_xa) + 128 * half + 4 * c;
        const float* pd3 = (pr ? wn.w3_yb : wn.w3_ya) + c;
        struct VSet { f32x4 x[2]; float d, d2; };
        auto vfetch = [&](const int s, VSet& o) {
            const unsigned r = (unsigned)(4 * s + q);
#pragma unroll
            for (int g = 0; g < 2; ++g) o.x[g] = *reinterpret_cast<const f32x4*>(px3 + r * (unsigned)H + 64 * g);
            o.d = pd3[r * (unsigned)FSRL_DOW];
            o.d2 = pd3[r * (unsigned)FSRL_DOW + 16];
        };
        auto vfma = [&](const VSet& o) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int t = 0; t < 4; ++t) ad[g][t] = mfma_16x16x4(o.x[g][t], o.d, ad[g][t]);
            s3a += o.d;
            s3b += o.d2;
            asm volatile("" : "+v"(s3a), "+v"(s3b));         // as above: the sums happen here, the set is dead behind its MFMAs
        };
        {
            VSet A, B, C;
#pragma unroll
            for (int g = 0; g < 2; ++g) A.x[g] = B.x[g] = C.x[g] = zero4;
            A.d = A.d2 = B.d = B.d2 = C.d = C.d2 = 0.0f;
            wg3_pipeline(KS0, KS, wave, A, B, C, vfetch, vfma, wa.dbg_skip);
        }
        if (head) {
            s3a += __shfl_xor(s3a, 16, 64); s3a += __shfl_xor(s3a, 32, 64);
            s3b += __shfl_xor(s3b, 16, 64); s3b += __shfl_xor(s3b, 32, 64);
            if (q == 0) { ext[wave * 64 + c] = s3a; ext[wave * 64 + 16 + c] = s3b; }
        }
        // [128 hidden units][16 outputs] per slot (17 floats a row)
        float* myred = red + (wave & 3) * SLOT;
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
        float* __restrict__ go = pr ? gout_b : gout;
#pragma unroll
        for (int e0 = 0; e0 < 128 * 16; e0 += 512) {
            const int e = e0 + tid, j = e >> 4, o = e & 15;
            const float v = (red[j * 17 + o] + red[SLOT + j * 17 + o]) + (red[2 * SLOT + j * 17 + o] + red[3 * SLOT + j * 17 + o]);
            if (o < out) go[no.W3 + (size_t)o * H + 128 * half + j] = v;
        }
        if (half == 0 && tid < 32) {
            float tot = 0.0f;
            if (head) {
#pragma unroll
                for (int w = 0; w < NW; ++w) tot += ext[w * 64 + tid];
            }
            if (tid < out) go[no.b3 + tid] = tot;
            if (no.sigma >= 0 && tid >= 16 && tid < 16 + md.Da) go[no.sigma + tid - 16] = tot;
        }
    }
}
