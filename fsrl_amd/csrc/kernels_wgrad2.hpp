// kernels_wgrad2.hpp -- the weight-side products of the full-batch passes, second form: one streaming pass per workgroup.
//
// Reference: the gradients autograd forms for cpo.py:147-162 (critics), :206-220 (_get_flat_grad), :177-182 (_MVP) and
// trpo_lag.py:234-259 -- dW2 = dz2^T h1, dW1 = dz1^T x, dW3 = dout^T h2, the bias sums, and their R-op twins.
//
// fb_wgrad_kernel (kernels_fb.hpp) gives every 64 x 64 tile of dW2 its own workgroup per row split, so each of the 20 MB
// operand arrays a full-batch pass spills is read once per tile column -- FOUR times -- straight into registers: 413-430 MB
// of memory-side traffic per launch at N = 20 000 (profiles/r03_pmc_traffic_updates.json), 5.4 TB/s for 70-80 us, the
// matrix pipe 11-40 % busy.  Here a workgroup owns a QUARTER of the output columns of one network (64 hidden units j: the
// rows 64 jq .. 64 jq + 63 of dW2 and dW1, the same columns of dW3, db1, db2) for one slice of the batch rows and walks
// that slice ONCE, KB rows per stage:
//   * the stage's operands -- Y[:, quarter] (64 floats a row), X (all 256), dz1[:, quarter], h2[:, quarter], dout, the
//     observations -- go from memory INTO LDS by LDS-DMA (`global_load_lds_dwordx4` for whole X rows, `_dword` for the
//     64-float rows, which keeps their padded row strides): no staging registers, no ds_write pass, and the copy of stage
//     s + 2 runs underneath the MFMAs of stages s and s + 1 (three LDS buffers, ONE barrier per stage: wait for the own
//     loads of stage s -> barrier -> issue stage s + 2 into the buffer stage s - 1 just left -> compute stage s).  The loads
//     are inline asm (hipcc would fence every barrier with vmcnt(0) for a compiler-visible LDS-DMA) and counted by hand:
//     every wave issues the same number of loads per stage, so `s_waitcnt vmcnt(NI)` is the stage boundary;
//   * 16 waves as a 2 x 8 grid each hold a 32 x 32 piece of the 64 x 256 dW2 quarter in 16 accumulator registers (four
//     MFMAs per 4-row k-step from two ds_read_b64); the small products ride along in the same stage (dW1: one or two more
//     MFMAs per wave and k-step, dW3: one in a quarter of the waves, the bias sums on the VALU).  No split-K across the
//     waves: every partial leaves the registers directly.
// The four quarter workgroups of a slice are placed on ONE XCD (hardware block L runs on XCD L % 8: observed placement, used
// for speed only), so the X rows they share come out of that XCD's L2 three times out of four.  Every workgroup writes one
// partial gradient of its quarter at out + z * split_stride; the consumers add the partials in z order as before (float64,
// fixed order: deterministic).
//
// NOT the default (fsrl_tr_set_plan(wgrad = 3) selects it).  Measured, CPO at N = 20 000, same box, per launch R-op / plain
// (the split-K kernel it would replace: 82 / 78 us, 413-430 MB of memory-side traffic):
//   two sequential phases (dW2, then the small products), operands staged through registers one stage ahead   81 / 92 us
//   the same two stages ahead                                                                                  74 / 87 us
//   ONE loop over both, two stages ahead in registers                                                          80 / 86 us
//     (loads taken out of the loop: 71 / 78 us; MFMA work taken out: 36 / 38 us: the stage was bound by its own LDS write pass
//      and barrier skew on top of the MFMA time, not by memory -- 160 MB per launch)
//   three register sets: 100-200 registers spilled (252 us)
//   LDS-DMA with 4-byte instructions for the 64-float rows (padded strides kept)                             100 / 108 us
//   LDS-DMA, 16 bytes per lane everywhere, swizzle through the source address (this file)                     80 / 80 us
//     loads alone 33 / 30 us, MFMA work alone 64 / 66 us (dW2 alone 39 us for 34 us of MFMA time at the chip's peak; the small
//     products 13 us for 6 us; launch + prologue + epilogue 11.5 us): the sum, not the maximum -- the copy does not hide under
//     the matrix pipe here.  Same time as the split-K kernel with 2.6x fewer bytes; the 32-64 partials it leaves cost
//     the consumers 3 us more per launch than the <= 24 of the split-K kernel: CPO 36.5 vs 36.0 ms, TRPO-Lag 29.7 vs 29.0 ms.
// profiles/r04_wgrad2_ab_*.csv hold the two kernel traces.
#pragma once
#include "kernels_fb.hpp"

#define WG2_Q 4                       // output quarters per network
#ifndef WG2_PROBE
#define WG2_PROBE 0                   // timing probes (ab builds only): 1 = no loads inside the loop, 2 = no MFMA work, 3 = neither
#endif
#define WG2_NI_MAX 7                  // LDS-DMA instructions per wave and stage, at most

template <bool PAIR2>
struct Wg2Geom {
    static constexpr int NP = PAIR2 ? 2 : 1;
    static constexpr int KB = PAIR2 ? 8 : 16;           // rows per stage
    static constexpr int LDX = 288;                     // X rows: 256 + 32 (ds_read_b64 wants a row stride = 32 (mod 64) banks); one
                                                        // 16-byte LDS-DMA instruction per row, so rows may be padded
    // The 64-float rows (Y, dz1, h2 quarters) and the 32-float dout rows are written FOUR (EIGHT) rows per 16-byte LDS-DMA
    // instruction -- 1 KB contiguous in LDS, no room for padding -- and swizzled through the SOURCE address instead: odd rows
    // hold their columns XOR 32 (ds_read_b64 operands) or XOR 16 (ds_read_b32 operands), which puts the two rows a lane group
    // reads on disjoint banks.  (4-byte LDS-DMA instructions kept the padded strides but moved a quarter of the bytes per
    // instruction: the loads alone took 51 us per launch against 36 us through registers.)
    static constexpr int LDQ = 64, LDD = 32;
    static constexpr int O_D1 = NP * KB * (LDX + LDQ);  // the small products' operands sit behind the dW2 operands
    static constexpr int O_A2 = O_D1 + KB * LDQ, O_DO = O_D1 + (1 + NP) * KB * LDQ, O_OBS = O_DO + NP * KB * LDD;
    static constexpr int BUF = O_OBS + KB * FSRL_MAX_OBS;   // floats per stage buffer (obs: a flat run of KB * Do floats)
    static constexpr int NBUF = 3;
    // instructions per stage: X rows | Y, dz1, h2: 4 rows each | dout: 8 rows each (at least one) | + the observation chunks
    static constexpr int N_FIXED = NP * KB + (NP + 1 + NP) * (KB / 4) + NP * ((KB + 7) / 8);
    static_assert((N_FIXED + KB * FSRL_MAX_OBS / 64 + 15) / 16 <= WG2_NI_MAX, "more LDS-DMA instructions per wave than planned");
};

// one LDS-DMA instruction: every lane's `gsrc` (16 or 4 bytes) lands at lds_dst + lane * size; lds_dst = wave-uniform LDS byte address
__device__ __forceinline__ void glds16(const void* gsrc, const unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, const unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_n(const int n) {          // n: kernel-uniform, 0 .. WG2_NI_MAX
    switch (n) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<1>(); break;
        case 2: wait_vm<2>(); break;
        case 3: wait_vm<3>(); break;
        case 4: wait_vm<4>(); break;
        case 5: wait_vm<5>(); break;
        case 6: wait_vm<6>(); break;
        default: wait_vm<7>(); break;
    }
}

// one workgroup: network wa.nets[blockIdx.y], output quarter jq, row slice z.  wa.rows and rows_per_split are multiples of 16:
// every stage is a whole stage (rows past N hold zeros in the Y-side arrays, as fb_wgrad_kernel assumes too).
template <int H, bool PAIR2>
__global__ __launch_bounds__(1024) void fb_wgrad2_kernel(const ModelDesc md, const FbWgradArgs wa, const int nsplit,
                                                        const int rows_per_split) {
    static_assert(H == 256, "fb_wgrad2_kernel is written for 256-wide layers (other widths keep fb_wgrad_kernel)");
    using G = Wg2Geom<PAIR2>;
    constexpr int NP = G::NP, KB = G::KB, LDX = G::LDX, LDQ = G::LDQ, LDD = G::LDD;
    constexpr int O_D1 = G::O_D1, O_A2 = G::O_A2, O_DO = G::O_DO, O_OBS = G::O_OBS;
    static_assert(G::NBUF * G::BUF * 4 <= 160 * 1024, "stage buffers exceed the CU's LDS");
    __shared__ float lds[G::NBUF][G::BUF];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: hardware block L sits on XCD L % 8; the four quarters of slice z share an XCD
    const int L = blockIdx.x, xcd = L & 7, m = L >> 3;
    const int jq = m & 3, z = (m >> 2) * 8 + xcd;
    if (z >= nsplit) return;
    const FbWgradNet wn = wa.nets[blockIdx.y];
    const NetOff no = md.net[wn.net];
    const int Do = md.Do, out = no.out;
    const int r0 = z * rows_per_split, r1 = min(wa.rows, r0 + rows_per_split);
    const int nstage = (r1 - r0) / KB;                       // whole stages (see above)
    float* __restrict__ gout = wa.out + (size_t)z * wa.split_stride;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int NKT = (Do + 15) >> 4;                          // 16-column chunks of the observations (<= 8)

    // ---- this wave's LDS-DMA instructions of a stage: item e = wave + 16 i of the list X | Y | dz1 | h2 | dout | obs chunks
    const int n_obs = (KB * Do + 63) >> 6;                   // 64-float chunks of the stage's observation run
    const int n_items = G::N_FIXED + n_obs;
    const int NI = (n_items + 15) >> 4;                      // per wave; a wave short of items repeats one (same bytes, same place)
    const char* gsrc[WG2_NI_MAX];                            // per lane: source address at stage 0
    int gstep[WG2_NI_MAX];                                   // bytes per stage
    unsigned ldso[WG2_NI_MAX];                               // wave-uniform: byte offset inside a stage buffer
    int kind[WG2_NI_MAX];                                    // 0 = 16 bytes per lane, 2 = 4 bytes of the observation run (clamped)
    const char* obs_last = reinterpret_cast<const char*>(wa.obs + (size_t)wa.N * Do - 1);
    const int l16 = lane & 15, l4 = lane >> 4, l8 = lane & 7, l3 = lane >> 3;
#pragma unroll
    for (int i = 0; i < WG2_NI_MAX; ++i) {
        int e = wave + 16 * i;
        if (e >= n_items) e -= 16 * ((e - n_items) / 16 + 1);          // repeat an earlier item of this wave
        const float* g = nullptr; int step = KB * H * 4, off = 0, kd = 0;
        int f = e;
        bool done = false;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (!done && f < KB) {                           // X of pair p, row f: 1 KB = one instruction
                g = (p ? wn.w2_xb : wn.w2_xa) + (size_t)(r0 + f) * H + 4 * lane;
                off = p * KB * (LDX + LDQ) + f * LDX; done = true;
            }
            f -= KB;
            if (!done && f >= 0 && f < KB / 4) {             // Y of pair p, rows 4 f .. 4 f + 3 of this quarter; odd rows: columns XOR 32
                const int row = 4 * f + l4;
                g = (p ? wn.w2_yb : wn.w2_ya) + (size_t)(r0 + row) * H + 64 * jq + ((4 * l16) ^ ((row & 1) * 32));
                off = p * KB * (LDX + LDQ) + KB * LDX + 4 * f * LDQ; done = true;
            }
            f -= KB / 4;
        }
        if (!done && f >= 0 && f < KB / 4) {                 // dz1 quarter; odd rows: columns XOR 16
            const int row = 4 * f + l4;
            g = wn.w1_y + (size_t)(r0 + row) * H + 64 * jq + ((4 * l16) ^ ((row & 1) * 16));
            off = O_D1 + 4 * f * LDQ; done = true;
        }
        f -= KB / 4;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (!done && f >= 0 && f < KB / 4) {             // h2 quarter of pair p; odd rows: columns XOR 16
                const int row = 4 * f + l4;
                g = (p ? wn.w3_xb : wn.w3_xa) + (size_t)(r0 + row) * H + 64 * jq + ((4 * l16) ^ ((row & 1) * 16));
                off = O_A2 + p * KB * LDQ + 4 * f * LDQ; done = true;
            }
            f -= KB / 4;
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (!done && f >= 0 && f < (KB + 7) / 8) {       // dout of pair p, rows 8 f .. 8 f + 7 (32 floats each); odd rows: XOR 16
                const int row = 8 * f + l3;
                g = (p ? wn.w3_yb : wn.w3_ya) + (size_t)(r0 + row) * FSRL_DOW + ((4 * l8) ^ ((row & 1) * 16));
                off = O_DO + p * KB * LDD + 8 * f * LDD; step = KB * FSRL_DOW * 4; done = true;
            }
            f -= (KB + 7) / 8;
        }
        if (!done) {                                         // chunk f of the observation run of the stage: 64 floats, 4 bytes a lane
            g = wa.obs + (size_t)r0 * Do + 64 * f + lane;
            off = O_OBS + 64 * f; step = KB * Do * 4; kd = 2;
        }
        gsrc[i] = reinterpret_cast<const char*>(g);
        gstep[i] = step;
        ldso[i] = (unsigned)__builtin_amdgcn_readfirstlane(off * 4);
        kind[i] = __builtin_amdgcn_readfirstlane(kd);
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&lds[0][0];
    auto issue = [&](const int s, const int b) {             // stage s into buffer b
        const unsigned base = lds0 + (unsigned)b * (unsigned)(G::BUF * 4);
#pragma unroll
        for (int i = 0; i < WG2_NI_MAX; ++i) {
            if (i < NI) {
                const char* p = gsrc[i] + (size_t)s * gstep[i];
                if (kind[i] == 0) glds16(p, base + ldso[i]);
                else {
                    p = p < obs_last ? p : obs_last;         // the last slice may reach past the N observation rows
                    glds4(p, base + ldso[i]);
                }
            }
        }
    };

    const int wj = wave >> 3, wk = wave & 7;                 // dW2: 2 x 8 wave grid over the 64 x 256 quarter
    const int jt = wave & 3, wg = wave >> 2;                 // small products: hidden-unit tile of the quarter, wave group 0..3
    f32x4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = zero4;
    f32x2 s2 = {0.f, 0.f};                                   // db2 (waves with wk == 0 write it)
    f32x4 a1[2] = {zero4, zero4};                            // dW1 tiles (jt, kt = wg) and (jt, kt = wg + 4)
    f32x4 a3 = zero4;                                        // dW3 of pair wg (wave groups 0 .. NP - 1)
    float s1 = 0.0f, s3 = 0.0f;                              // db1 (wave group 0), db3 | dsigma (quarter 0: wave 15, lanes 0..31)
    const bool k0ok = 16 * wg + c < Do, k1ok = 16 * (wg + 4) + c < Do;   // the ragged last chunk reads the next row's floats: masked
    const int sw32 = (q & 1) * 32, sw16 = (q & 1) * 16;     // the row a lane reads in a k-step is 4 ks + q: its parity is q's
    auto compute = [&](const float* buf) {
        if (!(WG2_PROBE & 8))
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float* xs = buf + p * KB * (LDX + LDQ) + q * LDX + 32 * wk + 2 * c;
            const float* ys = buf + p * KB * (LDX + LDQ) + KB * LDX + q * LDQ + ((32 * wj + 2 * c) ^ sw32);
#pragma unroll
            for (int ks = 0; ks < KB / 4; ++ks) {
                const f32x2 ya = *reinterpret_cast<const f32x2*>(ys + 4 * ks * LDQ);
                const f32x2 xa = *reinterpret_cast<const f32x2*>(xs + 4 * ks * LDX);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t][u] = mfma_16x16x4(ya[t], xa[u], acc[t][u]);
                if (p == 0) s2 += ya;                        // column sums of pair a's Y (b2_src == w2_ya in every caller)
            }
        }
        if (WG2_PROBE & 4) return;
        // the small products: every branch condition is wave-uniform and sits OUTSIDE the k-step loops, so that a branch body
        // is KB / 4 LDS reads followed by KB / 4 MFMAs (one read, one wait, one MFMA at a time cost 16 us per launch)
        constexpr int NK = KB / 4;
        float y[NK];
        const float* d1 = buf + O_D1 + q * LDQ + ((16 * jt + c) ^ sw16);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) y[ks] = d1[4 * ks * LDQ];
        const float* ob = buf + O_OBS + q * Do + c;
        if (wg < NKT) {
            float o[NK];
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) o[ks] = ob[4 * ks * Do + 16 * wg];
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) a1[0] = mfma_16x16x4(y[ks], k0ok ? o[ks] : 0.0f, a1[0]);
        }
        if (wg + 4 < NKT) {
            float o[NK];
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) o[ks] = ob[4 * ks * Do + 16 * (wg + 4)];
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) a1[1] = mfma_16x16x4(y[ks], k1ok ? o[ks] : 0.0f, a1[1]);
        }
        if (wg == 0) {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) s1 += y[ks];     // b1_src == w1_y in every caller
        }
        if (wg < NP) {
            float x3[NK], dd[NK];
            const float* a2 = buf + O_A2 + wg * KB * LDQ + q * LDQ + ((16 * jt + c) ^ sw16);
            const float* dq = buf + O_DO + wg * KB * LDD + q * LDD + (c ^ sw16);
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) { x3[ks] = a2[4 * ks * LDQ]; dd[ks] = dq[4 * ks * LDD]; }
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) a3 = mfma_16x16x4(x3[ks], dd[ks], a3);
        }
        if (jq == 0 && wave == 15 && lane < 32) {
#pragma unroll
            for (int r = 0; r < KB; ++r) s3 += buf[O_DO + r * LDD + (lane ^ ((r & 1) * 16))];       // do_src == w3_ya in every caller
        }
    };

    // ---- the pipeline.  Outstanding loads of this wave when it waits in iteration s: stage s (older) and stage s + 1 (NI
    //      instructions): vmcnt(NI) = stage s has landed.  Behind the barrier every wave's part of stage s is in LDS and every
    //      wave has left stage s - 1, whose buffer stage s + 2 may now overwrite.
    issue(0, 0);
    if (nstage > 1) issue(1, 1);
    int b = 0;
    for (int s = 0; s < nstage; ++s) {
        if (!(WG2_PROBE & 1) && s + 1 < nstage) wait_vm_n(NI); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int b2 = b == 0 ? 2 : b - 1;                   // (s + 2) % 3
        if (!(WG2_PROBE & 1) && s + 2 < nstage) issue(s + 2, b2);
        if (!(WG2_PROBE & 2)) compute(lds[b]);
        b = b == 2 ? 0 : b + 1;
    }

    // ---- dW2: acc[t][u][r] of lane (c, q) = out[j = 64 jq + 32 wj + 2 (4 q + r) + t][k = 32 wk + 2 c + u]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 64 * jq + 32 * wj + 2 * (4 * q + r) + t;
            *reinterpret_cast<f32x2*>(gout + no.W2 + (size_t)j * H + 32 * wk + 2 * c) = f32x2{acc[t][0][r], acc[t][1][r]};
        }
    if (wk == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            s2[t] += __shfl_xor(s2[t], 16, 64);
            s2[t] += __shfl_xor(s2[t], 32, 64);
        }
        if (q == 0) *reinterpret_cast<f32x2*>(gout + no.b2 + 64 * jq + 32 * wj + 2 * c) = s2;
    }
    // ---- dW1: a1[i][r] of lane (c, q) = dW1[j = 64 jq + 16 jt + 4 q + r][k = 16 kt + c]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kt = wg + 4 * i;
        if (kt < NKT && 16 * kt + c < Do) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gout[no.W1 + (size_t)(64 * jq + 16 * jt + 4 * q + r) * Do + 16 * kt + c] = a1[i][r];
        }
    }
    if (wg == 0) {
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (q == 0) gout[no.b1 + 64 * jq + 16 * jt + c] = s1;
    }
    // ---- dW3: a3[r] of lane (c, q) = sum_rows h2[row][j = 64 jq + 16 jt + 4 q + r] * dout[row][o = c]; the two pairs of the
    // R-op product meet in LDS (pair a + pair b, fixed order)
    __syncthreads();                                         // every wave has left the last stage's buffer
    float* red = lds[0];
    if (PAIR2 && wg == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(jt * 16 + 4 * q + r) * 16 + c] = a3[r];
    }
    if (PAIR2) __syncthreads();
    if (wg == 0 && c < out) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = a3[r];
            if (PAIR2) v = v + red[(jt * 16 + 4 * q + r) * 16 + c];
            gout[no.W3 + (size_t)c * H + 64 * jq + 16 * jt + 4 * q + r] = v;
        }
    }
    if (jq == 0 && wave == 15 && lane < 32) {
        if (lane < out) gout[no.b3 + lane] = s3;
        if (no.sigma >= 0 && lane >= 16 && lane < 16 + md.Da) gout[no.sigma + lane - 16] = s3;
    }
}
