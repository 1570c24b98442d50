// kernels_fbco.hpp -- CO-RESIDENT forms of the two full-batch tile kernels (round 5).
//
// fb_hvp_mixed_kernel / fb_tile_mixed_kernel (kernels_fb.hpp) run ONE 1024-thread workgroup per CU (158 KB of LDS): the weight
// ingest, the layer-1 product, the head and the spills of a tile serialise with its GEMMs behind __syncthreads, and there is no
// second tile on the CU whose MFMAs could run meanwhile (r4: 46 % / 42 % MFMA-busy with every CU occupied).  The kernels here
// process the SAME 32-row (and trailing 16-row) tiles with 2 H = 512 threads and <= 80 KB of LDS, so that TWO workgroups are
// resident per CU (16 waves, <= 128 VGPRs each): one tile's non-GEMM phases run under the other tile's MFMAs.
//
// How a 32-row tile fits in half the LDS: two H-wide activation slots instead of four, time-multiplexed --
//   HVP (theta-only half cached):  s0 = h1 -> h2 -> dz2 ;  s1 = R{h1} -> R{h2} -> R{dz2}
//   tile forward / backward:       s0 = h1 ;  s1 = x^T | W1 -> h2 -> dz2
// relu' masks that the four-slot kernels read from a slot that is gone by then come from the cached activations in HBM / MALL
// (the lane's own 8 elements).  A wave owns TWO 16-column groups (g, g + H/32) and runs them one after the other through the
// same 64 fragment registers, so the L2 -> register weight traffic per row is that of the 1024-thread kernels.
// Every output element sees the arithmetic of the four-slot kernels in the same order (same MFMA chains, same head code):
// the plans agree bit for bit (tests/test_gpu_fullsize.py::test_full_batch_kernel_plans_are_bit_identical).
//
// Reference: fsrl/policy/cpo.py:177-182 (_MVP), :147-162, :234-254; fsrl/policy/trpo_lag.py:148-171, :253-259.
#pragma once
#include "kernels_fb.hpp"

// Persistent scheduling of the co-resident kernels: the grid is 2 x (number of CUs) workgroups that stay resident and DRAW their
// tiles from a device counter (32-row tiles first, 16-row tiles behind them), so that every CU keeps a pair of tiles in flight
// until the batch runs out.  With one workgroup per tile (static grid) 625 tile units on 256 CUs end in a round that fills
// 44 % of the chip -- and a workgroup that is alone on its CU has nobody to overlap with: r5's PMC passes show the pair at
// ~75 % MFMA-busy while both are resident and the launch as a whole at 48 %.  The counter is never reset: launches on the
// stream are ordered, launch s starts at base_s = sum over earlier launches of (tiles + workgroups) -- every workgroup draws
// once more than it processes -- and unsigned wrap-around is harmless.  Which workgroup computes a tile does not change a bit of it.
struct CoSched {
    unsigned* counter;      // nullptr: static grid, blockIdx.x is the tile
    unsigned base;
    int total;
    int first_round;        // workgroups [0, first_round) are the ones resident when the launch starts
    int delay;              // s_sleep(127) periods (8 128 cycles each) the SECOND resident of a CU waits before its first tile
};
// Two workgroups that start on a CU in the same cycle run in LOCKSTEP: same phases at the same time, both waiting for the matrix
// pipe or both away from it -- and lockstep is stable, because the successor of each starts when its predecessor ends.  So the
// second resident of every CU (its waves sit in wave slots >= 2 of their SIMDs: HW_ID.WAVE_ID) starts half a tile late, once per
// launch; from then on the pair stays out of phase.  Only wave 0 sleeps: the others wait for it at the first barrier.
__device__ __forceinline__ void co_desync(const CoSched& cs) {
    if (cs.delay > 0 && (int)blockIdx.x < cs.first_round && threadIdx.x < 64) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID, all 32 bits
        if ((hw & 0xFu) >= 2u)
            for (int i = 0; i < cs.delay; ++i) __builtin_amdgcn_s_sleep(127);
    }
}
// first tile of this workgroup (static grid: its block index), or -1 / >= total when there is none
__device__ __forceinline__ int co_first_tile(const CoSched& cs, int* s_next) {
    if (threadIdx.x == 0) *s_next = (int)(atomicAdd(cs.counter, 1u) - cs.base);
    __syncthreads();
    return *s_next;
}

// A/B build only (-DFSRL_CO_SETPRIO): waves raise their issue priority while they run a GEMM phase
#ifdef FSRL_CO_SETPRIO
#define CO_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define CO_PRIO(n) do { } while (0)
#endif

template <int H>
struct HvpCoSmem {
    static constexpr int LD = H + 4;
    float s0[32 * LD], s1[32 * LD];
    float xd[64 * 32];                                        // obs tile transposed [k][i] (obs_dim <= 64); after the layer-1
                                                              // tangent: dout [32][FSRL_DOW] | rdout [32][FSRL_DOW]
    float out[32 * FSRL_MAX_ACT], rout[32 * FSRL_MAX_ACT];
};
static_assert(2 * 32 * FSRL_DOW <= 64 * 32, "dout | rdout alias the observation tile");
static_assert(sizeof(HvpCoSmem<256>) <= 80 * 1024, "two workgroups per CU");

// Compiler note (hipcc 7.2, -O3): the loops over the wave's two column groups are RUNTIME loops (`#pragma unroll 1`).  Unrolled,
// the machine scheduler hoists the second group's fragment loads over the first group's MFMAs and runs the two row halves one after
// the other, keeping 2-4 fragment sets alive through scratch (400-1000 spilled dwords per lane, measured); as runtime loops the
// kernel needs 116 VGPRs and no scratch, and the fragment loads travel a chunk or two ahead of the MFMAs that consume them.
// One tile of 16 NH rows starting at row0 of the CACHED product (h1 / h2 / dout / dz2 of this theta are in A1 / A2 / DO / D2).
// FIRST (r6): the first product of a solve -- nothing is cached yet: z1 rides the layer-1 tangent's MFMA pass (the W1 burst beside the
// V1 burst), z2 = W2 h1 the layer-2 pass (same fragments, one more accumulator set), and h1 / h2 / dout / dz2 leave for A1 / A2 /
// DO / D2 on the way.  fb_hvp_mixed_kernel<.., false>'s arithmetic, element by element (bit-identical plans); it took 262-357 us per
// launch alone on its CU, four times per update.
template <int H, int NH, bool GN, bool FIRST = false>
__device__ __forceinline__ void hvp_co_body(HvpCoSmem<H>& sm, const float* __restrict__ P, const ModelDesc& md,
                                            const HvpArgs& a, const int row0) {
    constexpr int LD = HvpCoSmem<H>::LD;
    constexpr int NT = 2 * H;
    constexpr int WAVES = H / 32;
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
    float* h1 = sm.s0; float* rh1 = sm.s1;
    float* h2 = sm.s0; float* rh2 = sm.s1;
    float* d2 = sm.s0; float* rd2 = sm.s1;
    float* dout = sm.xd; float* rdout = sm.xd + 32 * FSRL_DOW;

    // ---- h1 of this theta -> slot 0 ; observations (transposed) ; the head threads' mean_old / std_old (registers)
    if constexpr (!FIRST)
        for (int e = tid; e < R * H4; e += NT) {
            const int i = e / H4, c4 = e - i * H4;
            *reinterpret_cast<f32x4*>(&h1[i * LD + 4 * c4]) = *reinterpret_cast<const f32x4*>(a.A1 + base + (size_t)i * H + 4 * c4);
        }
    for (int e = tid; e < R * Do; e += NT) {
        const int i = e / Do, k = e - i * Do;
        sm.xd[k * 32 + i] = (i < n_valid) ? a.obs[(size_t)row0 * Do + e] : 0.0f;
    }
    float mo_mean = 0.0f, mo_std = 0.0f;
    if (tid < 16 * R) {
        const int i = tid >> 4, d = tid & 15;
        if (i < n_valid) {
            mo_mean = a.rd[(size_t)(row0 + i) * FSRL_RD + FSRL_RD_MEAN + d];
            mo_std = a.rd[(size_t)(row0 + i) * FSRL_RD + FSRL_RD_STD + d];
        }
    }
    __syncthreads();

    // ---- layer-1 tangent on MFMA, the wave's two column groups:  R{h1} = relu'(z1) (V1 x + vb1)
#pragma unroll 1
    for (int g = 0; g < 2; ++g) {
        const int cg = wave + g * WAVES;
        f32x4 racc[NH], zacc[NH];
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) racc[hf] = zacc[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* __restrict__ vrow = V + no.W1 + (size_t)(cg * 16 + li) * Do;
        const float* __restrict__ wrow = P + no.W1 + (size_t)(cg * 16 + li) * Do;
        for (int k0 = 0; k0 < Do; k0 += 64) {
            float vb_[16], wb_[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = k0 + 4 * s + q;
                if constexpr (FIRST) wb_[s] = (k < Do) ? wrow[k] : 0.0f;
                vb_[s] = (k < Do) ? vrow[k] : 0.0f;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = k0 + 4 * s + q;
                if (k0 + 4 * s < Do) {
#pragma unroll
                    for (int hf = 0; hf < NH; ++hf) {
                        const float a_ = (k < Do) ? sm.xd[k * 32 + 16 * hf + li] : 0.0f;
                        if constexpr (FIRST) zacc[hf] = mfma_16x16x4(a_, wb_[s], zacc[hf]);
                        racc[hf] = mfma_16x16x4(a_, vb_[s], racc[hf]);
                    }
                }
            }
        }
        const int j = cg * 16 + li;
        const float vb1 = V[no.b1 + j];
        const float b1 = FIRST ? P[no.b1 + j] : 0.0f;
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int l = (16 * hf + 4 * q + r) * LD + j;
                bool on;
                if constexpr (FIRST) {
                    const float z = zacc[hf][r] + b1;
                    on = z > 0.0f;
                    h1[l] = on ? z : 0.0f;
                } else on = h1[l] > 0.0f;
                rh1[l] = on ? racc[hf][r] + vb1 : 0.0f;
            }
        }
    }
    __syncthreads();                                  // R{h1} complete; the observation tile is dead
    for (int e = tid; e < R * FSRL_DOW; e += NT) { dout[e] = 0.0f; rdout[e] = 0.0f; }
    // R{h1} leaves for the weight-side kernel (dz2^T R{h1}: not in the Gauss-Newton form, where dz2 = 0)
    if constexpr (!GN)
        for (int e = tid; e < R * H4; e += NT) {
            const int i = e / H4, c4 = e - i * H4;
            store4_fb(a.RA1 + base + (size_t)i * H + 4 * c4, *reinterpret_cast<const f32x4*>(&rh1[i * LD + 4 * c4]));
        }
    if constexpr (FIRST)                               // h1 into the cache (later products load it; the relu'(z1) mask below reads it back)
        for (int e = tid; e < R * H4; e += NT) {
            const int i = e / H4, c4 = e - i * H4;
            store4_fb(a.A1 + base + (size_t)i * H + 4 * c4, *reinterpret_cast<const f32x4*>(&h1[i * LD + 4 * c4]));
        }
    // ---- layer 2 tangent:  R{z2} = W2 R{h1} + V2 h1   (per column group: W2 fragments, then V2 fragments, same registers)
    f32x4 rz[2][NH], z2[2][FIRST ? NH : 1];
    CO_PRIO(2);
#pragma unroll 1
    for (int g = 0; g < 2; ++g) {
        const int cg = wave + g * WAVES;
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) rz[g][hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        FwdW2Frag<H> wf;
        wf.load_buf(P + no.W2f, cg, lane);
        if constexpr (FIRST) {
#pragma unroll
            for (int hf = 0; hf < NH; ++hf) z2[g][hf] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma_rows_n<H, NH>(h1, wf, li, q, z2[g]);
        }
        mma_rows_n<H, NH>(rh1, wf, li, q, rz[g]);
        __builtin_amdgcn_sched_barrier(0);            // one fragment set (64 VGPRs) in flight at a time: the partner workgroup fills the gap
        wf.load_buf(V + no.W2f, cg, lane);
        mma_rows_n<H, NH>(h1, wf, li, q, rz[g]);
        __builtin_amdgcn_sched_barrier(0);            // the next group's fragment loads stay behind this group's MFMAs (64 VGPRs each)
    }
    CO_PRIO(0);
    __syncthreads();                                  // both slots have been read by every wave; the R{h1} spill is out
    // ---- h2 of this theta -> slot 0
    if constexpr (!FIRST) {
        for (int e = tid; e < R * H4; e += NT) {
            const int i = e / H4, c4 = e - i * H4;
            *reinterpret_cast<f32x4*>(&h2[i * LD + 4 * c4]) = *reinterpret_cast<const f32x4*>(a.A2 + base + (size_t)i * H + 4 * c4);
        }
        __syncthreads();
    }
#pragma unroll 1
    for (int g = 0; g < 2; ++g) {
        const int j = (wave + g * WAVES) * 16 + li;
        const float vbias = V[no.b2 + j];
        const float bias = FIRST ? P[no.b2 + j] : 0.0f;
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int l = (16 * hf + 4 * q + r) * LD + j;
                bool on;
                if constexpr (FIRST) {
                    const float zz = z2[g][hf][r] + bias;
                    on = zz > 0.0f;
                    h2[l] = on ? zz : 0.0f;
                } else on = h2[l] > 0.0f;
                rh2[l] = on ? rz[g][hf][r] + vbias : 0.0f;
            }
        }
    }
    __syncthreads();
    // ---- head pre-activations: out = W3 h2 + b3 ; R{out} = W3 R{h2} + V3 h2 + vb3
    // r6: output-major -- the head rows of W3 / V3 are fetched ONCE per output (they sat inside the row loop: R / WAVES x Da dependent L2
    // round trips per tile) and the wave's R / WAVES rows unroll into independent LDS -> FMA -> wave_sum chains.  Every (row, output)
    // sees the same lanes and the same order of operations: bit-identical.
    for (int o = 0; o < Da; ++o) {
        float w3r[H / 64], v3r[H / 64];
#pragma unroll
        for (int t = 0; t < H / 64; ++t) {
            w3r[t] = P[no.W3 + (size_t)o * H + lane + 64 * t];
            v3r[t] = V[no.W3 + (size_t)o * H + lane + 64 * t];
        }
        const float b3o = P[no.b3 + o], vb3o = V[no.b3 + o];
#pragma unroll
        for (int i = wave; i < R; i += WAVES) {
            float s = 0.0f, rs = 0.0f;
#pragma unroll
            for (int t = 0; t < H / 64; ++t) {
                const float h = h2[i * LD + lane + 64 * t];
                s = fmaf(h, w3r[t], s);
                rs = fmaf(rh2[i * LD + lane + 64 * t], w3r[t], rs);
                rs = fmaf(h, v3r[t], rs);
            }
            s = wave_sum(s);
            rs = wave_sum(rs);
            if (lane == 0) {
                sm.out[i * FSRL_MAX_ACT + o] = s + b3o;
                sm.rout[i * FSRL_MAX_ACT + o] = rs + vb3o;
            }
        }
    }
    // R{h2} leaves (dout^T R{h2}: not in the Gauss-Newton form); slot 1 is free after the barrier
    if constexpr (!GN)
        for (int e = tid; e < R * H4; e += NT) {
            const int i = e / H4, c4 = e - i * H4;
            store4_fb(a.RA2 + base + (size_t)i * H + 4 * c4, *reinterpret_cast<const f32x4*>(&rh2[i * LD + 4 * c4]));
        }
    if constexpr (FIRST)                               // h2 into the cache
        for (int e = tid; e < R * H4; e += NT) {
            const int i = e / H4, c4 = e - i * H4;
            store4_fb(a.A2 + base + (size_t)i * H + 4 * c4, *reinterpret_cast<const f32x4*>(&h2[i * LD + 4 * c4]));
        }
    __syncthreads();
    // ---- KL head (per row, per action dim): dout, R{dout}, and the sigma_param rows   (hvp_tile_body's arithmetic)
    if (tid < 16 * R) {
#pragma clang fp contract(off)       // the head's arithmetic is the four-slot kernels' to the bit in every instantiation
        const int i = tid >> 4, d = tid & 15;
        if (i < n_valid && d < Da) {
            const float x = sm.out[i * FSRL_MAX_ACT + d];
            const float t = md.unbounded ? 0.0f : tanhf(x);
            const float hs = md.unbounded ? 1.0f : a.max_action;
            const float ro = sm.rout[i * FSRL_MAX_ACT + d];
            const float sp = P[no.sigma + d], rls = V[no.sigma + d];
            const float sig = expf(sp), var = sig * sig;
            const float dt = hs * (1.0f - t * t);
            const float rmu = dt * ro;
            const float dmu_b = a.max_action * t - mo_mean;
            const float dmu = GN ? 0.0f : (md.unbounded ? x - mo_mean : dmu_b);
            const float so = mo_std, so2 = GN ? var : so * so;
            const float gmu = dmu / var;
            const float rgmu = rmu / var - 2.0f * gmu * rls;
            const float rgls = -2.0f * dmu * rmu / var + 2.0f * (so2 + dmu * dmu) / var * rls;
            const float rdt = hs * (-2.0f * t) * (1.0f - t * t) * ro;
            dout[i * FSRL_DOW + d] = invN * gmu * dt;
            rdout[i * FSRL_DOW + d] = invN * (rgmu * dt + gmu * rdt);
            dout[i * FSRL_DOW + 16 + d] = invN * (1.0f - (so2 + dmu * dmu) / var);
            rdout[i * FSRL_DOW + 16 + d] = invN * rgls;
        }
    }
    __syncthreads();
    // ---- dz2 = relu'(z2) (dout W3) -> slot 0 IN PLACE of h2 ;  R{dz2} = relu'(z2) (R{dout} W3 + dout V3) -> slot 1
    for (int t = tid; t < (R / 4) * H; t += NT) {
        const int k = t % H, rg = t / H;
        float g[4] = {0, 0, 0, 0}, rg_[4] = {0, 0, 0, 0};
        for (int o = 0; o < Da; ++o) {
            const float w = P[no.W3 + (size_t)o * H + k], v = V[no.W3 + (size_t)o * H + k];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dd = dout[(4 * rg + e) * FSRL_DOW + o];
                g[e] = fmaf(dd, w, g[e]);
                rg_[e] = fmaf(rdout[(4 * rg + e) * FSRL_DOW + o], w, rg_[e]);
                rg_[e] = fmaf(dd, v, rg_[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * rg + e;
            const bool on = h2[i * LD + k] > 0.0f;        // read before this thread overwrites the element
            d2[i * LD + k] = on ? g[e] : 0.0f;
            rd2[i * LD + k] = on ? rg_[e] : 0.0f;
        }
    }
    __syncthreads();
    if constexpr (FIRST && !GN) {                      // dz2 / dout of the KL head into the cache (exact zeros, unread, in the Gauss-Newton form)
        for (int e = tid; e < R * H4; e += NT) {
            const int i = e / H4, c4 = e - i * H4;
            store4_fb(a.D2 + base + (size_t)i * H + 4 * c4, *reinterpret_cast<const f32x4*>(&d2[i * LD + 4 * c4]));
        }
        for (int e = tid; e < R * FSRL_DOW; e += NT) a.DO[(size_t)row0 * FSRL_DOW + e] = dout[e];
    }
    // ---- R{dz1} = relu'(z1) (R{dz2} W2 + dz2 V2), the wave's two column groups; relu'(z1) off the cached h1 (the lane's own elements)
    CO_PRIO(2);
#pragma unroll 1
    for (int g = 0; g < 2; ++g) {
        const int cg = wave + g * WAVES;
        const int col = cg * 16 + li;
        float m1[NH][4];
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) m1[hf][r] = a.A1[base + (size_t)(16 * hf + 4 * q + r) * H + col];
        }
        f32x4 acc[NH];
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) acc[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        mma_cols_n<H, NH, true>(rd2, P + no.W2, cg, li, q, acc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!GN) mma_cols_n<H, NH, true>(d2, V + no.W2, cg, li, q, acc);         // dz2 V2: dz2 = 0 in the Gauss-Newton form
        float* __restrict__ RD1 = a.RD1 + base;
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * hf + 4 * q + r;
                RD1[(size_t)i * H + col] = (m1[hf][r] > 0.0f) ? acc[hf][r] : 0.0f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    CO_PRIO(0);
    // ---- the remaining operands of the weight-side products
    for (int e = tid; e < R * H4; e += NT) {
        const int i = e / H4, c4 = e - i * H4;
        store4_fb(a.RD2 + base + (size_t)i * H + 4 * c4, *reinterpret_cast<const f32x4*>(&rd2[i * LD + 4 * c4]));
    }
    for (int e = tid; e < R * FSRL_DOW; e += NT) a.RDO[(size_t)row0 * FSRL_DOW + e] = rdout[e];
}

// Mixed-height grid like fb_hvp_mixed_kernel: blocks [0, n32) take 32-row tiles, the rest 16-row tiles behind them.
// 2 H threads, <= 128 VGPRs (4 waves per SIMD) and 78.8 KB of LDS: two workgroups per CU.
// PERSIST = false: one workgroup per tile (straight-line: 116 VGPRs, no scratch); true: persistent workgroups (A/B)
template <int H, bool PERSIST, bool GN, bool FIRST = false>
__global__ __launch_bounds__(2 * H, 4) void fb_hvp_co_kernel(const float* __restrict__ P, const ModelDesc md, const HvpArgs a,
                                                            const int n32, const CoSched cs) {
    __shared__ HvpCoSmem<H> sm;
    co_desync(cs);
    static_assert(!(PERSIST && FIRST), "the first product has no persistent form");
    if constexpr (!PERSIST) {
        const int b = blockIdx.x;
        if (b < n32) hvp_co_body<H, 2, GN, FIRST>(sm, P, md, a, 32 * b);
        else hvp_co_body<H, 1, GN, FIRST>(sm, P, md, a, 32 * n32 + 16 * (b - n32));
    } else {
        __shared__ int s_next;
        int b = co_first_tile(cs, &s_next);
        while (b < cs.total) {
            unsigned nxt = 0u;
            if (threadIdx.x == 0) nxt = atomicAdd(cs.counter, 1u) - cs.base;   // in flight under this tile
            if (b < n32) hvp_co_body<H, 2, GN>(sm, P, md, a, 32 * b);
            else hvp_co_body<H, 1, GN>(sm, P, md, a, 32 * n32 + 16 * (b - n32));
            __syncthreads();                               // the slots and s_next are free
            if (threadIdx.x == 0) s_next = (int)nxt;
            __syncthreads();
            b = s_next;
        }
    }
}

// ------------------------------------------------------------------------------------------
// fb_tile_mixed_kernel's tiles (forward, loss head, activation backward of ONE network's 32 / 16 rows; modes VF, SUR, KL, EVAL
// -- the launches of the trust-region updates) in the co-resident form: slot 0 = h1; slot 1 = x^T | W1 (layer 1), then h2, then
// dz2 IN PLACE of h2.  W3, the biases and sigma_param are read from L1 / L2 where the 1024-thread kernel staged them in LDS
// (same values, same order of operations); the per-row loss inputs are kept as the 21 columns these modes read (action_dim <= 4).
#define TC_RD 24            // act[4] | logp_old | adv_n[4] | ret[4] | mean_old[4] | std_old[4] | pad
#define TC_LOGP 4
#define TC_ADV 5
#define TC_RET 9
#define TC_MEAN 13
#define TC_STD 17
template <int H>
struct TileCoSmem {
    static constexpr int LD = H + 4;
    float s0[32 * LD], s1[32 * LD];
    float rd[32 * TC_RD];
    float dout[32 * FSRL_DOW];
    float out[32 * FSRL_MAX_ACT];
    float stats[32 * FB_NSTAT];
};
static_assert(sizeof(TileCoSmem<256>) <= 80 * 1024, "two workgroups per CU");
static_assert(64 * 32 + 256 * FSRL_W1_LDS <= 32 * (256 + 4), "x^T and the staged W1 share slot 1 during layer 1");

// timing probes of the co-resident tile kernel (probe builds only: FSRL_TILE_PROBE=n ends every tile after phase n; results invalid).
// The phase number travels in FbArgs::eta, which only the FOCOPS mode reads.
#ifdef FSRL_PROBES
#define TC_PROBE(a, n) ((a).mode != FB_MODE_FOCOPS && (int)(a).eta == (n))
#else
#define TC_PROBE(a, n) false
#endif

template <int H, int NH>
__device__ __forceinline__ void tile_co_body(TileCoSmem<H>& sm, const float* __restrict__ P, const ModelDesc& md, const FbArgs& a,
                                             const int row0, const int stat_tile, const int y, const int ny) {
    constexpr int LD = TileCoSmem<H>::LD;
    constexpr int NT = 2 * H;
    constexpr int WAVES = H / 32;
    constexpr int H4 = H / 4;
    constexpr int R = 16 * NH;
    static_assert(NT >= 16 * R, "one head thread per (row, dim)");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
    const int net = a.net0 + y;
    const NetOff no = md.net[net];
    const int Do = md.Do, Da = md.Da;
    const int n_valid = max(0, min(R, a.N - row0));
    const float invN = 1.0f / (float)a.N;
    float* h1 = sm.s0; float* h2 = sm.s1; float* d2 = sm.s1;
    float* xT = sm.s1; float* w1s = sm.s1 + 64 * 32;

    // ---- inputs: observations (transposed, 32-row stride like the 1024-thread kernel), W1 when it is small, the loss inputs
    for (int e = tid; e < R * Do; e += NT) {
        const int i = e / Do, k = e - i * Do;
        xT[k * 32 + i] = (i < n_valid) ? a.obs[(size_t)row0 * Do + e] : 0.0f;
    }
    if (Do <= FSRL_W1_LDS)
        for (int e = tid; e < H * Do; e += NT) w1s[e] = P[no.W1 + e];
    for (int e = tid; e < R * TC_RD; e += NT) {
        const int i = e / TC_RD, f = e - i * TC_RD;
        const int src = f < TC_LOGP ? f : f < TC_ADV ? FSRL_RD_LOGP : f < TC_RET ? FSRL_RD_ADV + (f - TC_ADV)
                        : f < TC_MEAN ? FSRL_RD_RET + (f - TC_RET) : f < TC_STD ? FSRL_RD_MEAN + (f - TC_MEAN)
                        : FSRL_RD_STD + min(f - TC_STD, 3);
        sm.rd[e] = (a.rd != nullptr && i < n_valid && f < TC_STD + 4) ? a.rd[(size_t)(row0 + i) * FSRL_RD + src] : 0.0f;
    }
    for (int e = tid; e < R * FSRL_DOW; e += NT) sm.dout[e] = 0.0f;
    __syncthreads();
    if (TC_PROBE(a, 1)) return;

    // ---- layer 1 -> h1 (slot 0): tile_forward's two paths, the wave's two column groups on the MFMA one
    if (Do <= FSRL_W1_LDS) {
        for (int t = tid; t < (R / 4) * H; t += NT) {
            const int j = t % H, rg = t / H;
            const float b = P[no.b1 + j];
            float acc[4] = {b, b, b, b};
            const float* __restrict__ w = &w1s[j * Do];
            for (int k = 0; k < Do; ++k) {
                const float wk = w[k];
                const f32x4 x = *reinterpret_cast<const f32x4*>(&xT[k * 32 + 4 * rg]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(x[e], wk, acc[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) h1[(4 * rg + e) * LD + j] = fmaxf(acc[e], 0.0f);
        }
    } else {
#pragma unroll 1
        for (int g = 0; g < 2; ++g) {
            const int cg = wave + g * WAVES;
            f32x4 acc[NH];
#pragma unroll
            for (int hf = 0; hf < NH; ++hf) acc[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* __restrict__ wrow = P + no.W1 + (size_t)(cg * 16 + li) * Do;
            for (int k0 = 0; k0 < Do; k0 += 64) {
                float b[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int k = k0 + 4 * s + q;
                    b[s] = (k < Do) ? wrow[k] : 0.0f;
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int k = k0 + 4 * s + q;
                    if (k0 + 4 * s < Do) {
#pragma unroll
                        for (int hf = 0; hf < NH; ++hf)
                            acc[hf] = mfma_16x16x4((k < Do) ? xT[k * 32 + 16 * hf + li] : 0.0f, b[s], acc[hf]);
                    }
                }
            }
            const int j = cg * 16 + li;
            const float bias = P[no.b1 + j];
#pragma unroll
            for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[(16 * hf + 4 * q + r) * LD + j] = fmaxf(acc[hf][r] + bias, 0.0f);
            }
        }
    }
    __syncthreads();                                  // h1 complete; x^T / W1 in slot 1 are dead
    if (TC_PROBE(a, 2)) return;

    // ---- layer 2 -> h2 (slot 1)
    CO_PRIO(2);
#pragma unroll 1
    for (int g = 0; g < 2; ++g) {
        const int cg = wave + g * WAVES;
        FwdW2Frag<H> wf;
        wf.load_buf(P + no.W2f, cg, lane);
        f32x4 acc[NH];
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) acc[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        mma_rows_n<H, NH>(h1, wf, li, q, acc);
        const int j = cg * 16 + li;
        const float bias = P[no.b2 + j];
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[(16 * hf + 4 * q + r) * LD + j] = fmaxf(acc[hf][r] + bias, 0.0f);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    CO_PRIO(0);
    __syncthreads();
    if (TC_PROBE(a, 3)) return;

    // ---- head pre-activations: one wave per (row, output)
    // r6: output-major, the head row of W3 fetched once per output, the wave's R / WAVES rows as independent chains (see fb_hvp_co_kernel)
    for (int o = 0; o < no.out; ++o) {
        float w3r[H / 64];
#pragma unroll
        for (int t = 0; t < H / 64; ++t) w3r[t] = P[no.W3 + (size_t)o * H + lane + 64 * t];
        const float b3o = P[no.b3 + o];
#pragma unroll
        for (int i = wave; i < R; i += WAVES) {
            float s = 0.0f;
#pragma unroll
            for (int t = 0; t < H / 64; ++t) s = fmaf(h2[i * LD + lane + 64 * t], w3r[t], s);
            s = wave_sum(s);
            if (lane == 0) sm.out[i * FSRL_MAX_ACT + o] = s + b3o;
        }
    }
    __syncthreads();

    const bool backward = (a.mode != FB_MODE_EVAL);
    // ---- loss head: thread (row i = tid >> 4, dim d = tid & 15)       (fb_tile_body's arithmetic, actor and V-critic cases)
    if (tid < 16 * R) {
        const int i = tid >> 4, d = tid & 15;
        const bool valid = i < n_valid;
        const float* rd = &sm.rd[i * TC_RD];
        float st[FB_NSTAT];
#pragma unroll
        for (int k = 0; k < FB_NSTAT; ++k) st[k] = 0.0f;
        if (net == 0) {
            float dmu_ = 0.0f, dsg_ = 0.0f;
            fb_tr_actor_head(d < Da ? sm.out[i * FSRL_MAX_ACT + d] : 0.0f, d < Da ? P[no.sigma + d] : 0.0f, rd[d & 3], rd[TC_MEAN + (d & 3)],
                             rd[TC_STD + (d & 3)], rd[TC_LOGP], rd[TC_ADV], rd[TC_ADV + 1], d, Da, lane, valid, a.mode, a.cr, a.cc,
                             a.max_action, invN, md.unbounded, dmu_, dsg_, st);
            if (valid && d < Da && (a.mode == FB_MODE_SUR || a.mode == FB_MODE_KL)) {
                sm.dout[i * FSRL_DOW + d] = dmu_;
                sm.dout[i * FSRL_DOW + 16 + d] = dsg_;
            }
        } else {
            const int c = net - 1;
            const float dd = rd[TC_RET + c] - sm.out[i * FSRL_MAX_ACT];
            if (valid) {
                if (d == 0) sm.dout[i * FSRL_DOW] = -2.0f * dd * invN;
                st[0] = dd * dd;
            }
        }
        if (d == 0) {
#pragma unroll
            for (int k = 0; k < FB_NSTAT; ++k) sm.stats[i * FB_NSTAT + k] = st[k];
        }
    }
    __syncthreads();
    // the partial sums of the 16-row halves in the slots the 16-row tiles would write (fb_reduce_stats_kernel adds them in order)
    if (tid < NH * FB_NSTAT) {
        const int half = tid >> 3, f = tid & 7;
        float t = 0.0f;
        for (int i = 0; i < 16; ++i) t += sm.stats[(16 * half + i) * FB_NSTAT + f];
        a.statp[((size_t)(stat_tile + half) * ny + y) * FB_NSTAT + f] = t;
    }
    if (!backward) return;
    if (TC_PROBE(a, 4)) return;

    // ---- activation backward (tile_backward): spills for the weight-gradient kernel
    const size_t nb = ((size_t)y * a.rows_pad + row0);
    float* __restrict__ A1 = a.A1 + nb * H; float* __restrict__ A2 = a.A2 + nb * H;
    float* __restrict__ D1 = a.D1 + nb * H; float* __restrict__ D2 = a.D2 + nb * H;
    float* __restrict__ DOb = a.DO + nb * FSRL_DOW;
    for (int e = tid; e < R * H4; e += NT) {
        const int i = e / H4, c4 = e - i * H4;
        store4_fb(&A1[(size_t)i * H + 4 * c4], *reinterpret_cast<const f32x4*>(&h1[i * LD + 4 * c4]));
        store4_fb(&A2[(size_t)i * H + 4 * c4], *reinterpret_cast<const f32x4*>(&h2[i * LD + 4 * c4]));
    }
    __syncthreads();                                  // h2 has left: dz2 may take its place
    for (int t = tid; t < (R / 4) * H; t += NT) {     // dz2 = (dout @ W3) * relu'(z2)
        const int k = t % H, rg = t / H;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        for (int o = 0; o < no.out; ++o) {
            const float w = P[no.W3 + (size_t)o * H + k];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = fmaf(sm.dout[(4 * rg + e) * FSRL_DOW + o], w, g[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * rg + e;
            d2[i * LD + k] = (h2[i * LD + k] > 0.0f) ? g[e] : 0.0f;      // in place: this thread alone touches the element
        }
    }
    __syncthreads();
    for (int e = tid; e < R * H4; e += NT) {
        const int i = e / H4, c4 = e - i * H4;
        store4_fb(&D2[(size_t)i * H + 4 * c4], *reinterpret_cast<const f32x4*>(&d2[i * LD + 4 * c4]));
    }
    for (int e = tid; e < R * FSRL_DOW; e += NT) DOb[e] = sm.dout[e];
    if (TC_PROBE(a, 5)) return;
    // ---- dz1 = (dz2 @ W2) * relu'(z1), the wave's two column groups
    CO_PRIO(2);
#pragma unroll 1
    for (int g = 0; g < 2; ++g) {
        const int cg = wave + g * WAVES;
        const int col = cg * 16 + li;
        f32x4 acc[NH];
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) acc[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
        mma_cols_n<H, NH, true>(d2, P + no.W2, cg, li, q, acc);
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * hf + 4 * q + r;
                D1[(size_t)i * H + col] = (h1[i * LD + col] > 0.0f) ? acc[hf][r] : 0.0f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The grid of fb_tile_mixed_kernel: ny * n32 32-row tiles first, ny * n16 16-row tiles behind them.
template <int H, bool PERSIST>
__global__ __launch_bounds__(2 * H, 4) void fb_tile_co_kernel(const float* __restrict__ P, const ModelDesc md, const FbArgs a,
                                                             const int n32, const int n16, const int ny, const CoSched cs) {
    __shared__ TileCoSmem<H> sm;
    co_desync(cs);
    auto one = [&](int b) {
        if (b < ny * n32) {
            const int y = b / n32, t = b - y * n32;
            tile_co_body<H, 2>(sm, P, md, a, 32 * t, 2 * t, y, ny);
        } else {
            const int b2 = b - ny * n32;
            const int y = b2 / n16, t = b2 - y * n16;
            tile_co_body<H, 1>(sm, P, md, a, 32 * n32 + 16 * t, 2 * n32 + t, y, ny);
        }
    };
    if constexpr (!PERSIST) {
        one((int)blockIdx.x);
    } else {
        __shared__ int s_next;
        int b = co_first_tile(cs, &s_next);
        while (b < cs.total) {
            unsigned nxt = 0u;
            if (threadIdx.x == 0) nxt = atomicAdd(cs.counter, 1u) - cs.base;
            one(b);
            __syncthreads();
            if (threadIdx.x == 0) s_next = (int)nxt;
            __syncthreads();
            b = s_next;
        }
    }
}
