// kernels_mlp.hpp -- the dense part of the policy update, hand-written for gfx950.
//
// Topology (tianshou-0.5 Net + ActorProb/Critic as FSRL builds them,
// fsrl/agent/ppo_lag_agent.py:136-153):  x[Do] -> Linear(H) -> ReLU -> Linear(H) -> ReLU ->
// Linear(out);  actor: mu = max_action*tanh(out), sigma = exp(sigma_param);  critic: V = out.
//
// Work decomposition (DESIGN.md "Kernels"):
//   * one workgroup owns ONE 16-row M-tile of ONE network and runs the whole forward, the loss
//     head and the activation backward for it (ppo_fwd_bwd_kernel).  The workgroup has 4*H
//     threads = H/16 waves, ONE 16-wide N-tile per wave, so that every wave's slice of W2
//     (H/16 float4 per lane) is requested from L2 in a single burst at kernel entry: after a
//     kernel boundary the caches are cold and a dependent load round trip costs ~1 us, which
//     is what bounded the first version of this kernel -- not the MFMA pipe.
//     The two H x H GEMMs run on v_mfma_f32_16x16x4_f32 (exact fp32); the weight operand goes
//     L2 -> VGPR directly (each element is used by exactly one wave once per tile; an LDS
//     round trip would be pure overhead), the activation operand comes from LDS as b128.
//   * weight gradients are a second kernel (ppo_wgrad_kernel): 32x32 output tiles per
//     workgroup with 16-way split-K over the waves (again: all loads of a wave in one burst),
//     so no per-tile partial gradients ever reach HBM.
#pragma once
#include "common.hpp"

// rows a tile of R rows needs room for: 4-, 8- and 16-row tiles share the 16-row layout, a 32-row tile (two 16-row MFMA
// passes per weight fragment: half the L2 -> register weight traffic per row) has its own
__host__ __device__ constexpr int tile_rows(int R) { return R > 16 ? R : 16; }

template <int H, int ROWS = 16>
struct TileSmem {
    static constexpr int LD = H + 4;  // +4 floats: rows land on different banks, b128-aligned
    // observation columns held: a 32-row tile of a 256-wide network sits at 158 KB of the CU's 160 KB with 64 columns
    // (the hosts select 32-row tiles only for obs_dim <= 64)
    static constexpr int XK = (ROWS > 16) ? 64 : FSRL_MAX_OBS;
    float xT[XK * ROWS];              // obs tile transposed [k][i]
    float h1[ROWS * LD];
    float h2[ROWS * LD];
    float d2[ROWS * LD];              // dL/dz2 (after ReLU mask)
    float out[ROWS * FSRL_MAX_ACT];   // head pre-activation [i][o]
    float dout[ROWS * FSRL_DOW];      // [i][0..16) dL/dout, [i][16..32) dL/dsigma_param rows
    float rd[ROWS * FSRL_RD];         // per-row loss inputs (act, logp_old, adv_n, ret)
    float st[ROWS * 4];               // per-row partial stats
    // small parameters staged once per tile (one burst at kernel entry)
    float w3[FSRL_MAX_ACT * H];
    float b1[H], b2[H], b3[FSRL_MAX_ACT], sig[FSRL_MAX_ACT];
    float w1[H * FSRL_W1_LDS];        // W1 rows when Do <= FSRL_W1_LDS (else read from L2)
};

template <int H>
struct TileGeom {
    static constexpr int NT = 4 * H;      // threads per workgroup
    static constexpr int WAVES = H / 16;  // one 16-wide N tile per wave
};

// W2 slice of one wave for the forward GEMM: lane (li = lane&15, q = lane>>4) holds
// W2[n0+li][16*kc + 4q .. +3] for every kc.  Issued as one burst (H/16 x 16-byte loads) from the
// wave-contiguous mirror (common.hpp: w2f_index): 1 KB contiguous per instruction.
template <int H>
struct FwdW2Frag {
    f32x4 b[H / 16];
    __device__ __forceinline__ void load(const float* __restrict__ W2f, int wave, int lane) {
        const float* base = W2f + ((size_t)wave * (H / 16) * 64 + lane) * 4;
#pragma unroll
        for (int kc = 0; kc < H / 16; ++kc) b[kc] = *reinterpret_cast<const f32x4*>(base + kc * 256);
    }
    // r6, the co-resident full-batch kernels: the same burst as buffer loads (common.hpp wg3_ld4s) -- one resource for the mirror, the
    // lane's 16 bytes as the VGPR offset, the fragment (wave, kc) as an SGPR offset: no 64-bit address arithmetic per load.  Same-box
    // A/B: fb_hvp_co_kernel 138.9 -> 128.5 us, fb_tile_co_kernel 101.3 -> 97.7 us (what a backward-ORDER mirror of W2 would have
    // given, measured with a probe build, without keeping one); the fused PPO kernel LOSES with them (12.9 -> 13.85 us per launch,
    // 121.6 -> 114 updates/s) and keeps the global loads.
    __device__ __forceinline__ void load_buf(const float* __restrict__ W2f, int wave, int lane) {
        const Wg3Buf bw = wg3_buf_here(W2f);
        const unsigned frag0 = (unsigned)wave * (unsigned)(H / 16) * 256u;
#pragma unroll
        for (int kc = 0; kc < H / 16; ++kc) b[kc] = wg3_ld4s(bw, 4u * (unsigned)lane, frag0 + 256u * (unsigned)kc);
    }
};

// Prologue: the small parameters and this tile's inputs are loaded into registers first
// (issue()), the W2 burst is issued behind them, and only then are the registers written to
// LDS (commit()).  VMEM returns in issue order, so commit() waits for the small loads only and
// the 256 KB W2 burst keeps streaming in underneath layer 1.
// `x` points at the tile's first row (rows are contiguous), n_valid rows exist.
// Straight-line code: every load is unconditional on a clamped index and masked afterwards (a predicated load
// compiles to an exec-mask branch: the first version of this prologue had ~30 of them and three integer divisions by
// the runtime obs_dim, and took 2 900 shader cycles -- 1.2 us -- just to ISSUE its loads; tools/tstamp_probe.py).
template <int H, int ROWS = 16>
struct TileStage {
    static constexpr int NT = TileGeom<H>::NT;
    static constexpr int NX = (ROWS * TileSmem<H, ROWS>::XK + NT - 1) / NT;   // obs elements per thread
    static constexpr int NR = (ROWS * FSRL_RD + NT - 1) / NT;                // row-data elements per thread
    float xv[NX], w3v[4], w1v[4], rdv[NR], b1v, b2v, b3v, sgv;

    __device__ __forceinline__ void issue(const float* __restrict__ P, const NetOff no, const int Do,
                                          const int Da, const float* __restrict__ x,
                                          const float* __restrict__ rd, const int n_valid,
                                          const int tid) {
        // rows of the tile are contiguous: element e = row * Do + k is valid iff e < n_valid * Do
        const int nx = n_valid * Do;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int e = tid + u * NT;
            const float v = x[min(e, max(nx - 1, 0))];
            xv[u] = (e < nx) ? v : 0.0f;
        }
        const int n3 = no.out * H, n1 = (Do <= FSRL_W1_LDS) ? H * Do : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * NT;
            const float a = P[no.W3 + min(e, n3 - 1)];
            const float b = P[no.W1 + min(e, max(n1 - 1, 0))];
            w3v[u] = (e < n3) ? a : 0.0f;
            w1v[u] = (e < n1) ? b : 0.0f;
        }
        const int nr = n_valid * FSRL_RD;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int e = tid + u * NT;
            const float v = (rd != nullptr) ? rd[min(e, max(nr - 1, 0))] : 0.0f;      // rd: kernel-uniform
            rdv[u] = (rd != nullptr && e < nr) ? v : 0.0f;
        }
        const float t1 = P[no.b1 + min(tid, H - 1)], t2 = P[no.b2 + min(tid, H - 1)];
        const float t3 = P[no.b3 + min(tid, no.out - 1)];
        const float t4 = P[max(no.sigma, 0) + min(tid, Da - 1)];
        b1v = (tid < H) ? t1 : 0.0f;
        b2v = (tid < H) ? t2 : 0.0f;
        b3v = (tid < no.out) ? t3 : 0.0f;
        sgv = (no.sigma >= 0 && tid < Da) ? t4 : 0.0f;
    }
    // the small parameters alone, for a caller that fills xv / rdv itself (the SAC actors' forward launch gathers its rows
    // straight into the stage registers)
    __device__ __forceinline__ void issue_params(const float* __restrict__ P, const NetOff no, const int Do, const int Da,
                                                 const int tid) {
        const int n3 = no.out * H, n1 = (Do <= FSRL_W1_LDS) ? H * Do : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * NT;
            const float a = P[no.W3 + min(e, n3 - 1)];
            const float b = P[no.W1 + min(e, max(n1 - 1, 0))];
            w3v[u] = (e < n3) ? a : 0.0f;
            w1v[u] = (e < n1) ? b : 0.0f;
        }
        const float t1 = P[no.b1 + min(tid, H - 1)], t2 = P[no.b2 + min(tid, H - 1)];
        const float t3 = P[no.b3 + min(tid, no.out - 1)];
        const float t4 = P[max(no.sigma, 0) + min(tid, Da - 1)];
        b1v = (tid < H) ? t1 : 0.0f;
        b2v = (tid < H) ? t2 : 0.0f;
        b3v = (tid < no.out) ? t3 : 0.0f;
        sgv = (no.sigma >= 0 && tid < Da) ? t4 : 0.0f;
    }

    __device__ __forceinline__ void commit(TileSmem<H, ROWS>& sm, const NetOff no, const int Do,
                                           const int tid) const {
        // e -> (row i, column k) = (e / Do, e % Do) by a multiply-high with ceil(2^32 / Do): exact for e < 2^16
        const unsigned magic = div_magic(Do);
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int e = tid + u * NT;
            if (e < ROWS * Do) {
                const int i = div_by_magic((unsigned)e, magic), k = e - i * Do;
                sm.xT[k * ROWS + i] = xv[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * NT;
            if (e < no.out * H) sm.w3[e] = w3v[u];
            if (Do <= FSRL_W1_LDS && e < H * Do) sm.w1[e] = w1v[u];
        }
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int e = tid + u * NT;
            if (e < ROWS * FSRL_RD) sm.rd[e] = rdv[u];
        }
        if (tid < H) { sm.b1[tid] = b1v; sm.b2[tid] = b2v; }
        if (tid < FSRL_MAX_ACT) { sm.b3[tid] = b3v; sm.sig[tid] = sgv; }
    }
};

// ---------------------------------------------------------------- forward of one 16-row tile
// Pre: sm.xT filled + __syncthreads() done; wf holds this wave's W2 slice (may still be in
// flight).  Post: sm.h1, sm.h2, sm.out valid (synced).
// R = rows of the tile: 16 (v_mfma_f32_16x16x4_f32) or 4 (v_mfma_f32_4x4x1_16B_f32: the same FLOP
// rate with a quarter of the rows, for launches whose 16-row tiles would leave most CUs idle).  The
// 4-row variant uses the SAME W2 fragment: the 16 blocks of the instruction are (q = k-class, 4
// column quads), each q-class accumulates its quarter of K and the classes are added at the end.
// R = 32: two 16-row MFMA passes per weight fragment (the fragment registers are the same; the per-element K order is
// the 16-row tile's, so a row's result does not depend on the tile height).
// WIDE_HEAD (the replay agents' actors: 2 * act_dim outputs, 16 at act_dim 8): the head Linear of a tile of up to 16 rows on MFMA, split-K
// over the waves, partial 16 x 16 tiles through sm.d2 (free until the backward pass) -- the wave-per-(row, output) loop below takes
// R * out / WAVES dependent LDS -> FMA -> wave_sum trips (16 at out = 16: 5 us of the actors' forward launch, phase probes r6).
template <int H, int R = 16, bool WIDE_HEAD = false>
__device__ __forceinline__ void tile_forward(TileSmem<H, tile_rows(R)>& sm, const float* __restrict__ P,
                                             const NetOff no, const int Do, const int tid,
                                             const FwdW2Frag<H>& wf, unsigned long long* ts = nullptr) {
    constexpr int LD = TileSmem<H>::LD;
    constexpr int WAVES = TileGeom<H>::WAVES;
    constexpr int ROWS = tile_rows(R);
    constexpr int NT = TileGeom<H>::NT;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;

    // ---- layer 1.  Do <= FSRL_W1_LDS: W1 sits in LDS, plain FMA, thread = (column j, group of 4
    //      rows).  Wider inputs (SAC: obs+act = 41, SafetyPointGoal: 60): MFMA with the wave's 16 rows of
    //      W1 fetched in ONE burst of <= 16 independent loads per 64 inputs -- a k-loop of dependent
    //      global loads cost one L2 round trip per input (measured: 20 us of a 27 us kernel).
    if (Do <= FSRL_W1_LDS) {
        for (int t = tid; t < (R / 4) * H; t += NT) {   // thread = (column j, group of 4 rows): R / 4 row groups (one trip up to 16 rows)
            const int j = t % H, rg = t / H;
            const float b = sm.b1[j];
            float acc[4] = {b, b, b, b};
            const float* __restrict__ w = &sm.w1[j * Do];
            for (int k = 0; k < Do; ++k) {
                const float wk = w[k];
                const f32x4 x = *reinterpret_cast<const f32x4*>(&sm.xT[k * ROWS + 4 * rg]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(x[e], wk, acc[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) sm.h1[(4 * rg + e) * LD + j] = fmaxf(acc[e], 0.0f);
        }
    } else {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc8 = {0.f, 0.f, 0.f, 0.f};     // acc8: rows 4..7 of an 8-row tile / rows 16..31 of a 32-row tile
        const float* __restrict__ wrow = P + no.W1 + (size_t)(wave * 16 + li) * Do;
        const int arow = (R >= 16) ? li : (lane & 3);
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
                    const float a = (k < Do) ? sm.xT[k * ROWS + arow] : 0.0f;
                    if constexpr (R >= 16) acc = mfma_16x16x4(a, b[s], acc);
                    else acc = mfma_4x4x1(a, b[s], acc);
                    if constexpr (R == 8) acc8 = mfma_4x4x1((k < Do) ? sm.xT[k * ROWS + 4 + arow] : 0.0f, b[s], acc8);
                    if constexpr (R == 32) acc8 = mfma_16x16x4((k < Do) ? sm.xT[k * ROWS + 16 + arow] : 0.0f, b[s], acc8);
                }
            }
        }
        const int j = wave * 16 + li;
        const float bias = sm.b1[j];
        if constexpr (R < 16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[r] += __shfl_xor(acc[r], 16, 64);
                acc[r] += __shfl_xor(acc[r], 32, 64);
                if constexpr (R == 8) { acc8[r] += __shfl_xor(acc8[r], 16, 64); acc8[r] += __shfl_xor(acc8[r], 32, 64); }
            }
            if (q == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sm.h1[r * LD + j] = fmaxf(acc[r] + bias, 0.0f);
                    if constexpr (R == 8) sm.h1[(4 + r) * LD + j] = fmaxf(acc8[r] + bias, 0.0f);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sm.h1[(4 * q + r) * LD + j] = fmaxf(acc[r] + bias, 0.0f);
                if constexpr (R == 32) sm.h1[(16 + 4 * q + r) * LD + j] = fmaxf(acc8[r] + bias, 0.0f);
            }
        }
    }
    __syncthreads();
    FSRL_TS(ts, 3);

    // ---- layer 2: h2[R,H] = relu(h1[R,H] @ W2^T + b2) on MFMA (fp32)
    if constexpr (R < 16) {           // 4 rows per v_mfma_f32_4x4x1 pass; an 8-row tile runs two passes on the same fragments
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc8 = {0.f, 0.f, 0.f, 0.f};
        const float* arow = &sm.h1[(lane & 3) * LD + 4 * q];
#pragma unroll
        for (int kc = 0; kc < H / 16; ++kc) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * kc);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma_4x4x1(a[s], wf.b[kc][s], acc);
            if constexpr (R == 8) {
                const f32x4 a8 = *reinterpret_cast<const f32x4*>(arow + 4 * LD + 16 * kc);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc8 = mfma_4x4x1(a8[s], wf.b[kc][s], acc8);
            }
        }
        FSRL_TS(ts, 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {            // add the four k-classes: (q0 + q1) + (q2 + q3)
            acc[r] += __shfl_xor(acc[r], 16, 64);
            acc[r] += __shfl_xor(acc[r], 32, 64);
            if constexpr (R == 8) { acc8[r] += __shfl_xor(acc8[r], 16, 64); acc8[r] += __shfl_xor(acc8[r], 32, 64); }
        }
        if (q == 0) {
            const int j = wave * 16 + li;
            const float bias = sm.b2[j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sm.h2[r * LD + j] = fmaxf(acc[r] + bias, 0.0f);
                if constexpr (R == 8) sm.h2[(4 + r) * LD + j] = fmaxf(acc8[r] + bias, 0.0f);
            }
        }
    } else {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};     // acc2: rows 16..31 of a 32-row tile
        const float* arow = &sm.h1[li * LD + 4 * q];
#pragma unroll
        for (int kc = 0; kc < H / 16; ++kc) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * kc);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma_16x16x4(a[s], wf.b[kc][s], acc);
            if constexpr (R == 32) {
                const f32x4 a2 = *reinterpret_cast<const f32x4*>(arow + 16 * LD + 16 * kc);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc2 = mfma_16x16x4(a2[s], wf.b[kc][s], acc2);
            }
        }
        const int j = wave * 16 + li;
        const float bias = sm.b2[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sm.h2[(4 * q + r) * LD + j] = fmaxf(acc[r] + bias, 0.0f);
            if constexpr (R == 32) sm.h2[(16 + 4 * q + r) * LD + j] = fmaxf(acc2[r] + bias, 0.0f);
        }
    }
    __syncthreads();
    FSRL_TS(ts, 5);

    if constexpr (WIDE_HEAD && R <= 16) {        // a 4- or 8-row tile runs the same 16-row MFMAs: the rows beyond R are never read back
        if (no.out > 4) {
            // wave w: k in [16 w, 16 w + 16), lane (li, q) holds h2[li][16 w + 4 q + s] and W3[li][16 w + 4 q + s], s = 0 .. 3
            const f32x4 av = *reinterpret_cast<const f32x4*>(&sm.h2[li * LD + 16 * wave + 4 * q]);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(&sm.w3[li * H + 16 * wave + 4 * q]);     // rows >= out: whatever LDS holds, masked
            const bool on = li < no.out;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc = mfma_16x16x4(av[s2], on ? wv[s2] : 0.0f, acc);
            float* __restrict__ part = sm.d2;                 // [WAVES][16 rows][16 outputs]
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(wave * 16 + 4 * q + r) * 16 + li] = acc[r];
            __syncthreads();
            if (tid < 16 * R) {
                const int i = tid >> 4, o = tid & 15;
                float t = 0.0f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) t += part[(w * 16 + i) * 16 + o];      // fixed order: deterministic
                if (o < no.out) sm.out[i * FSRL_MAX_ACT + o] = t + sm.b3[o];
            }
            __syncthreads();
            return;
        }
    }
    // ---- head (out <= 16): one wave per row, 64-lane shuffle reduce
    for (int idx = wave; idx < R * no.out; idx += WAVES) {      // one wave per (row, output): a 4-row actor tile keeps 8 waves busy
        const int i = idx / no.out, o = idx - i * no.out;
        const float* w3 = &sm.w3[o * H];
        float s = 0.0f;
#pragma unroll
        for (int k = lane; k < H; k += 64) s = fmaf(sm.h2[i * LD + k], w3[k], s);
        s = wave_sum(s);
        if (lane == 0) sm.out[i * FSRL_MAX_ACT + o] = s + sm.b3[o];
    }
    __syncthreads();
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
    float* raw_out;         // optional [N][raw_cols] raw head outputs (SAC actor: mu | log sigma)
    int raw_cols;
    float* sigma_param_out; // optional [Da]: the actor's sigma_param (collector: one launch, no extra copy)
    unsigned* done;         // optional [gridDim.x] in pinned host memory: block b stores `seq` here after its outputs are
    unsigned seq;           //   visible system-wide -- the collector spins on these instead of a stream synchronisation
};

#define LOG_SQRT_2PI 0.9189385332046727f

// Persistent over tiles: workgroup bx of a job takes the 16-row tiles bx, bx + gridDim.x, ...  The W2 fragment (registers)
// and the small parameters (LDS) are loaded ONCE per workgroup; at N = 20 000 the one-tile-per-workgroup version pulled the
// 256 KB fragment set 6 250 times per launch (1.6 GB through the L2s).  The next tile's observations are prefetched into
// registers while the current tile is computed.
template <int H>
__global__ __launch_bounds__(4 * H) void mlp_infer_kernel(const float* __restrict__ P,
                                                         const ModelDesc md, const InferArgs a) {
    __shared__ TileSmem<H> sm;
    constexpr int NT = TileGeom<H>::NT;
    constexpr int NX = TileStage<H>::NX;
    const int tid = threadIdx.x;
    const int job = blockIdx.y;
    const int C = a.C;
    const bool is_actor = (job == 2 * C);
    const int net = is_actor ? 0 : 1 + (C > 0 ? job % C : 0);
    const bool use_next = (!is_actor) && job >= C;
    const NetOff no = md.net[net];
    const int Do = md.Do;
    const int n_tiles = (a.N + 15) >> 4;
    const unsigned magic = div_magic(Do);      // e / Do as a multiply-high (exact for e < 2^16)
    const float* __restrict__ X = use_next ? a.obs_next : a.obs;
    int tile = blockIdx.x;
    int row0 = tile * 16;
    int n_valid = min(16, a.N - row0);
    TileStage<H> stg;
    stg.issue(P, no, Do, md.Da, X + (size_t)row0 * Do, nullptr, n_valid, tid);
    FwdW2Frag<H> wf;
    wf.load(P + no.W2f, tid >> 6, tid & 63);
    stg.commit(sm, no, Do, tid);
    while (true) {
        // prefetch the next tile's observations (registers), consumed after this tile's epilogue
        const int ntile = tile + gridDim.x;
        const int nrow0 = ntile * 16, nn_valid = min(16, a.N - nrow0);
        float xn[NX];
        if (ntile < n_tiles) {
#pragma unroll
            for (int u = 0; u < NX; ++u) {
                const int e = tid + u * NT;
                xn[u] = (e < nn_valid * Do) ? X[(size_t)nrow0 * Do + e] : 0.0f;      // rows are contiguous
            }
        }
        __syncthreads();
        tile_forward<H>(sm, P, no, Do, tid, wf);
        if (a.sigma_param_out && is_actor && tile == 0 && tid < md.Da && no.sigma >= 0) a.sigma_param_out[tid] = sm.sig[tid];
        if (a.raw_out) {
            for (int e = tid; e < n_valid * a.raw_cols; e += NT) {
                const int i = e / a.raw_cols, o = e - i * a.raw_cols;
                a.raw_out[(size_t)(row0 + i) * a.raw_cols + o] = sm.out[i * FSRL_MAX_ACT + o];
            }
        } else if (tid < n_valid) {
            const int r = row0 + tid;
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
                    const float x = sm.out[tid * FSRL_MAX_ACT + d];
                    const float mu = md.unbounded ? x : a.max_action * tanhf(x);     // ActorProb(unbounded=True): mu = head
                    const float sig = expf(sm.sig[d]);
                    if (a.mu_out) a.mu_out[(size_t)r * md.Da + d] = mu;
                    if (a.act) {
                        const float diff = a.act[(size_t)r * md.Da + d] - mu;
                        logp += -(diff * diff) / (2.0f * sig * sig) - logf(sig) - LOG_SQRT_2PI;
                    }
                }
                if (a.logp_old) a.logp_old[r] = logp;
            }
        }
        if (ntile >= n_tiles) break;
        __syncthreads();                       // everybody is done with sm.xT / sm.out of this tile
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int e = tid + u * NT;
            if (e < 16 * Do) { const int i = div_by_magic((unsigned)e, magic), k = e - i * Do; sm.xT[k * 16 + i] = xn[u]; }
        }
        tile = ntile; row0 = nrow0; n_valid = nn_valid;
    }
    if (a.done) {                 // the collector's call (C = 0: actor rows only).  Only the waves that stored to pinned memory wait for their
        // stores -- rows by threads 0 .. 15, raw outputs by elements 0 .. 255: a system-scope fence is an L2 write-back per wave, and
        // sixteen of them serialise (r6: 3.7 us of a 15.7 us resident call)
        if (tid < 256) __atomic_thread_fence(__ATOMIC_RELEASE);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.done + blockIdx.x, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------- the collector's actor, resident for the length of a collect (r6)
// Up to PACTOR_BLOCKS workgroups (one per 16-row tile of the vector env) that stay on their CUs between the vector steps of a collect
// instead of one launch per step: the host rings a DOORBELL in pinned memory ({k rows, sequence number} in one 8-byte word, after
// writing the k observations next to it), thread 0 of every workgroup polls it with system-scope loads, workgroup b runs the actor
// over rows 16 b .. of the request (if it has any) and writes the head outputs and its completion word back to pinned memory, where
// the host spins on them.  The W2 fragment (registers) and the small parameters (LDS) are loaded ONCE per launch -- parameters only
// change in launches that are behind this kernel in stream order.  The arithmetic is mlp_infer_kernel's (same staging, same
// tile_forward, same head): bit-identical actions.
// A workgroup ends on the EXIT command (k = ~0: the host rings it before it enqueues anything else on the stream) and BY ITSELF
// after `timeout_ticks` of the 100 MHz wall clock without a doorbell (or 2^26 polls), so that a host that went away never leaves it
// behind; it then stores its generation number in state[b], which the host checks before it trusts a doorbell to be heard.
#define PACTOR_BLOCKS 4
struct PActorArgs {
    const float* obs;                 // pinned [16 * blocks rows][Do]
    float* mu_out;                    // pinned [16 * blocks rows][Da]; raw_cols > 0 (replay contexts' actors): [rows][raw_cols] raw head outputs
    float* sigma_param_out;           // pinned [Da]
    const unsigned long long* bell;   // pinned: (k << 32) | seq
    unsigned* done;                   // pinned [blocks]: seq of the last request workgroup b served
    unsigned* state;                  // pinned [blocks]: generation of the last kernel whose workgroup b ended
    unsigned gen, last_seq;
    unsigned long long timeout_ticks;
    float max_action;
    int raw_cols;
};
#define PACTOR_EXIT 0xFFFFFFFFu

template <int H, bool RAW>
__global__ __launch_bounds__(4 * H) void actor_resident_kernel(const float* __restrict__ P, const ModelDesc md, const PActorArgs a) {
    __shared__ TileSmem<H> sm;
    __shared__ unsigned k_s, seq_s;
    __shared__ unsigned last_s;                                  // the last sequence number seen (thread 0's; in LDS: the kernel sits at its register cap)
    constexpr int NT = TileGeom<H>::NT;
    constexpr int NX = TileStage<H>::NX;
    const int tid = threadIdx.x, blk = blockIdx.x;
    const NetOff no = md.net[0];
    const int Do = md.Do;
    const unsigned magic = div_magic(Do);
    TileStage<H> stg;
    stg.issue(P, no, Do, md.Da, P, nullptr, 0, tid);          // the small parameters; no rows yet (n_valid = 0: the x loads are masked)
    FwdW2Frag<H> wf;
    wf.load(P + no.W2f, tid >> 6, tid & 63);
    stg.commit(sm, no, Do, tid);
    if (tid == 0) last_s = a.last_seq;
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            const unsigned long long t0 = wall_clock64();
            const unsigned last = last_s;
            unsigned long long v;
            unsigned polls = 0;
            for (;;) {
                v = __hip_atomic_load(a.bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned)v != last) break;
                if (wall_clock64() - t0 > a.timeout_ticks || ++polls > (1u << 26)) { v = (unsigned long long)PACTOR_EXIT << 32; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);             // system scope: the observations were written before the doorbell
            k_s = (unsigned)(v >> 32); seq_s = (unsigned)v;
            last_s = (unsigned)v;
        }
        __syncthreads();
        const unsigned k = k_s;
        if (k == PACTOR_EXIT) break;
        const int row0 = blk * 16, n_valid = min(16, (int)k - row0);
        if (n_valid <= 0) continue;                              // this request has no rows for this workgroup (block-uniform)
        const float* xrow = a.obs + row0 * Do;                   // block-uniform (scalar registers)
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int e = tid + u * NT;
            if (e < 16 * Do) {
                const float x = (e < n_valid * Do)              // rows are contiguous; past the caches (host memory, rewritten between requests)
                    ? __hip_atomic_load(xrow + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0f;
                const int i = div_by_magic((unsigned)e, magic), kk = e - i * Do;
                sm.xT[kk * 16 + i] = x;
            }
        }
        __syncthreads();
        tile_forward<H>(sm, P, no, Do, tid, wf);
        if constexpr (RAW) {                                     // SAC-Lag / DDPG-Lag / CVPO: [mu | log sigma] as the head left them
            for (int e = tid; e < n_valid * a.raw_cols; e += NT) {
                const int i = e / a.raw_cols, o = e - i * a.raw_cols;
                a.mu_out[(row0 + i) * a.raw_cols + o] = sm.out[i * FSRL_MAX_ACT + o];
            }
        } else {
            if (blk == 0 && tid < md.Da && no.sigma >= 0) a.sigma_param_out[tid] = sm.sig[tid];
            if (tid < n_valid) {
                const int r = row0 + tid;
                for (int d = 0; d < md.Da; ++d) {
                    const float x = sm.out[tid * FSRL_MAX_ACT + d];
                    a.mu_out[r * md.Da + d] = md.unbounded ? x : a.max_action * tanhf(x);
                }
            }
        }
        // only the waves that stored to pinned memory wait for their stores (RAW: elements 0 .. 16 * raw_cols - 1 <= 255, else rows by
        // threads 0 .. 15): a system-scope fence is an L2 write-back per wave, sixteen of them serialise
        if (tid < (RAW ? 256 : 64)) __atomic_thread_fence(__ATOMIC_RELEASE);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.done + blk, seq_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (tid == 0) __hip_atomic_store(a.state + blk, a.gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------- fused fwd + loss + bwd
// One PPO minibatch step, activation side.  1-D grid of 8*slots blocks mapped to
// (network, 16-row tile) by xcd_assign(); block = 4*H threads.
// Implements, for its 16 rows: PPOLagrangian.policy_loss / critics_loss gradients
// (fsrl/policy/ppo_lag.py:152-212, lagrangian_base.py:145-166) analytically.
struct PpoBatchPtrs {
    // the pass's batch, already permuted into minibatch order by ppo_permute_rows_kernel
    const float* obs_p;      // [N + pad][Do]
    const float* rd_p;       // [N + pad][FSRL_RD]  act | logp_old | adv_n[c] | ret[c]
    // per-net activation side buffers, row = position inside the minibatch
    float* A1;               // [n_nets][mbp_max][H]   relu(z1)
    float* A2;               // [n_nets][mbp_max][H]   relu(z2)
    float* D1;               // [n_nets][mbp_max][H]   dL/dz1
    float* D2;               // [n_nets][mbp_max][H]   dL/dz2
    float* DO;               // [n_nets][mbp_max][FSRL_DOW]
    float* statp;            // [n_tiles_max][n_nets][4] partial sums of the logged stats
    int mbp_max;
    unsigned long long* ts;  // probe builds: [blocks][16] shader-clock stamps of the phase boundaries (else null)
};

// R = 32 (r6, grouped launches): two 16-row MFMA passes per weight fragment; a row's arithmetic does not depend on its tile's height and
// the per-tile statistics keep their 16-row slots (stat_tile, stat_tile + 1), so 32- and 16-row tiles mix bit-identically.
template <int H, int R>
__device__ __forceinline__ void ppo_fwd_bwd_body(TileSmem<H, tile_rows(R)>& sm, const float* __restrict__ P, const ModelDesc& md,
                                                 const PpoBatchPtrs& bp, const PpoStepArgs& sa, int tile = -1, int net = -1,
                                                 int stat_tile = -1) {
    constexpr int LD = TileSmem<H>::LD;
    constexpr int NT = TileGeom<H>::NT;
    constexpr int ROWS = tile_rows(R);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
    const bool placed = tile >= 0;          // the caller chose (tile, net, stat_tile): the grouped launch's mixed tile heights
    // block id = tile + n_tiles*net: consecutive tiles of one network land on different XCDs
    // (grouping a network's tiles on few XCDs was measured 25% slower: same-line contention)
    // tiles cover round_up(mb_size, 16) rows: the weight-gradient kernel reads whole 16-row groups,
    // rows past mb_size are written as zeros
    const int n_tiles = ((sa.mb_size + 15) >> 4) * (16 / R);
    // Single-agent launches (sa.xcd_pair, round 6): one network per XCD PAIR -- hardware block b runs on XCD b % 8 (observed placement,
    // used for speed only), so network (b % 8) / 2 keeps its weights behind two L2s instead of all eight (memory-side traffic of the
    // launch 16.4 -> ~7 MB); XCDs 6 and 7 stay empty with three networks, the host sizes the grid 8 * ceil(n_tiles / 2).  Same tiles,
    // same arithmetic.  DESIGN_HISTORY 3 (i-e) measured + 3 % on a probe build and left it out; re-tested on the product build, six
    // alternations on one box: 119.7 vs 116.5 updates/s mean, 120.7 vs 117.9 median (step 25.7 vs 26.4 us; the fused kernel itself
    // 13.05 vs 13.0 us -- the gain is what the next two launches no longer wait for).  Grouped launches keep the tile-major order.
    if (placed) {
        // nothing to decode
    } else if (sa.xcd_pair) {
        net = (int)(blockIdx.x & 7) >> 1; tile = 2 * (int)(blockIdx.x >> 3) + (int)(blockIdx.x & 1);
        if (net >= md.n_nets || tile >= n_tiles) return;
    } else {
        tile = blockIdx.x % n_tiles; net = blockIdx.x / n_tiles;
        if (net >= md.n_nets) return;       // grouped launches size grid.x for the largest member's minibatch
    }
    if (!placed) stat_tile = tile;
    const int row0 = placed ? tile : tile * R;      // placed: `tile` IS the first row
    const NetOff no = md.net[net];
    const int Do = md.Do, Da = md.Da, C = md.n_nets - 1;
    const int n_valid = max(0, min(R, sa.mb_size - row0));
    const size_t grow0 = (size_t)sa.mb_start + row0;   // first row of the tile in pass order

    // ---- prologue: ONE burst of independent loads (W2 slice, obs tile, row data, small
    //      parameters); nothing below waits on a second cold round trip.
    if (FSRL_PROBE(sa, 10)) return;                     // pure launch floor of this kernel
    if ((FSRL_PROBE(sa, 15) && net != 0) || (FSRL_PROBE(sa, 16) && net == 0)) return;      // the actor's tiles alone | the critics' alone
    FSRL_TS(bp.ts, 0);
    TileStage<H, ROWS> stg;
    stg.issue(P, no, Do, Da, bp.obs_p + grow0 * Do, bp.rd_p + grow0 * FSRL_RD, n_valid, tid);
    if (FSRL_PROBE(sa, 13)) { asm volatile("" ::"v"(stg.b1v), "v"(stg.xv[0])); return; }
    FwdW2Frag<H> wf;
    if (FSRL_PROBE(sa, 14)) {                           // a quarter of the W2 burst
        const float* row = P + no.W2 + (size_t)(wave * 16 + li) * H + 4 * q;
#pragma unroll
        for (int kc = 0; kc < H / 64; ++kc) wf.b[kc] = *reinterpret_cast<const f32x4*>(row + 16 * kc);
        asm volatile("" ::"v"(stg.b1v), "v"(wf.b[0]), "v"(wf.b[H / 64 - 1]));
        return;
    }
    wf.load(P + no.W2f, wave, lane);
    for (int e = tid; e < ROWS * FSRL_DOW; e += NT) sm.dout[e] = 0.0f;
    if (FSRL_PROBE(sa, 11)) {                           // loads issued, nobody waits for them
        asm volatile("" ::"v"(stg.b1v), "v"(wf.b[0]));
        return;
    }
    FSRL_TS(bp.ts, 1);
    stg.commit(sm, no, Do, tid);
    if (FSRL_PROBE(sa, 12)) {                           // + small loads landed, W2 not awaited
        asm volatile("" ::"v"(wf.b[0]));
        return;
    }
    if (FSRL_PROBE(sa, 9)) return;                      // launch + address setup only
    __syncthreads();
    FSRL_TS(bp.ts, 2);
    if (FSRL_PROBE(sa, 1)) { if (wf.b[0][0] == 123.f && sm.xT[tid] == 1.f) bp.statp[0] = 1.f; return; }
    tile_forward<H, R>(sm, P, no, Do, tid, wf, bp.ts);
    FSRL_TS(bp.ts, 6);
    if (FSRL_PROBE(sa, 4)) { if (sm.out[tid & 15] == 123.f) bp.statp[0] = 1.f; return; }
    const size_t nb = (size_t)net * bp.mbp_max;
    {   // spill relu(z1), relu(z2) for the weight-gradient kernel now: the stores retire while
        // the loss head and the backward GEMM run (coalesced float4)
        float* __restrict__ A1 = bp.A1 + (nb + row0) * H;
        float* __restrict__ A2 = bp.A2 + (nb + row0) * H;
        constexpr int H4 = H / 4;
        for (int e = tid; e < R * H4; e += NT) {
            const int i = e / H4, c4 = e - i * H4;
            store4_next(&A1[(size_t)i * H + 4 * c4], *reinterpret_cast<const f32x4*>(&sm.h1[i * LD + 4 * c4]));
            store4_next(&A2[(size_t)i * H + 4 * c4], *reinterpret_cast<const f32x4*>(&sm.h2[i * LD + 4 * c4]));
        }
    }


    // W2 slice for the backward GEMM dz1 = dz2 @ W2 (column block of this wave): issued now (L2-warm after the forward
    // burst), consumed after the loss head.  The 256 KB ingest is NOT hidden by issuing it earlier: a wave stalls at a VMEM
    // instruction while the CU's memory queue is full, so the burst costs its ~1.4 us wherever it sits (in-kernel stamps,
    // tools/tstamp_probe.py: right after the forward MFMA loop 25 600 cycles per workgroup, interleaved chunk by chunk
    // into that loop 24 700, here 23 900); nor by a fragment-ordered mirror (16 dwordx4 instead of 64 dword loads per
    // lane: same time in this kernel, +1.0 us in the Adam kernel that has to keep the mirror).  DESIGN.md section 3.
    float wb[H / 16][4];
    {
        const float* __restrict__ W2c = P + no.W2 + wave * 16 + li;
#pragma unroll
        for (int jc = 0; jc < H / 16; ++jc) {
#pragma unroll
            for (int s = 0; s < 4; ++s) wb[jc][s] = W2c[(size_t)(16 * jc + 4 * q + s) * H];
        }
    }
    FSRL_TS(bp.ts, 7);
    if (FSRL_PROBE(sa, 5)) { if (wb[0][0] == 123.f) bp.statp[0] = 1.f; return; }
    // ---- loss head: thread (row i = tid>>4, dim d = tid&15), 16-lane shuffles per row
    if (tid < 16 * R) {
        const int i = tid >> 4, d = tid & 15;
        const bool valid = i < n_valid;
        const float* rd = &sm.rd[i * FSRL_RD];
        const float invB = 1.0f / (float)sa.mb_size;
        float st0 = 0.f, st1 = 0.f, st2 = 0.f;
        if (net == 0) {
            float th = 0.f, var = 1.f, df = 0.f, lp = 0.f;
            float hs = sa.max_action;      // d mu / d head = hs * (1 - th * th); an unbounded head: th = 0, hs = 1
            if (d < Da) {
                const float x = sm.out[i * FSRL_MAX_ACT + d];
                th = tanhf(x);
                const float sig = expf(sm.sig[d]);
                var = sig * sig;
                df = rd[d] - sa.max_action * th;
                if (md.unbounded) { df = rd[d] - x; th = 0.0f; hs = 1.0f; }     // ActorProb(unbounded=True): mu = head
                lp = -(df * df) / (2.0f * var) - logf(sig) - LOG_SQRT_2PI;
            }
            // sum over the action dims in ascending order (Independent(Normal).log_prob)
            float logp = 0.0f;
            for (int dd = 0; dd < Da; ++dd) logp += __shfl(lp, (lane & 48) + dd, 64);
            const float lpo = rd[FSRL_RD_LOGP];
            const float ratio = expf(logp - lpo);
            const float ar = rd[FSRL_RD_ADV];
            const float s1 = ratio * ar;
            const float rc = fminf(fmaxf(ratio, 1.0f - sa.eps_clip), 1.0f + sa.eps_clip);
            const float s2 = rc * ar;
            const bool inrange = (ratio >= 1.0f - sa.eps_clip) && (ratio <= 1.0f + sa.eps_clip);
            // d min(s1,s2)/d ratio with torch's tie rule (equal => gradient shared)
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
            if (valid && d < Da) {
                sm.dout[i * FSRL_DOW + d] = dL_dlogp * (df / var) * hs * (1.0f - th * th);
                sm.dout[i * FSRL_DOW + 16 + d] = dL_dlogp * (df * df / var - 1.0f);
            }
            if (valid) { st0 = term; st1 = safety_sum; st2 = lpo - logp; }
        } else {
            const int c = net - 1;
            const float v = sm.out[i * FSRL_MAX_ACT];
            const float dd = rd[FSRL_RD_RET + c] - v;
            float g = -2.0f * dd, vf = dd * dd;
            if (sa.value_clip) {
                // ppo_lag.py:158-164: v_clip = v_old + clamp(v - v_old, -eps, eps); vf = max((ret - v)^2, (ret - v_clip)^2).
                // Gradients as autograd routes them: clamp passes 1 inside [-eps, eps] (bounds included), max splits a tie
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
                if (d == 0) sm.dout[i * FSRL_DOW] = sa.vf_coef * g * invB;
                st0 = vf;
            }
        }
        if (d == 0) { sm.st[i * 4 + 0] = st0; sm.st[i * 4 + 1] = st1; sm.st[i * 4 + 2] = st2; }
    }
    FSRL_TS(bp.ts, 8);
    __syncthreads();
    FSRL_TS(bp.ts, 9);
    if constexpr (R <= 16) {
        if (tid < 4) {   // rows summed in ascending order (fixed => deterministic)
            float t = 0.0f;
            if (tid < 3)
                for (int i = 0; i < R; ++i) t += sm.st[i * 4 + tid];
            bp.statp[((size_t)stat_tile * md.n_nets + net) * 4 + tid] = t;
        }
    } else {
        if (tid < 8) {   // the two 16-row halves in the slots the 16-row tiles would write
            const int half = tid >> 2, f = tid & 3;
            float t = 0.0f;
            if (f < 3)
                for (int i = 0; i < 16; ++i) t += sm.st[(16 * half + i) * 4 + f];
            bp.statp[((size_t)(stat_tile + half) * md.n_nets + net) * 4 + f] = t;
        }
    }

    if (FSRL_PROBE(sa, 6)) { if (wb[0][0] == 123.f) bp.statp[1] = 1.f; return; }
    // ---- dL/dz2 = (dout @ W3) * relu'(z2); thread = (column k, group of 4 rows)
#pragma unroll
    for (int t0 = 0; t0 < (R / 4) * H; t0 += NT) {     // one trip up to 16 rows, two for 32
        const int t = t0 + tid;
        if (t < (R / 4) * H) {
            const int k = t % H, rg = t / H;
            float g[4] = {0.f, 0.f, 0.f, 0.f};
            for (int o = 0; o < no.out; ++o) {
                const float w = sm.w3[o * H + k];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = fmaf(sm.dout[(4 * rg + e) * FSRL_DOW + o], w, g[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * rg + e;
                sm.d2[i * LD + k] = (sm.h2[i * LD + k] > 0.0f) ? g[e] : 0.0f;
            }
        }
    }
    __syncthreads();
    FSRL_TS(bp.ts, 10);

    {   // dz2 and dout tiles -> side buffers (retire under the backward GEMM)
        float* __restrict__ D2 = bp.D2 + (nb + row0) * H;
        constexpr int H4 = H / 4;
        for (int e = tid; e < R * H4; e += NT) {
            const int i = e / H4, c4 = e - i * H4;
            store4_next(&D2[(size_t)i * H + 4 * c4], *reinterpret_cast<const f32x4*>(&sm.d2[i * LD + 4 * c4]));
        }
        float* __restrict__ DOb = bp.DO + (nb + row0) * FSRL_DOW;
        for (int e = tid; e < R * FSRL_DOW; e += NT) DOb[e] = sm.dout[e];
    }
    FSRL_TS(bp.ts, 11);
    // ---- dL/dz1 = (dz2 @ W2) * relu'(z1) on MFMA; result goes straight to L2/HBM
    if constexpr (R < 16) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc8 = {0.f, 0.f, 0.f, 0.f};
        const float* arow = &sm.d2[(lane & 3) * LD + 4 * q];
#pragma unroll
        for (int jc = 0; jc < H / 16; ++jc) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * jc);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma_4x4x1(a[s], wb[jc][s], acc);
            if constexpr (R == 8) {
                const f32x4 a8 = *reinterpret_cast<const f32x4*>(arow + 4 * LD + 16 * jc);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc8 = mfma_4x4x1(a8[s], wb[jc][s], acc8);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[r] += __shfl_xor(acc[r], 16, 64);
            acc[r] += __shfl_xor(acc[r], 32, 64);
            if constexpr (R == 8) { acc8[r] += __shfl_xor(acc8[r], 16, 64); acc8[r] += __shfl_xor(acc8[r], 32, 64); }
        }
        FSRL_TS(bp.ts, 12);
        if (FSRL_PROBE(sa, 7)) { if (acc[0] == 123.f) bp.statp[1] = 1.f; return; }
        if (q == 0) {
            float* __restrict__ D1 = bp.D1 + (nb + row0) * H;
            const int col = wave * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                D1[(size_t)r * H + col] = (sm.h1[r * LD + col] > 0.0f) ? acc[r] : 0.0f;
                if constexpr (R == 8) D1[(size_t)(4 + r) * H + col] = (sm.h1[(4 + r) * LD + col] > 0.0f) ? acc8[r] : 0.0f;
            }
        }
    } else {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};     // acc2: rows 16..31 of a 32-row tile, same fragment
        const float* arow = &sm.d2[li * LD + 4 * q];
#pragma unroll
        for (int jc = 0; jc < H / 16; ++jc) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * jc);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma_16x16x4(a[s], wb[jc][s], acc);
            if constexpr (R == 32) {
                const f32x4 a2 = *reinterpret_cast<const f32x4*>(arow + 16 * LD + 16 * jc);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc2 = mfma_16x16x4(a2[s], wb[jc][s], acc2);
            }
        }
        if (FSRL_PROBE(sa, 7)) { if (acc[0] == 123.f) bp.statp[1] = 1.f; return; }
        float* __restrict__ D1 = bp.D1 + (nb + row0) * H;
        const int col = wave * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            D1[(size_t)i * H + col] = (sm.h1[i * LD + col] > 0.0f) ? acc[r] : 0.0f;
            if constexpr (R == 32) D1[(size_t)(16 + i) * H + col] = (sm.h1[(16 + i) * LD + col] > 0.0f) ? acc2[r] : 0.0f;
        }
    }
    FSRL_TS(bp.ts, 13);
}

template <int H, int R>
__global__ __launch_bounds__(4 * H) void ppo_fwd_bwd_kernel(const float* __restrict__ P,
                                                           const ModelDesc md,
                                                           const PpoBatchPtrs bp,
                                                           const PpoStepArgs sa) {
    __shared__ TileSmem<H> sm;
    ppo_fwd_bwd_body<H, R>(sm, P, md, bp, sa);
}

// r6 (late): tall tiles.  Once a launch's 16-row tiles exceed the CU count (a group of 8 members x 3 networks x 16 tiles = 384 workgroups on
// 256 CUs; one agent at minibatches above ~1 360 rows), the first `n32` tiles of every network are 32 rows tall (two MFMA passes per weight
// fragment: half the L2 -> register weight ingest per row) and the rest stay 16 rows: block b of a network covers rows [32 b, 32 b + 32) for
// b < n32 and 16 rows behind them otherwise; per_net = n32 + the remaining 16-row tiles, grid.x = n_nets * per_net.  A row's arithmetic and
// the 16-row statistic slots do not depend on the tile height (ppo_fwd_bwd_body), so every plan gives the same bits.
template <int H>
__device__ __forceinline__ void ppo_fwd_bwd_mixed(unsigned char* raw, const float* __restrict__ P, const ModelDesc& md,
                                                  const PpoBatchPtrs& bp, const PpoStepArgs& sa, const int n32, const int per_net) {
    static_assert(sizeof(TileSmem<H, 32>) >= sizeof(TileSmem<H, 16>), "the 16-row layout lives inside the 32-row one");
    const int net = (int)blockIdx.x / per_net, b = (int)blockIdx.x - net * per_net;
    const int row0 = b < n32 ? 32 * b : 32 * n32 + 16 * (b - n32);
    if (net >= md.n_nets || row0 >= ((sa.mb_size + 15) & ~15)) return;        // a smaller member of a group has fewer tiles
    if (b < n32) ppo_fwd_bwd_body<H, 32>(*reinterpret_cast<TileSmem<H, 32>*>(raw), P, md, bp, sa, row0, net, row0 >> 4);
    else ppo_fwd_bwd_body<H, 16>(*reinterpret_cast<TileSmem<H, 16>*>(raw), P, md, bp, sa, row0, net, row0 >> 4);
}
template <int H>
__global__ __launch_bounds__(4 * H) void ppo_fwd_bwd_tall_kernel(const float* __restrict__ P, const ModelDesc md, const PpoBatchPtrs bp,
                                                                const PpoStepArgs sa, const int n32, const int per_net) {
    __shared__ __align__(16) unsigned char raw[sizeof(TileSmem<H, 32>)];
    ppo_fwd_bwd_mixed<H>(raw, P, md, bp, sa, n32, per_net);
}

// ---------------------------------------------------------------- weight gradients
// block = 1024 threads (16 waves).  grid.x = wg_grid(H, n_nets) = n_nets * (NT2 + 2 NA) + n_nets + 1:
//   NT2  = (H/32)^2 MFMA tile blocks (dW2, 32x32 outputs, 16-way split-K over the waves)
//   2 NA = H/32 aux blocks for dW1 + db1 and H/32 for dW3 + db2 (32 columns each; also MFMA, 16-way split-K)
//   + n_nets blocks reducing the column sums of the dout side buffer (db3, dsigma) of one network each
//   + 1 block that finalises the logged statistics of this step.
// Every block emits the sum of squares of the gradient entries it produced
// (clip_grad_norm_, ppo_lag.py:237-240).
struct WgradPtrs {
    const float* A1; const float* A2; const float* D1; const float* D2; const float* DO;
    const float* X;    // obs_p + mb_start*Do : the minibatch's observation rows
    float* grad;       // flat, same layout as the parameters
    float* gsq_part;   // [gridDim.x]
    CtrlBlock* ctrl;
    const float* P;    // parameters (entropy of the logged stats)
    const float* statp;
    float* stats;      // [steps][FSRL_PPO_NSTATS]
    int mbp_max;
    float* Pw; float* M; float* V;   // fused-Adam instantiation (FUSE): parameters and Adam moments, updated in place
    float* gsq_net;    // optional [n_nets]: the extra block's share of the squared norm (db3, dsigma) PER NETWORK -- FOCOPS clips the
                       // actor alone (focops.py:205-213); with `stats` null the extra block skips the PPO row
};

// torch.optim.Adam single-tensor update of ONE element with an unclipped gradient -- the same operations in the same
// order as adam_clip_kernel with coef == 1 (g * 1.0f == g), so the two launch sequences are bit-identical.  Used by the
// weight-gradient kernel when max_grad_norm is off (PPOLagAgent's default, fsrl/agent/ppo_lag_agent.py:97): every gradient
// element is final the moment its block has reduced it, nothing global (no norm) stands between it and the update.
// `mi`: position of the element in the forward-fragment mirror of W2, or -1 (the caller knows: only the dW2 tiles have one).
__device__ __forceinline__ void ppo_adam_elem(const WgradPtrs& wp, const PpoStepArgs& sa, const int i, const float g,
                                              const int mi = -1) {
    float m = wp.M[i], v = wp.V[i], p = wp.Pw[i];
    ppo_adam_math(g, m, v, p, sa);
    wp.M[i] = m; wp.V[i] = v; wp.Pw[i] = p;
    if (mi >= 0) wp.Pw[mi] = p;
}

// grid of a weight-gradient launch (host and device): per network (H/32)^2 dW2 tiles + 2 x H/32 aux parts, then one extra block per
// network and the logged-row block; every block leaves one squared-norm partial in gsq_part
__host__ __device__ constexpr int wg_blocks_per_net(int H) { return (H / 32) * (H / 32) + 2 * (H / 32); }
__host__ __device__ constexpr int wg_grid(int H, int n_nets) { return n_nets * wg_blocks_per_net(H) + n_nets + 1; }

#define WG_MAXU 8      // k-steps per wave and load burst in the tile role (512 rows per burst)
#define AUX_MAXU 32    // rows per thread in the aux role:   mbp/16   <= 32

// Logged statistics of one minibatch step (parameters are still pre-update here, like the
// reference which builds `dist` before optim.step, ppo_lag.py:225-247).
__device__ __forceinline__ void ppo_stats_finalize(const ModelDesc& md, const WgradPtrs& wp,
                                                   const PpoStepArgs& sa, int n_tiles, int lane) {
    const int nn = md.n_nets, C = nn - 1;
    // lane = (tile group g = lane>>4, field slot f = lane&15 -> (net, field)); every lane issues
    // all of its loads at once (<= 8 tiles), then the 4 groups are combined in a fixed order
    float mine = 0.0f;
    {
        const int g = lane >> 4, fs = lane & 15;
        for (int t0 = 0; t0 < n_tiles; t0 += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + g + 4 * u;
                v[u] = (fs < nn * 4 && t < n_tiles) ? wp.statp[(size_t)t * nn * 4 + fs] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) mine += v[u];
        }
        mine += __shfl_xor(mine, 16, 64);
        mine += __shfl_xor(mine, 32, 64);
    }
    const float invB = 1.0f / (float)sa.mb_size;
    const float term = __shfl(mine, 0, 64), safety = __shfl(mine, 1, 64), kls = __shfl(mine, 2, 64);
    float vf[FSRL_MAX_CRITICS];
#pragma unroll
    for (int c = 0; c < FSRL_MAX_CRITICS; ++c) vf[c] = __shfl(mine, 4 * (c + 1), 64) * invB;
    if (lane == 0) {
        float ent = 0.0f;
        for (int d = 0; d < md.Da; ++d)
            ent += 1.4189385332046727f + logf(expf(wp.P[md.net[0].sigma + d]));
        const float actor_rew = -term * invB;
        const float actor_safety = sa.use_lagrangian ? safety * invB : 0.0f;
        const float actor_total = sa.rescale * (actor_rew + actor_safety);
        const float kl = kls * invB;
        float vf_total = 0.0f;
#pragma unroll
        for (int c = 0; c < FSRL_MAX_CRITICS; ++c)
            if (c < C) vf_total += vf[c];
        float* o = wp.stats + (size_t)sa.step * FSRL_PPO_NSTATS;
        o[0] = sa.rescale;
        o[1] = (sa.use_lagrangian && C > 1) ? sa.lam[0] : 0.0f;
        o[2] = actor_safety;
        o[3] = actor_rew;
        o[4] = actor_total;
        o[5] = kl;
        o[6] = vf[0];
        o[7] = (C > 1) ? vf[1] : 0.0f;
        o[8] = vf_total;
        o[9] = actor_total + sa.vf_coef * vf_total;
        o[10] = ent;
        wp.ctrl->kl_sum = (sa.first_in_pass ? 0.0 : wp.ctrl->kl_sum) + (double)kl;
    }
}

// The extra blocks of a weight-gradient launch (r6 late: one per role; up to r6 early ONE block walked all of it -- the logged row's
// chain of dependent cold loads, then the networks one after the other: 2.6 us over the launch floor where a dW2 tile takes 1.0):
//   e < n_nets : db3 / dsigma of network e (column sums of its dout side buffer) and their share of the squared gradient norm
//   e == n_nets: the step's logged row (FUSE: the block of network 0 writes it first -- it reads the sigma_param its own Adam step
//                is about to change -- and this block only leaves a zero behind)
// `red`: at least 32 x 33 floats of LDS.
template <bool BIG, bool FUSE>
__device__ __forceinline__ void ppo_wgrad_extra_block(const ModelDesc& md, const WgradPtrs& wp, const int mbp, const PpoStepArgs& sa,
                                                      const int n_stat_tiles, float* red, const int e) {
    const int CH = BIG ? (mbp + 511) / 512 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (FSRL_PROBE(sa, 21)) return;
    const bool stats_here = FUSE ? (e == 0) : (e == md.n_nets);
    if (stats_here && wave == 0 && wp.stats) {
        ppo_stats_finalize(md, wp, sa, n_stat_tiles, lane);
        // fused mode: the pass-level KL stop that adam_clip_kernel's first block decides otherwise (ppo_lag.py:251-255)
        if (FUSE && lane == 0 && sa.last_in_pass && sa.target_kl > 0.0f) {
            const double mean_kl = wp.ctrl->kl_sum / ((double)sa.iters_in_pass + 1e-7);
            if (mean_kl > sa.kl_thresh) wp.ctrl->stopped_after = sa.pass;
        }
    }
    if (e == md.n_nets) {                     // nothing of the gradient is produced here
        if (tid == 0) wp.gsq_part[blockIdx.x] = 0.0f;
        return;
    }
    // db3[o] / dsigma[d] = column sums of DO over the minibatch rows:
    // thread (col = tid & 31, row phase = tid >> 5); all loads of a thread in one burst
    float sqs = 0.0f;
    {
        const int net = e;
        const NetOff no = md.net[net];
        const float* __restrict__ DOn = wp.DO + (size_t)net * wp.mbp_max * FSRL_DOW;
        const int col = tid & 31, php = tid >> 5;
        float t = 0.0f;
        for (int ch = 0; ch < CH; ++ch) {
            float v[16];
            int rbase = 512 * ch;
            if constexpr (BIG) asm volatile("" : "+s"(rbase));
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = rbase + php + 32 * u;
                v[u] = (r < mbp) ? DOn[(size_t)r * FSRL_DOW + col] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) t += v[u];
        }
        __syncthreads();
        red[php * 33 + col] = t;
        __syncthreads();
        if (tid < 32) {
            float tot = 0.0f;
#pragma unroll
            for (int p2 = 0; p2 < 32; ++p2) tot += red[p2 * 33 + tid];
            if (tid < no.out) {
                wp.grad[no.b3 + tid] = tot; sqs = fmaf(tot, tot, sqs);
                if constexpr (FUSE) ppo_adam_elem(wp, sa, no.b3 + tid, tot);
            }
            if (no.sigma >= 0 && tid >= 16 && tid < 16 + md.Da) {
                wp.grad[no.sigma + tid - 16] = tot; sqs = fmaf(tot, tot, sqs);
                if constexpr (FUSE) ppo_adam_elem(wp, sa, no.sigma + tid - 16, tot);   // wave 0 logged the entropy first
            }
        }
    }
    if (wave == 0) {
        sqs = wave_sum(sqs);                  // lanes >= 32 hold 0
        if (lane == 0) {
            wp.gsq_part[blockIdx.x] = sqs;
            if (wp.gsq_net) wp.gsq_net[e] = sqs;          // this network's share on its own (FOCOPS clips the actor alone)
        }
    }
}

// ---- aux roles of the weight-gradient launch: 32 columns j0..j0+31 of one network, the 16-way split-K MFMA structure of the dW2 tiles.
//        PA: dW1[j][k] = sum_r D1[r][j] * X[r][k]  (A = D1, B = x row)  and  db1[j] = column sums of D1
//        PB: dW3[o][j] = sum_r A2[r][j] * DO[r][o] (A = A2, B = dout row) and  db2[j] = column sums of D2
//      r6 (late): one workgroup per PART (up to r6 early one workgroup did both: 5 operand streams, 128 KB through one CU's L1 at 256 rows
//      where a dW2 tile moves 64 KB -- the role timing of a probe build: launch floor 4.4 us, tiles alone 5.4, aux blocks alone 7.2, with
//      D1 / X only 6.2).  Per element the loads, MFMA order and the two-round LDS reduction are unchanged: same bits.
//      (v1 did this with VALU + LDS broadcasts on 12 blocks and took 16 us.)
template <int H, bool BIG, bool FUSE, int U, bool PA, bool PB>
__device__ __forceinline__ void ppo_wgrad_aux(const ModelDesc& md, const WgradPtrs& wp, const int mbp, const PpoStepArgs& sa,
                                              const NetOff& no, const float* __restrict__ A2, const float* __restrict__ D1,
                                              const float* __restrict__ D2, const float* __restrict__ DOb, const int j0,
                                              float* red, float& sq) {
    const int CH = BIG ? (mbp + 64 * U - 1) / (64 * U) : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int Do = md.Do, out = no.out;
    const int KS = mbp >> 2;
    const float* __restrict__ X = wp.X;
    for (int k0 = 0; k0 < (PA ? Do : 1); k0 += 16) {
        const bool first = (k0 == 0);
        f32x4 ax0 = {0, 0, 0, 0}, ax1 = {0, 0, 0, 0}, ad0 = {0, 0, 0, 0}, ad1 = {0, 0, 0, 0};
        f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
        for (int ch = 0; ch < CH; ++ch)
        for (int ub = 0; ub < U; ub += 4) {       // 4 k-steps per burst (64 rows / wave set)
            int sbase = 16 * U * ch + 16 * ub;
            if constexpr (BIG) asm volatile("" : "+s"(sbase));             // see the tile role: nothing of a burst is hoisted
            if (sbase + wave >= KS) break;                                  // wave-uniform
            f32x2 a1[4], a2[4], a3[4];
            float bx[4], bd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {               // one burst of independent loads
                const int sidx = sbase + wave + 16 * u;
                a1[u] = f32x2{0.f, 0.f}; a2[u] = f32x2{0.f, 0.f}; a3[u] = f32x2{0.f, 0.f};
                bx[u] = 0.f; bd[u] = 0.f;
                if (sidx < KS) {
                    const size_t r = (size_t)(4 * sidx + q);
                    if constexpr (PA) {
                        a1[u] = *reinterpret_cast<const f32x2*>(D1 + r * H + j0 + 2 * c);
                        if (k0 + c < Do) bx[u] = X[r * Do + k0 + c];
                    }
                    if constexpr (PB) {
                        if (first) {
                            a2[u] = *reinterpret_cast<const f32x2*>(A2 + r * H + j0 + 2 * c);
                            a3[u] = *reinterpret_cast<const f32x2*>(D2 + r * H + j0 + 2 * c);
                            bd[u] = DOb[r * FSRL_DOW + c];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {               // zero operands beyond KS add nothing
                if constexpr (PA) {
                    ax0 = mfma_16x16x4(a1[u][0], bx[u], ax0);
                    ax1 = mfma_16x16x4(a1[u][1], bx[u], ax1);
                    if (first) s1 += a1[u];
                }
                if constexpr (PB) {
                    if (first) {
                        ad0 = mfma_16x16x4(a2[u][0], bd[u], ad0);
                        ad1 = mfma_16x16x4(a2[u][1], bd[u], ad1);
                        s2 += a3[u];
                    }
                }
            }
        }
        // bias sums: add the 4 k-slots (q) of the wave; lanes q==0 then hold columns 2c, 2c+1
        if (first) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if constexpr (PA) { s1[t] += __shfl_xor(s1[t], 16, 64); s1[t] += __shfl_xor(s1[t], 32, 64); }
                if constexpr (PB) { s2[t] += __shfl_xor(s2[t], 16, 64); s2[t] += __shfl_xor(s2[t], 32, 64); }
            }
        }
        // slot layout (1088 floats): [0,512) dW1 tile [32 j][16 k], [512,1024) dW3^T tile
        // [32 j][16 o], [1024,1056) db1[32], [1056,1088) db2[32].  Two rounds over 8 slots.
        float* slot = red + (wave & 7) * 1088;
        __syncthreads();
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            if ((wave >> 3) == round) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jl = 2 * (4 * q + r);
                    if (round == 0) {
                        if constexpr (PA) {
                            slot[(jl + 0) * 16 + c] = ax0[r];
                            slot[(jl + 1) * 16 + c] = ax1[r];
                        }
                        if constexpr (PB) {
                            if (first) {
                                slot[512 + (jl + 0) * 16 + c] = ad0[r];
                                slot[512 + (jl + 1) * 16 + c] = ad1[r];
                            }
                        }
                    } else {
                        if constexpr (PA) {
                            slot[(jl + 0) * 16 + c] += ax0[r];
                            slot[(jl + 1) * 16 + c] += ax1[r];
                        }
                        if constexpr (PB) {
                            if (first) {
                                slot[512 + (jl + 0) * 16 + c] += ad0[r];
                                slot[512 + (jl + 1) * 16 + c] += ad1[r];
                            }
                        }
                    }
                }
                if (first && q == 0) {
                    if (round == 0) {
                        if constexpr (PA) { slot[1024 + 2 * c] = s1[0]; slot[1024 + 2 * c + 1] = s1[1]; }
                        if constexpr (PB) { slot[1056 + 2 * c] = s2[0]; slot[1056 + 2 * c + 1] = s2[1]; }
                    } else {
                        if constexpr (PA) { slot[1024 + 2 * c] += s1[0]; slot[1024 + 2 * c + 1] += s1[1]; }
                        if constexpr (PB) { slot[1056 + 2 * c] += s2[0]; slot[1056 + 2 * c + 1] += s2[1]; }
                    }
                }
            }
            __syncthreads();
        }
        // final: threads [0,512) dW1, [512,1024) dW3^T (+ the first 64 of them the biases)
        {
            const int e = tid & 511;
            float v = 0.0f;
            const int off = (tid < 512) ? e : 512 + e;
            if ((PA && tid < 512) || (PB && tid >= 512)) {
#pragma unroll
                for (int w = 0; w < 8; ++w) v += red[w * 1088 + off];
            }
            const int jl = e >> 4, kk = e & 15;
            if (tid < 512) {
                if constexpr (PA) {
                    if (k0 + kk < Do) {
                        const int gi = no.W1 + (j0 + jl) * Do + k0 + kk;
                        wp.grad[gi] = v;
                        sq = fmaf(v, v, sq);
                        if constexpr (FUSE) ppo_adam_elem(wp, sa, gi, v);
                    }
                }
            } else if (PB && first && kk < out) {
                const int gi = no.W3 + kk * H + j0 + jl;
                wp.grad[gi] = v;
                sq = fmaf(v, v, sq);
                if constexpr (FUSE) ppo_adam_elem(wp, sa, gi, v);
            }
            if (first && ((PA && tid < 32) || (PB && tid >= 32 && tid < 64))) {
                float bsum = 0.0f;
#pragma unroll
                for (int w = 0; w < 8; ++w) bsum += red[w * 1088 + 1024 + tid];
                const int gi = (tid < 32) ? no.b1 + j0 + tid : no.b2 + j0 + tid - 32;
                wp.grad[gi] = bsum;
                sq = fmaf(bsum, bsum, sq);
                if constexpr (FUSE) ppo_adam_elem(wp, sa, gi, bsum);
            }
        }
    }
}

// BIG = false: mbp <= 512, every role is one straight-line load burst (the common case: batch <= 256).
// BIG = true: the same code inside a loop over 512-row chunks (merged last minibatch of batch 512: 1023).
// FUSE: apply Adam to every gradient element as soon as it is reduced (max_grad_norm off); a separate instantiation so
// that the clipped path keeps its register budget (one kernel with a runtime switch spilled 42 VGPRs).
// U: k-steps per wave and load burst in the tile role (a chunk = 64 U rows).  8 for the single-agent launches; the grouped launches
// take 4 -- a 256-row minibatch needs no more -- to fit 64 VGPRs, so that TWO workgroups share a CU (r6).
template <int H, bool BIG, bool FUSE, int U = WG_MAXU>
__device__ __forceinline__ void ppo_wgrad_body(const ModelDesc& md, const WgradPtrs& wp, const int mbp,
                                               const PpoStepArgs& sa, const int n_stat_tiles) {
    const int CH = BIG ? (mbp + 64 * U - 1) / (64 * U) : 1;      // row chunks
    constexpr int TPD = H / 32;          // tiles per dimension
    constexpr int NT2 = TPD * TPD;
    constexpr int NA = H / 32;
    constexpr int PB = NT2 + 2 * NA;     // per network: the dW2 tiles, then the two aux parts (dW1 + db1 | dW3 + db2) of every 32 columns
    static_assert(PB == wg_blocks_per_net(H), "host and device agree on the grid");
    __shared__ float red[1024 * 9];      // 8 split-K partial slots of a 32x32 tile / aux reduce scratch
    __shared__ float wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (FSRL_PROBE(sa, 20)) return;
    if ((int)blockIdx.x >= md.n_nets * PB) {  // the extra blocks: one per network + the logged row
        ppo_wgrad_extra_block<BIG, FUSE>(md, wp, mbp, sa, n_stat_tiles, red, (int)blockIdx.x - md.n_nets * PB);
        return;
    }
    const int net = blockIdx.x / PB, rb = blockIdx.x % PB;
    const NetOff no = md.net[net];
    const size_t nb = (size_t)net * wp.mbp_max;
    const float* __restrict__ A1 = wp.A1 + nb * H;
    const float* __restrict__ A2 = wp.A2 + nb * H;
    const float* __restrict__ D1 = wp.D1 + nb * H;
    const float* __restrict__ D2 = wp.D2 + nb * H;
    const float* __restrict__ DOb = wp.DO + nb * FSRL_DOW;
    float sq = 0.0f;

    if (FSRL_PROBE(sa, 23)) return;
    if (FSRL_PROBE(sa, 21) && rb >= NT2) return;
    if (FSRL_PROBE(sa, 22) && rb < NT2) return;
    if (rb < NT2) {
        // ---- dW2[j][k] = sum_r D2[r][j] * A1[r][k]; wave w takes k-steps s = w, w+16, ...
        const int tj = rb / TPD, tk = rb % TPD;
        const int c = lane & 15, q = lane >> 4;
        const float* __restrict__ pa = D2 + tj * 32 + 2 * c;
        const float* __restrict__ pb = A1 + tk * 32 + 2 * c;
        const int KS = mbp >> 2;
        f32x4 acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
        for (int ch = 0; ch < CH; ++ch) {
            int s0 = 16 * U * ch;
            // BIG: the chunk base is made opaque to the optimiser.  Left visible, it hoists the 64-bit address of every load of the
            // burst out of the chunk loop (2 x 8 address pairs here, 5 x 4 in the aux role) and spills them: 167-223 VGPRs, 472-544
            // bytes of scratch per lane in every BIG instantiation (round 5's code-object notes).  One add per load instead.
            if constexpr (BIG) asm volatile("" : "+s"(s0));
            f32x2 a[U], b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = s0 + wave + 16 * u;
                if (s < KS) {
                    const size_t r = (size_t)(4 * s + q) * H;
                    a[u] = *reinterpret_cast<const f32x2*>(pa + r);
                    b[u] = *reinterpret_cast<const f32x2*>(pb + r);
                } else {
                    a[u] = f32x2{0.f, 0.f};
                    b[u] = f32x2{0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (s0 + wave + 16 * u < KS) {   // wave-uniform
                    acc00 = mfma_16x16x4(a[u][0], b[u][0], acc00);
                    acc01 = mfma_16x16x4(a[u][0], b[u][1], acc01);
                    acc10 = mfma_16x16x4(a[u][1], b[u][0], acc10);
                    acc11 = mfma_16x16x4(a[u][1], b[u][1], acc11);
                }
            }
        }
        // acc_tu[r]: j_local = 2*(4q+r)+t, k_local = 2c+u.  Two rounds: waves 0-7 store their
        // partial tile, then waves 8-15 add theirs into the same slot (fixed order).
        float* myred = red + (wave & 7) * (32 * 33);
        if (wave < 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jl = 2 * (4 * q + r);
                myred[(jl + 0) * 33 + 2 * c + 0] = acc00[r];
                myred[(jl + 0) * 33 + 2 * c + 1] = acc01[r];
                myred[(jl + 1) * 33 + 2 * c + 0] = acc10[r];
                myred[(jl + 1) * 33 + 2 * c + 1] = acc11[r];
            }
        }
        __syncthreads();
        if (wave >= 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jl = 2 * (4 * q + r);
                myred[(jl + 0) * 33 + 2 * c + 0] += acc00[r];
                myred[(jl + 0) * 33 + 2 * c + 1] += acc01[r];
                myred[(jl + 1) * 33 + 2 * c + 0] += acc10[r];
                myred[(jl + 1) * 33 + 2 * c + 1] += acc11[r];
            }
        }
        __syncthreads();
        {
            const int jl = tid >> 5, kl = tid & 31;   // 1024 threads = 32x32 outputs
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[w * (32 * 33) + jl * 33 + kl];
            const int gi = no.W2 + (tj * 32 + jl) * H + tk * 32 + kl;
            wp.grad[gi] = v;
            sq = v * v;
            if constexpr (FUSE)
                ppo_adam_elem(wp, sa, gi, v, no.W2f + w2f_index(H, tj * 32 + jl, tk * 32 + kl));
        }
    } else if (rb < NT2 + NA) {
        ppo_wgrad_aux<H, BIG, FUSE, U, true, false>(md, wp, mbp, sa, no, A2, D1, D2, DOb, (rb - NT2) * 32, red, sq);
    } else {
        ppo_wgrad_aux<H, BIG, FUSE, U, false, true>(md, wp, mbp, sa, no, A2, D1, D2, DOb, (rb - NT2 - NA) * 32, red, sq);
    }
    // ---- block sum of squares (fixed order => deterministic)
    sq = wave_sum(sq);
    __syncthreads();
    if (lane == 0) wsum[wave] = sq;
    __syncthreads();
    if (tid == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += wsum[w];
        wp.gsq_part[blockIdx.x] = t;
    }
}

// The logged row of one step on its own: used when the weight gradients of a LARGE minibatch (more than 512 rows) go through
// fb_wgrad_kernel (split-K over workgroups) instead of ppo_wgrad_kernel, whose extra block writes the row otherwise.
__global__ __launch_bounds__(64) void ppo_stats_kernel(const ModelDesc md, const WgradPtrs wp, const PpoStepArgs sa,
                                                      const int n_stat_tiles) {
    ppo_stats_finalize(md, wp, sa, n_stat_tiles, (int)threadIdx.x);
}

template <int H, bool BIG, bool FUSE>
__global__ __launch_bounds__(1024) void ppo_wgrad_kernel(const ModelDesc md, const WgradPtrs wp,
                                                        const int mbp, const PpoStepArgs sa,
                                                        const int n_stat_tiles) {
    ppo_wgrad_body<H, BIG, FUSE>(md, wp, mbp, sa, n_stat_tiles);
}

// ---------------------------------------------------------------- grouped launches (several agents, one stream)
// k independent agents of one shape (multi-seed runs: SURVEY 8e "within-GPU batching of k seeds") step in lock step:
// every launch of the minibatch step carries all members as grid.y.  One agent's step is a chain of dependent launches
// whose fixed costs (launch floor, cold first touch of the freshly written parameters) dominate; k agents share each of
// them.  The arithmetic per member is the single-agent body, inlined; grouped and one-by-one updates are bit-identical when
// the tile shapes agree (a group of one, minibatches <= 512 rows) and within the golden tolerances otherwise.
struct GroupAgent {          // per member; device memory, rewritten at the start of every grouped update
    const float* P;
    float *Pw, *M, *V, *G;
    PpoBatchPtrs bp;
    WgradPtrs wp;            // wp.X is per step: obs_p + mb_start * Do, formed in the kernel
    const float* gsq_part; CtrlBlock* ctrl;
    float rescale; float lam[FSRL_MAX_CRITICS];
    int n_dev, nparts;
};
struct GroupStep {           // per (minibatch index of the pass, member); device memory, rewritten every pass
    int mb_start, mb_size, step, mb_index;
    int first_in_pass, last_in_pass, iters_in_pass, active;
    int pass; float step_size, bc2_sqrt; int pad;
};
__device__ __forceinline__ PpoStepArgs group_step_args(const PpoStepArgs& base, const GroupAgent& a, const GroupStep& st) {
    PpoStepArgs sa = base;
    sa.mb_start = st.mb_start; sa.mb_size = st.mb_size; sa.step = st.step; sa.mb_index = st.mb_index;
    sa.first_in_pass = st.first_in_pass; sa.last_in_pass = st.last_in_pass; sa.iters_in_pass = st.iters_in_pass;
    sa.pass = st.pass; sa.step_size = st.step_size; sa.bc2_sqrt = st.bc2_sqrt;
    sa.rescale = a.rescale;
#pragma unroll
    for (int i = 0; i < FSRL_MAX_CRITICS; ++i) sa.lam[i] = a.lam[i];
    return sa;
}

template <int H, int R>
__global__ __launch_bounds__(4 * H) void ppo_fwd_bwd_group_kernel(const ModelDesc md, const GroupAgent* __restrict__ tab,
                                                                 const GroupStep* __restrict__ steps,
                                                                 const PpoStepArgs base) {
    __shared__ TileSmem<H> sm;
    const GroupStep st = steps[blockIdx.y];
    if (!st.active) return;
    const GroupAgent& a = tab[blockIdx.y];
    const PpoStepArgs sa = group_step_args(base, a, st);
    ppo_fwd_bwd_body<H, R>(sm, a.P, md, a.bp, sa);
}

// the grouped form of ppo_fwd_bwd_tall_kernel (grid.y = member); the host's automatic plan makes ALL tiles tall once the group's 16-row
// tiles exceed the CU count (host_group.inc: k = 8 245 -> 283 updates/s aggregate)
template <int H>
__global__ __launch_bounds__(4 * H) void ppo_fwd_bwd_group_mix_kernel(const ModelDesc md, const GroupAgent* __restrict__ tab,
                                                                     const GroupStep* __restrict__ steps, const PpoStepArgs base,
                                                                     const int n32, const int per_net) {
    __shared__ __align__(16) unsigned char raw[sizeof(TileSmem<H, 32>)];
    const GroupStep st = steps[blockIdx.y];
    if (!st.active) return;
    const GroupAgent& a = tab[blockIdx.y];
    const PpoStepArgs sa = group_step_args(base, a, st);
    ppo_fwd_bwd_mixed<H>(raw, a.P, md, a.bp, sa, n32, per_net);
}

// r6: minibatches of up to 256 rows (BIG = false) take bursts of 4 k-steps per wave and a 64-VGPR budget, so that TWO workgroups share a
// CU: a group of 8 is 1 736 workgroups per launch (6.8 rounds of one per CU: 33.0 us; 3.4 rounds of two: 24.4 us; k = 8 232.9 -> 253.1
// updates/s aggregate, k = 4 212.6 -> 224.8, same box).  1-2 registers spill at that cap, outside the loops.  Larger (merged last)
// minibatches keep the chunked 512-row form, one workgroup per CU, nothing spilled.
template <int H, bool BIG, bool FUSE, int R>
__global__ __launch_bounds__(1024, BIG ? 4 : 8) void ppo_wgrad_group_kernel(const ModelDesc md, const GroupAgent* __restrict__ tab,
                                                              const GroupStep* __restrict__ steps,
                                                              const PpoStepArgs base) {
    const GroupStep st = steps[blockIdx.y];
    if (!st.active) return;
    const GroupAgent& a = tab[blockIdx.y];
    const PpoStepArgs sa = group_step_args(base, a, st);
    WgradPtrs wp = a.wp;
    wp.X = a.bp.obs_p + (size_t)st.mb_start * md.Do;
    const int tiles = (st.mb_size + 15) >> 4;
    ppo_wgrad_body<H, BIG, FUSE, BIG ? WG_MAXU : 4>(md, wp, tiles * 16, sa, tiles * (16 / R));
}

__global__ __launch_bounds__(ADAM_NT) void adam_clip_group_kernel(const ModelDesc md, const GroupAgent* __restrict__ tab,
                                                                 const GroupStep* __restrict__ steps,
                                                                 const PpoStepArgs base) {
    const GroupStep st = steps[blockIdx.y];
    if (!st.active) return;
    const GroupAgent& a = tab[blockIdx.y];
    const PpoStepArgs sa = group_step_args(base, a, st);
    adam_clip_body(a.Pw, a.M, a.V, a.G, a.gsq_part, a.nparts, a.n_dev, sa, a.ctrl, md);
}
