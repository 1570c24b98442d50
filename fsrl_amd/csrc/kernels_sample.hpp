// kernels_sample.hpp -- the replay agents' device-side sampler (library RNG mode) and row gather as device functions, shared by
// the stand-alone launches (kernels_sac.hpp), the actors' forward launch that draws its own rows (sac_actor_tile_kernel), and the
// rider blocks of the actor's weight-gradient launch that draw the NEXT update's rows (fb_wgrad_kernel, kernels_fb.hpp).
// Reference: tianshou ReplayBufferManager.sample_indices / next / unfinished_index as used by fsrl/policy/base_policy.py:453-512.
#pragma once
#include "kernels_misc.hpp"

// ---- replay gather: rows `idx` (sampled) and `term` (n-step terminal) of the store
struct SacGatherArgs {
    StorePtrs st;
    const int* idx; const int* term;
    float* XQ;      // [B][Do+Da] concat(obs, act)               (critic update)
    float* OBS;     // [B][Do]
    float* OBSN;    // [B][Do]   obs_next at the terminal index
    float* XN;      // [B][Do+Da] obs part of concat(obs_next_T, a')
    float* XP;      // [B][Do+Da] obs part of concat(obs, a_pi)
    int B, Do, Da;
};
// ---- device-side sampling (library RNG mode): uniform rows of the store, their n-step chains
//      (tianshou ReplayBuffer.next / unfinished_index) and the two rsample N(0,1) blocks.
//      Philox4x32-10 counter RNG: counter = (row, draw, update lo, update hi), key = seed.
struct SacBook { int size, index, last_index, pad; };     // one sub-buffer's bookkeeping
struct SacSampleArgs {
    const SacBook* book; const uint8_t* flags;
    int* idx; int* chain; uint8_t* endbits; float* eps_t; float* eps_p;
    int env_num, sub_size, B, n_step, Da;
    unsigned long long stored, key, counter;
    float* eps_k; int K;      // CVPO: the K particles' N(0,1) block [K][B][Da] (NULL otherwise)
};
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);      // (0, 1]
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);              // [0, 1)
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    n0 = rad * cs; n1 = rad * sn;
}
// the two rsample noise values of row b, dimensions d0 and d0 + 1: 4 normals per Philox block, 2 for each stream
__device__ __forceinline__ void sac_sample_noise(const SacSampleArgs& a, const int b, const int d0, const uint32_t k0, const uint32_t k1) {
    uint32_t r[4] = {(uint32_t)b, 1u + (uint32_t)(d0 >> 1), (uint32_t)a.counter, (uint32_t)(a.counter >> 32)};
    philox4x32_10(r, k0, k1);
    float t0, t1, p0, p1;
    box_muller(r[0], r[1], t0, t1);
    box_muller(r[2], r[3], p0, p1);
    a.eps_t[(size_t)b * a.Da + d0] = t0; a.eps_p[(size_t)b * a.Da + d0] = p0;
    if (d0 + 1 < a.Da) { a.eps_t[(size_t)b * a.Da + d0 + 1] = t1; a.eps_p[(size_t)b * a.Da + d0 + 1] = p1; }
}
// index and n-step chain of sampled row b -> (index, terminal index); the chain's LAST end bit is left to the caller as (flag byte,
// at-the-write-head) so that the load behind it need not have landed when the caller moves on (sac_sample_end_last)
__device__ __forceinline__ void sac_sample_index(const SacSampleArgs& a, const SacBook* __restrict__ book, const int b,
                                                 const uint32_t k0, const uint32_t k1, int& idx, int& term,
                                                 unsigned char& f_last, bool& head_last) {
    uint32_t c[4] = {(uint32_t)b, 0u, (uint32_t)a.counter, (uint32_t)(a.counter >> 32)};
    philox4x32_10(c, k0, k1);
    unsigned long long k = ((unsigned long long)c[0] * a.stored) >> 32;    // uniform over the stored rows
    int e = 0;
    while (e < a.env_num - 1 && k >= (unsigned long long)book[e].size) { k -= book[e].size; ++e; }
    int cur = e * a.sub_size + (int)k;
    a.idx[b] = cur;
    idx = cur;
    unsigned char f = a.flags[cur];
    bool head = false;
    for (int n = 0; n < a.n_step; ++n) {
        const int env = cur / a.sub_size, local = cur - env * a.sub_size;
        const SacBook bk = book[env];
        if (n > 0) {                                   // indices[n] = buffer.next(indices[n-1])
            const bool end = f != 0 || local == bk.last_index;
            if (!end && bk.size > 0) { cur = env * a.sub_size + (local + 1) % bk.size; f = a.flags[cur]; }
        }
        const int loc2 = cur - env * a.sub_size;
        a.chain[(size_t)n * a.B + b] = cur;
        head = bk.size > 0 && loc2 == (bk.index - 1 + bk.size) % bk.size;
        if (n + 1 < a.n_step) a.endbits[(size_t)n * a.B + b] = (f != 0 || head) ? 1 : 0;
    }
    term = cur;                                        // the chain's last element
    f_last = f; head_last = head;
}
__device__ __forceinline__ void sac_sample_end_last(const SacSampleArgs& a, const int b, const unsigned char f_last, const bool head_last) {
    a.endbits[(size_t)(a.n_step - 1) * a.B + b] = (f_last != 0 || head_last) ? 1 : 0;
}
// one sampled row: the uniform index, its n-step chain with the end flags, the two rsample noise rows.  -> (index, terminal index)
__device__ __forceinline__ void sac_sample_row(const SacSampleArgs& a, const SacBook* __restrict__ book, const int b,
                                               const uint32_t k0, const uint32_t k1, int* idx_out = nullptr, int* term_out = nullptr) {
    int idx, term;
    unsigned char f;
    bool head;
    sac_sample_index(a, book, b, k0, k1, idx, term, f, head);
    sac_sample_end_last(a, b, f, head);
    if (idx_out) *idx_out = idx;
    if (term_out) *term_out = term;
    for (int d0 = 0; d0 < a.Da; d0 += 2) sac_sample_noise(a, b, d0, k0, k1);
}

// ---- sample + gather of SG_ROWS rows by one workgroup of any size >= 192 threads: index + chain by one lane per row in wave 0, the
//      noise by one lane per (row, pair of action dimensions) from wave 1 on -- the Philox counters of sac_sample_kernel, so the same
//      sample -- then every thread gathers.
#define SG_ROWS 16
__device__ __forceinline__ void sac_sample_gather_block(const SacSampleArgs& a, const SacGatherArgs& g, const int blk) {
    constexpr int BOOK_LDS = 512;
    __shared__ SacBook book_s[BOOK_LDS];
    __shared__ int idx_s[SG_ROWS], term_s[SG_ROWS];
    const int tid = threadIdx.x, nt = blockDim.x;
    const bool in_lds = a.env_num <= BOOK_LDS;
    if (in_lds) {
        for (int e = tid; e < a.env_num; e += nt) book_s[e] = a.book[e];
        __syncthreads();
    }
    const SacBook* __restrict__ book = in_lds ? book_s : a.book;
    const int b0 = blk * SG_ROWS;
    const uint32_t k0 = (uint32_t)a.key, k1 = (uint32_t)(a.key >> 32);
    const int npair = (a.Da + 1) >> 1;
    const bool sampler = tid < SG_ROWS && b0 + tid < a.B;
    unsigned char f_last = 0;
    bool head_last = false;
    if (sampler) {
        int idx, term;
        sac_sample_index(a, book, b0 + tid, k0, k1, idx, term, f_last, head_last);
        idx_s[tid] = idx; term_s[tid] = term;
    } else if (tid >= 64 && tid < 64 + SG_ROWS * npair) {
        const int t = tid - 64, rl = t / npair, pp = t - rl * npair;
        if (b0 + rl < a.B) sac_sample_noise(a, b0 + rl, 2 * pp, k0, k1);
    }
    __syncthreads();
    if (sampler) sac_sample_end_last(a, b0 + tid, f_last, head_last);
    const int Din = g.Do + g.Da, nrow = max(0, min(SG_ROWS, a.B - b0));
    for (int e = tid; e < nrow * Din; e += nt) {
        const int rl = e / Din, f = e - rl * Din, r = b0 + rl;
        const size_t s_ = (size_t)idx_s[rl], t_ = (size_t)term_s[rl];
        const size_t o = (size_t)r * Din + f;
        if (f < g.Do) {
            const float ob = g.st.obs[s_ * g.Do + f], on = g.st.obs_next[t_ * g.Do + f];
            g.XQ[o] = ob; g.XP[o] = ob; g.XN[o] = on;
            g.OBS[(size_t)r * g.Do + f] = ob; g.OBSN[(size_t)r * g.Do + f] = on;
        } else {
            g.XQ[o] = g.st.act[s_ * g.Da + (f - g.Do)];
        }
    }
}

// Rider of a launch with idle CUs (the SAC actor's weight-gradient launch: 100 workgroups on 256 CUs): blocks x >= x0 of the grid draw
// and gather the NEXT update's batch into the set of batch arrays this update does not use.  The next update finds its rows in
// place (same store version, key, counter, batch size -- host_sac.inc) and its first launch is the actors' forward alone.
struct SgRider {
    static constexpr bool on = true;
    SacSampleArgs sa; SacGatherArgs ga;
    int x0, nx;          // rider blocks sit at blockIdx.x in [x0, x0 + nx); block (x, y, z) takes rows SG_ROWS * ((z * gy + y) * nx + x - x0)
};
struct NoRider { static constexpr bool on = false; };
