// common.hpp -- shared types for libfsrl_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fsrl_hip.h"

#define FSRL_MAX_NETS (1 + FSRL_MAX_CRITICS)
#define FSRL_TILE_M 16      // rows of one MFMA M-tile (v_mfma_f32_16x16x4_f32)
#define FSRL_DOW 32         // floats per row of the dout side buffer: [0,16) dout, [16,32) dsigma
#define FSRL_MAX_OBS 128
#define FSRL_MAX_ACT 16
#define FSRL_W1_LDS 16      // W1 is staged to LDS when obs_dim <= 16
// per-row loss inputs: act[16] | logp_old | adv_n[4] | ret[4] | v_old[4] | pad | mean_old[16] | std_old[16]
#define FSRL_RD 64
#define FSRL_RD_LOGP 16
#define FSRL_RD_ADV 17
#define FSRL_RD_RET 21
#define FSRL_RD_VOLD 25     // value_clip: the critics' process_fn-time outputs (batch.values)
#define FSRL_RD_MEAN 32
#define FSRL_RD_STD 48

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Offsets (in floats) of one network inside the flat parameter / gradient / Adam vectors.
struct NetOff {
    int sigma;  // -1 for critics
    int W1, b1, W2, b2, W3, b3;
    int W2f;    // forward-fragment mirror of W2 (same values, wave-contiguous order), beyond the main vector
    int out;    // output width of the head (Da for the actor, 1 for a V critic)
    int begin, end;  // [begin,end) slice of the flat vector owned by this net
};

struct ModelDesc {
    int Do, Da, H, n_nets;  // n_nets = 1 + n_critics ; net 0 = actor
    int unbounded;          // on-policy actor head: 0 -> mu = max_action * tanh(head); 1 -> mu = head (ActorProb unbounded=True)
    NetOff net[FSRL_MAX_NETS];
};

// ---- forward-fragment mirror of W2.  The forward GEMM wants, per wave w and k-chunk kc, lane
// (li, q) to hold W2[16w + li][16kc + 4q .. +3].  Read from the row-major matrix that is 16 rows
// x 64 B per instruction (1 KB stride): measured 5.9 us for the 256 KB burst, against 0.9 us when the
// same bytes are 1 KB-contiguous per instruction (tools/ubench/ingest.hip).  So every network keeps a
// second copy of W2 in exactly that order; whoever writes W2 (Adam, Polyak, host uploads) writes
// the mirror too.  The backward GEMM and everything else keep using the row-major matrix.
__host__ __device__ __forceinline__ int w2f_index(int H, int n, int k) {
    return ((((n >> 4) * (H >> 4) + (k >> 4)) * 64) + ((k >> 2) & 3) * 16 + (n & 15)) * 4 + (k & 3);
}
// mirror position of main-vector element i, or -1 when i is not inside a W2 tensor
__device__ __forceinline__ int w2f_mirror_of(const ModelDesc& md, int i) {
    const int HH = md.H * md.H;
    for (int net = 0; net < md.n_nets; ++net) {
        const int o = i - md.net[net].W2;
        if ((unsigned)o < (unsigned)HH) return md.net[net].W2f + w2f_index(md.H, o / md.H, o % md.H);
    }
    return -1;
}

// Scalars of one PPO minibatch step that the kernels need (passed by value).
struct PpoStepArgs {
    int mb_start;   // offset of this minibatch inside the pass permutation
    int mb_size;    // rows in this minibatch (B, or up to 2B-1 for the merged last one)
    int step;       // global step index inside the update (row of the stats table)
    int mb_index;   // index into the per-minibatch advantage statistics of this pass
    int last_in_pass;
    int first_in_pass;
    int iters_in_pass;
    int pass;       // index of the pass inside the update (for the KL early-stop gate)
    float rescale;  // 1/(sum(lambda)+1)
    float lam[FSRL_MAX_CRITICS];  // lambda_i for cost critic i (index 0 = first cost)
    float eps_clip, dual_clip, vf_coef, max_action, max_grad_norm, target_kl;
    int norm_adv, use_lagrangian;
    // Adam
    float lr, beta1, beta2, adam_eps;
    float one_minus_b1, one_minus_b2;  // (float)(1.0 - (double)beta), torch's lerp/addcmul scalars
    double kl_thresh;  // 1.5 * target_kl in float64 (python float in the reference)
    float step_size;   // lr / (1 - beta1^t)          (host float64 -> f32, like torch)
    float bc2_sqrt;    // sqrt(1 - beta2^t)
    int value_clip;    // ppo_lag.py:158-164 (only with reward_normalization, as the reference asserts)
    int fuse_adam;     // max_grad_norm off (the agent default): the weight-gradient kernel applies Adam itself, 2 launches per step
    int xcd_pair;      // fused forward/backward launch: 1 = one network per XCD pair (single-agent step, <= 4 networks), 0 = tile-major blocks
    int dbg_phase;     // probe builds only (-DFSRL_PROBES, env FSRL_DBG_PHASE): early-exit timing experiments, results invalid
};
// Early-exit timing probes of the step kernels (tools/phase_probe.sh).  They exist only in a build with -DFSRL_PROBES
// (fsrl_amd/csrc/build.sh --probes -> libfsrl_hip_probe.so); in the shipped library the conditions fold to false and
// no environment variable can change a result.
#ifdef FSRL_PROBES
#define FSRL_PROBE(sa, n) ((sa).dbg_phase == (n))
// in-kernel timeline (probe builds): lane 0 of wave 0 stores the shader clock at phase boundary k of its workgroup
#define FSRL_TS(buf, k) do { if ((buf) && threadIdx.x == 0) (buf)[(size_t)blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define FSRL_PROBE(sa, n) false
#define FSRL_TS(buf, k) do { } while (0)
#endif

// torch.optim.Adam single-tensor update of one element (lerp_ / mul_ + addcmul_ / sqrt / div / add_(eps) / addcdiv_), the
// PPO step's ONE definition: adam_clip_kernel and the fused weight-gradient kernel both inline it, with the contractions
// written out (contract(off) + explicit fmaf), so that the 3-launch and the 2-launch step round identically.
__device__ __forceinline__ void ppo_adam_math(const float g, float& m, float& v, float& p, const PpoStepArgs& sa) {
#pragma clang fp contract(off)
    m = fmaf(sa.one_minus_b1, g - m, m);
    v = v * sa.beta2;
    v = fmaf(sa.one_minus_b2 * g, g, v);
    const float denom = sqrtf(v) / sa.bc2_sqrt + sa.adam_eps;
    p = p + (-sa.step_size * m) / denom;
}

// Device-resident control block (one per context).
struct CtrlBlock {
    int stopped_after;   // passes with index > stopped_after are skipped (INT_MAX = none)
    int pad;
    double kl_sum;       // sum of per-minibatch approx-KL of the current pass (python float sum)
    float last_grad_norm;
    float pad2;
};

__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    // D[16x16] += A[16x4] * B[4x16]; lane l: a = A[l&15][l>>4], b = B[l>>4][l&15];
    // c[r] = D[4*(l>>4)+r][l&15]
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 mfma_4x4x1(float a, float b, f32x4 c) {
    // 16 independent blocks blk = l>>2:  D_blk[4x4] += A_blk[4x1] * B_blk[1x4];
    // lane l: a = A_blk[l&3], b = B_blk[l&3]; c[r] = D_blk[r][l&3]   (checked: tools/ubench/mfma4.hip)
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// 16-byte store of data the NEXT launch reads on other XCDs (activation side buffers, freshly stepped parameters): written
// through (`sc1`), so the line leaves this XCD's L2 at once instead of sitting dirty until the kernel-boundary write-back and
// the reader's first touch finds it in the Infinity Cache.  Same-box A/B on the headline bench: 112.4 -> 116.6 updates/s, the
// fused kernel 13.8 -> 12.9 us (MI355X_MICROARCH.md "publish-large"); 4-byte sc1 stores (D1, DO, gradient) added nothing.
__device__ __forceinline__ void store4_next(float* p, const f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");   // s_nop: the store-data hazard of a >8-byte VMEM store
}
// the full-batch / replay kernels' 16-byte spills (A1 / A2 / D2 and the R-op operands, 20 MB arrays at N = 20 000): the same
// write-through.  A/B: CPO 38.9 -> 38.65 ms, TRPO-Lag 32.4 -> 32.0 ms, SAC-Lag 6 450 -> 6 610 updates/s.
#ifdef FSRL_FB_PLAIN_STORES   // A/B build only (tools/ab_trust_co.py): the spills as plain stores (ack from the XCD's L2, not from the memory side)
__device__ __forceinline__ void store4_fb(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
#else
__device__ __forceinline__ void store4_fb(float* p, const f32x4 v) { store4_next(p, v); }
#endif

// e / d for e < 2^16 as a multiply-high with ceil(2^32 / d) (exact there); d == 1 would need 2^32 itself: magic 0 stands for it
__host__ __device__ __forceinline__ unsigned div_magic(const int d) { return d > 1 ? 0xFFFFFFFFu / (unsigned)d + 1u : 0u; }
__device__ __forceinline__ int div_by_magic(const unsigned e, const unsigned magic) { return magic ? (int)__umulhi(e, magic) : (int)e; }

// Global loads of the hot paths as BUFFER loads (r6): one SGPR resource per array + a 32-bit byte offset per lane
// (`buffer_load_dwordx4 v, v_off, s[rsrc], 0 offen`) instead of a 64-bit address pair per lane and load.  fb_wgrad3_kernel, same box:
// 60.4 -> 52.5 us per launch (the load side alone, MFMAs taken out: 28.4 -> 25.0 us), and the loads overlap the matrix work better.
#ifndef WG3_NO_BUFFER
typedef unsigned wg3_u32x4 __attribute__((ext_vector_type(4)));
struct Wg3Buf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ Wg3Buf wg3_buf(const float* base) {        // base: workgroup-uniform
    Wg3Buf b; b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7FFFFFFF, 0x00020000); return b;
}
// the same, built where it is used: the base goes through an opaque SGPR pair, so the four descriptor registers are not hoisted out of
// an enclosing loop and kept alive next to the other resources of a register-bound kernel (SGPR spills land in VGPR lanes)
__device__ __forceinline__ Wg3Buf wg3_buf_here(const float* base) {
    unsigned long long p = (unsigned long long)base;
    asm volatile("" : "+s"(p));
    return wg3_buf(reinterpret_cast<const float*>(p));
}
__device__ __forceinline__ f32x4 wg3_ld4(const Wg3Buf& b, const unsigned off_floats) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, off_floats * 4u, 0, 0));
}
__device__ __forceinline__ float wg3_ld1(const Wg3Buf& b, const unsigned off_floats) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, off_floats * 4u, 0, 0));
}
// the same with a wave-uniform part of the offset in an SGPR (`soffset`): bursts that walk a matrix in fixed strides
__device__ __forceinline__ float wg3_ld1s(const Wg3Buf& b, const unsigned lane_off_floats, const unsigned uni_off_floats) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, lane_off_floats * 4u, uni_off_floats * 4u, 0));
}
__device__ __forceinline__ f32x4 wg3_ld4s(const Wg3Buf& b, const unsigned lane_off_floats, const unsigned uni_off_floats) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, lane_off_floats * 4u, uni_off_floats * 4u, 0));
}
#else
struct Wg3Buf { const float* p; };
__device__ __forceinline__ Wg3Buf wg3_buf(const float* base) { Wg3Buf b; b.p = base; return b; }
__device__ __forceinline__ Wg3Buf wg3_buf_here(const float* base) { return wg3_buf(base); }
__device__ __forceinline__ f32x4 wg3_ld4(const Wg3Buf& b, const unsigned off_floats) { return *reinterpret_cast<const f32x4*>(b.p + off_floats); }
__device__ __forceinline__ float wg3_ld1(const Wg3Buf& b, const unsigned off_floats) { return b.p[off_floats]; }
__device__ __forceinline__ float wg3_ld1s(const Wg3Buf& b, const unsigned l, const unsigned u) { return b.p[l + u]; }
__device__ __forceinline__ f32x4 wg3_ld4s(const Wg3Buf& b, const unsigned l, const unsigned u) { return *reinterpret_cast<const f32x4*>(b.p + l + u); }
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
