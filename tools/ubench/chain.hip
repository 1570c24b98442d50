// One launch per optimiser step -- or per PASS -- instead of three: the step's three roles (A: fused forward/backward tiles,
// B: weight gradients, C: clip + Adam) as block ranges of ONE grid, in dependency order.  A block only ever waits for blocks
// with LOWER ids; the dispatcher hands out workgroups in id order, so everything a block waits for is already resident or done
// and no wait can deadlock inside one grid.  What replaces the kernel boundary:
//   producer: all stores of data another role reads are write-through (`sc1`); s_waitcnt vmcnt(0); barrier; thread 0 adds 1 to
//             one of 8 arrival counters of its (step, role), the last arriver of a counter adds 1 to the (step, role) top word
//   consumer: thread 0 polls the top word (agent-scope load, s_sleep), then `buffer_inv sc1` (agent-scope acquire: the XCD's
//             L2 holds last step's lines of the very same buffers), barrier
// This microbenchmark has the SHAPE of the PPO step (192 / 217 / 49 blocks of 1024 threads; A re-reads 512 KB of "parameters"
// that C rewrote, B reads what A wrote, C reads what B wrote) with busy-waits in place of arithmetic, every datum stamped with
// its step so that a stale read is counted.  Modes: three launches per step | one chained launch per step | one chained launch
// for all steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NA = 192, NB = 217, NC = 49, PER = NA + NB + NC;
constexpr int P_FLOATS = NC * 4096;           // 49 x 16 KB = 784 KB of "parameters"
constexpr int A_OUT = 4096, B_OUT = 1024;     // floats per block: 16 KB of activations, 4 KB of gradient
struct Sync { unsigned sub[3][8][32]; unsigned top[3][32]; unsigned flag[3][8][32]; };      // per role: 8 arrival counters + the word consumers poll; epochs count up
struct Bufs { float* P; float* act; float* grad; unsigned* bad; Sync* sy; long long busyA, busyB, busyC; int use_wb, inv_all, naps, xcd_flags; };

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7u; }   // HW_REG_XCC_ID
__device__ __forceinline__ void st_sc1(float* p, float4 v4) {
    const f32x4 v = {v4.x, v4.y, v4.z, v4.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void busy(long long ticks) {   // wall_clock64: 100 MHz
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}
// role r of step `epoch` (1-based) is complete when top[r] == 8 * epoch
__device__ __forceinline__ void arrive(const Bufs& b, int role, int idx, int members, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (b.use_wb) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int c = idx & 7, mine = (members - c + 7) / 8;           // members of counter c
        const unsigned old = __hip_atomic_fetch_add(&b.sy->sub[role][c][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == (unsigned)mine * epoch) {
            const unsigned t = __hip_atomic_fetch_add(&b.sy->top[role][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b.xcd_flags && t + 1 == 8u * epoch)                  // the role's very last arriver publishes one word per XCD
                for (int x = 0; x < 8; ++x) __hip_atomic_store(&b.sy->flag[role][x][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ void wait_for(const Bufs& b, int role, unsigned epoch) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        const unsigned* w = b.xcd_flags ? &b.sy->flag[role][xcc_id()][0] : &b.sy->top[role][0];
        const unsigned want = b.xcd_flags ? epoch : 8u * epoch;
        while (ld_agent(w) < want) {
            for (int n = 0; n < b.naps; ++n) __builtin_amdgcn_s_sleep(1);
            if (++spins > 4000000u) { atomicAdd(b.bad, 1u << 20); break; }   // never hang the box: a lost dependency shows as 2^20 "stale reads"
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (b.inv_all) asm volatile("buffer_inv sc1" ::: "memory");
}

template <bool CHAIN>
__device__ void role_A(const Bufs& b, int idx, unsigned step) {
    if (CHAIN && step > 1) wait_for(b, 2, step - 1);
    // 512 KB of parameters per block through the L2 (what C wrote in the previous step), 2 x 16 B per thread and round
    unsigned bad = 0;
    float acc = 0.f;
    for (int r = 0; r < 32; ++r) {
        const float4 v = reinterpret_cast<const float4*>(b.P)[((r * 1024 + threadIdx.x) * 1) % (P_FLOATS / 4)];
        if (v.x != (float)(step - 1)) ++bad;
        acc += v.y;
    }
    busy(b.busyA);
    for (int w = 0; w < A_OUT / 4096; ++w)
        st_sc1(b.act + ((size_t)idx * A_OUT) + (w * 1024 + threadIdx.x) * 4, float4{(float)step, acc, 0.f, 0.f});
    if (bad) atomicAdd(b.bad, bad);
    if (CHAIN) arrive(b, 0, idx, NA, step);
}
template <bool CHAIN>
__device__ void role_B(const Bufs& b, int idx, unsigned step) {
    if (CHAIN) wait_for(b, 0, step);
    unsigned bad = 0;
    float acc = 0.f;
    for (int r = 0; r < 3; ++r) {                                       // 48 KB of other blocks' activations
        const float4 v = reinterpret_cast<const float4*>(b.act)[(size_t)((idx * 7 + r * 61) % NA) * (A_OUT / 4) + threadIdx.x];
        if (v.x != (float)step) ++bad;
        acc += v.y;
    }
    busy(b.busyB);
    if (threadIdx.x < B_OUT / 4) st_sc1(b.grad + (size_t)idx * B_OUT + threadIdx.x * 4, float4{(float)step, acc, 0.f, 0.f});
    if (bad) atomicAdd(b.bad, bad);
    if (CHAIN) arrive(b, 1, idx, NB, step);
}
template <bool CHAIN>
__device__ void role_C(const Bufs& b, int idx, unsigned step) {
    if (CHAIN) wait_for(b, 1, step);
    unsigned bad = 0;
    float acc = 0.f;
    for (int r = 0; r < 4; ++r) {                                       // the gradient slice of this block (spread over B's outputs)
        const float v = b.grad[(size_t)((idx * 4 + r) % NB) * B_OUT + (threadIdx.x & (B_OUT / 4 - 1)) * 4];
        if (v != (float)step) ++bad;
        acc += v;
    }
    busy(b.busyC);
    st_sc1(b.P + (size_t)idx * 4096 + threadIdx.x * 4, float4{(float)step, acc, 0.f, 0.f});
    if (bad) atomicAdd(b.bad, bad);
    if (CHAIN) arrive(b, 2, idx, NC, step);
}
__global__ __launch_bounds__(1024) void kA(Bufs b, unsigned step) { role_A<false>(b, blockIdx.x, step); }
__global__ __launch_bounds__(1024) void kB(Bufs b, unsigned step) { role_B<false>(b, blockIdx.x, step); }
__global__ __launch_bounds__(1024) void kC(Bufs b, unsigned step) { role_C<false>(b, blockIdx.x, step); }
// chained: grid = steps_in_launch * PER blocks; the first A of a launch does not wait (the launch boundary did)
__global__ __launch_bounds__(1024) void kchain(Bufs b, unsigned step0) {
    const unsigned s = blockIdx.x / PER, r = blockIdx.x % PER, step = step0 + s;
    if (r < NA) {
        if (s == 0) { role_A<false>(b, r, step); arrive(b, 0, r, NA, step); } else role_A<true>(b, r, step);
    } else if (r < NA + NB) role_B<true>(b, r - NA, step);
    else role_C<true>(b, r - NA - NB, step);
}
__global__ void fill(float* p, size_t n, float v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = v; }

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 78;
    Bufs b{};
    CHECK(hipMalloc(&b.P, P_FLOATS * 4)); CHECK(hipMalloc(&b.act, (size_t)NA * A_OUT * 4)); CHECK(hipMalloc(&b.grad, (size_t)NB * B_OUT * 4));
    CHECK(hipMalloc(&b.bad, 4)); CHECK(hipMalloc(&b.sy, sizeof(Sync)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const long long us = 100;                                           // wall_clock64 ticks per microsecond
    const double work[3][3] = {{0, 0, 0}, {6.0, 1.5, 0.5}, {9.0, 2.5, 1.0}};
    struct V { const char* name; int per_pass, wb, inv_all, naps, xcd; };
    const V vs[] = {{"three launches per step", -1, 0, 0, 1, 0},
                    {"chain/step: inv by every wave, nap 1", 0, 0, 1, 1, 0}, {"chain/step: inv by thread 0, nap 1", 0, 0, 0, 1, 0},
                    {"chain/step: thread 0, nap 8", 0, 0, 0, 8, 0}, {"chain/step: thread 0, nap 1, per-XCD flags", 0, 0, 0, 1, 1},
                    {"chain/step: thread 0, nap 4, per-XCD flags", 0, 0, 0, 4, 1}, {"chain/step: + wbl2 release", 0, 1, 0, 1, 1},
                    {"chain/pass: thread 0, nap 1", 1, 0, 0, 1, 0}, {"chain/pass: thread 0, nap 8", 1, 0, 0, 8, 0},
                    {"chain/pass: thread 0, nap 1, per-XCD flags", 1, 0, 0, 1, 1}, {"chain/pass: thread 0, nap 4, per-XCD flags", 1, 0, 0, 4, 1},
                    {"chain/pass: thread 0, nap 16, per-XCD flags", 1, 0, 0, 16, 1}};
    for (int wi = 0; wi < 3; ++wi) {
        b.busyA = (long long)(work[wi][0] * us); b.busyB = (long long)(work[wi][1] * us); b.busyC = (long long)(work[wi][2] * us);
        for (const V& v : vs) {
            b.use_wb = v.wb; b.inv_all = v.inv_all; b.naps = v.naps; b.xcd_flags = v.xcd;
            float best = 1e9f; unsigned bad = 0;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipMemset(b.sy, 0, sizeof(Sync))); CHECK(hipMemset(b.bad, 0, 4));
                hipLaunchKernelGGL(fill, dim3((P_FLOATS + 255) / 256), dim3(256), 0, 0, b.P, (size_t)P_FLOATS, 0.f);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0, 0));
                if (v.per_pass < 0)
                    for (int s = 1; s <= steps; ++s) {
                        hipLaunchKernelGGL(kA, dim3(NA), dim3(1024), 0, 0, b, (unsigned)s);
                        hipLaunchKernelGGL(kB, dim3(NB), dim3(1024), 0, 0, b, (unsigned)s);
                        hipLaunchKernelGGL(kC, dim3(NC), dim3(1024), 0, 0, b, (unsigned)s);
                    }
                else if (v.per_pass == 0)
                    for (int s = 1; s <= steps; ++s) hipLaunchKernelGGL(kchain, dim3(PER), dim3(1024), 0, 0, b, (unsigned)s);
                else
                    hipLaunchKernelGGL(kchain, dim3(PER * steps), dim3(1024), 0, 0, b, 1u);
                CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                unsigned bd; CHECK(hipMemcpy(&bd, b.bad, 4, hipMemcpyDeviceToHost)); bad += bd;
            }
            printf("work A/B/C = %.1f/%.1f/%.1f us | %-46s %7.2f us per step | stale reads %u\n", work[wi][0], work[wi][1], work[wi][2],
                   v.name, best * 1000.f / steps, bad);
        }
    }
    return 0;
}
