// Cost of a device-wide barrier INSIDE a kernel of the PPO step's shape (217 blocks x 1024 threads, one per CU), against the
// kernel boundary it would replace (adam_clip_kernel's launch floor: 4.8 us).  Variants:
//   flat : one agent-scope counter, every block arrives (release fence) and polls it, acquire fence after
//   xcd  : XCD-hierarchical (MI355X_MICROARCH.md "barrier-xcd"): per-XCD arrival counter; the LAST arriver of an XCD does the
//          release fence (one L2 write-back per XCD), arrives on the top counter, polls it, acquires, then publishes the XCD's
//          generation word; the other blocks of the XCD poll that word (same L2) and acquire
// Each barrier is preceded by `work` 16-byte stores per thread into a per-block slab (dirty lines for the release to flush);
// after it every block reads its neighbour's slab and counts stale values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Bar {
    unsigned xcd_arrive[8][32];   // one 128-byte line each
    unsigned xcd_gen[8][32];
    unsigned xcd_members[8][32];
    unsigned top[32];
    unsigned flat[32];
};
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7u; }   // HW_REG_XCC_ID
__device__ __forceinline__ unsigned ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ void barrier_flat(Bar* b, unsigned epoch, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&b->flat[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ld(&b->flat[0]) < nblocks * epoch) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
__device__ void barrier_xcd(Bar* b, unsigned epoch, unsigned xcd) {
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this block's stores have reached the XCD's L2
        const unsigned members = ld(&b->xcd_members[xcd][0]);
        const unsigned old = __hip_atomic_fetch_add(&b->xcd_arrive[xcd][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == members * epoch) {                          // the XCD's last arriver
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // one write-back of this XCD's L2
            __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld(&b->top[0]) < 8u * epoch) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(&b->xcd_gen[xcd][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld(&b->xcd_gen[xcd][0]) < epoch) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
}

template <int MODE>   // 0: no barrier, 1: flat, 2: xcd
__global__ __launch_bounds__(1024) void k(Bar* b, float4* slab, int iters, int work, unsigned* check) {
    const unsigned xcd = xcc_id();
    unsigned bad = 0;
    for (int it = 1; it <= iters; ++it) {
        for (int w = 0; w < work; ++w)
            slab[((size_t)blockIdx.x * work + w) * 1024 + threadIdx.x] = float4{(float)it, 0.f, 0.f, 0.f};
        if (MODE == 1) barrier_flat(b, (unsigned)it, gridDim.x);
        if (MODE == 2) barrier_xcd(b, (unsigned)it, xcd);
        if (MODE != 0 && work > 0) {                              // read what the NEXT block wrote before the barrier
            const unsigned nb = (blockIdx.x + 1) % gridDim.x;
            const float v = slab[((size_t)nb * work) * 1024 + threadIdx.x].x;
            if (v < (float)it) ++bad;
        }
    }
    if (bad) atomicAdd(check, bad);
}
__global__ void count_members(Bar* b) { if (threadIdx.x == 0) atomicAdd(&b->xcd_members[xcc_id()][0], 1u); }

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 217;
    Bar* b; float4* slab; unsigned* check;
    CHECK(hipMalloc(&b, sizeof(Bar))); CHECK(hipMalloc(&slab, (size_t)blocks * 4 * 1024 * 16)); CHECK(hipMalloc(&check, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 200;
    for (int work : {0, 1, 4}) {
        float us[3];
        unsigned bad_total = 0;
        for (int mode = 0; mode < 3; ++mode) {
            CHECK(hipMemset(b, 0, sizeof(Bar))); CHECK(hipMemset(check, 0, 4));
            hipLaunchKernelGGL(count_members, dim3(blocks), dim3(1024), 0, 0, b);     // same placement as the timed grid (block b -> XCD b % 8)
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 0, 0, b, slab, iters, work, check);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 0, 0, b, slab, iters, work, check);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 0, 0, b, slab, iters, work, check);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            us[mode] = ms * 1000.f / iters;
            unsigned bad; CHECK(hipMemcpy(&bad, check, 4, hipMemcpyDeviceToHost)); bad_total += bad;
        }
        printf("%d blocks x 1024 thr, %2d KB stored per block before each barrier: no barrier %.2f us/iter | flat %.2f (+%.2f) | xcd %.2f (+%.2f) | stale reads %u\n",
               blocks, work * 16, us[0], us[1], us[1] - us[0], us[2], us[2] - us[0], bad_total);
    }
    return 0;
}
