// What does a device-wide barrier cost on MI355X (8 XCDs) for a 209 x 1024-thread grid?
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(1024) void k_cg(int n, float* out) {
    cg::grid_group g = cg::this_grid();
    for (int i = 0; i < n; ++i) g.sync();
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.f;
}
// hand-rolled: monotonically increasing counter, agent-scope atomics, bounded spin
__global__ __launch_bounds__(1024) void k_own(int n, unsigned* counter, float* out) {
    const unsigned nb = gridDim.x;
    for (int i = 0; i < n; ++i) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);   // agent scope by default in HIP
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = nb * (unsigned)(i + 1);
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
    }
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.f;
}
int main() {
    float* out; unsigned* ctr; hipMalloc(&out, 4); hipMalloc(&ctr, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {64, 209, 256}) {
        for (int n : {0, 1, 11, 101}) {
            float ms_cg = 0, ms_own = 0;
            for (int rep = 0; rep < 3; ++rep) {
                void* args[] = {&n, &out};
                hipEventRecord(a, 0);
                hipError_t e = hipLaunchCooperativeKernel((void*)k_cg, dim3(grid), dim3(1024), args, 0, 0);
                hipEventRecord(b, 0); hipEventSynchronize(b);
                if (e != hipSuccess) { printf("coop launch failed: %s\n", hipGetErrorString(e)); return 1; }
                hipEventElapsedTime(&ms_cg, a, b);
                hipMemset(ctr, 0, 4); hipDeviceSynchronize();
                hipEventRecord(a, 0);
                hipLaunchKernelGGL(k_own, dim3(grid), dim3(1024), 0, 0, n, ctr, out);
                hipEventRecord(b, 0); hipEventSynchronize(b);
                hipEventElapsedTime(&ms_own, a, b);
            }
            printf("grid %3d  syncs %3d : cg %.2f us   own %.2f us\n", grid, n, ms_cg * 1e3, ms_own * 1e3);
        }
    }
    return 0;
}
