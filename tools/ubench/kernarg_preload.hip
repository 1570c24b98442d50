// kernarg_preload.hip -- what does the cold kernel-argument fetch cost at the head of a small dependent launch?
// Two kernels doing the same thing (one dependent global load per thread, one store), 192 workgroups of 1024 threads like the PPO step's
// fused launch, in a chain of dependent launches on one stream:
//   k_struct: arguments inside a by-value struct (what libfsrl_hip's kernels take) -> every wave s_loads them from the kernarg segment
//             (cold after the kernel boundary) before it can form its first address
//   k_flat  : the same arguments as leading scalar / pointer parameters, compiled with -mllvm -amdgpu-kernarg-preload-count=16: the
//             command processor hands them over in user SGPRs, no load in front of the first address
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/ubench/kernarg_preload.hip -o tools/ubench/kernarg_preload.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Args { const float* p; float* o; int n; float s; int pad[8]; };
__global__ __launch_bounds__(1024) void k_struct(const Args a) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < a.n) a.o[i] = a.p[i] * a.s;
}
__global__ __launch_bounds__(1024) void k_flat(const float* __restrict__ p, float* __restrict__ o, int n, float s) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) o[i] = p[i] * s;
}
__global__ __launch_bounds__(1024) void k_empty() {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int nb = 192, n = nb * 1024, reps = 2000;
    float *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int round = 0; round < 3; ++round) {
        for (int which = 0; which < 3; ++which) {
            for (int w = 0; w < 2; ++w) {       // warm-up pass, then the timed one
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < reps; ++r) {
                    float* src = (r & 1) ? b : a; float* dst = (r & 1) ? a : b;      // every launch reads what the previous one wrote
                    if (which == 0) hipLaunchKernelGGL(k_empty, dim3(nb), dim3(1024), 0, s);
                    else if (which == 1) { Args g{}; g.p = src; g.o = dst; g.n = n; g.s = 1.0f; hipLaunchKernelGGL(k_struct, dim3(nb), dim3(1024), 0, s, g); }
                    else hipLaunchKernelGGL(k_flat, dim3(nb), dim3(1024), 0, s, (const float*)src, dst, n, 1.0f);
                }
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("round %d %-9s %.3f us per launch\n", round, which == 0 ? "empty" : which == 1 ? "struct" : "preload", ms * 1e3 / reps);
        }
    }
    return 0;
}
