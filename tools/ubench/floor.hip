// What sets the ~2-4 us "floor" of a dependent kernel in a stream on MI355X?  Empty kernels of the three
// launch shapes of the PPO step, alone and behind a kernel that leaves D MB of dirty lines in the L2s.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { char b[360]; };                      // kernarg block of the size the real kernels pass
template <int LDS>
__global__ __launch_bounds__(1024) void empty1024(Big a, float* p) { __shared__ float s[LDS / 4 + 1]; if (p == nullptr) s[threadIdx.x] = a.b[0]; }
__global__ __launch_bounds__(256) void empty256(Big a, float* p) { if (p == nullptr) p[0] = a.b[0]; }
__global__ __launch_bounds__(256) void dirty(float* p, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = i;
}
int main() {
    float* buf; hipMalloc(&buf, 64 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    Big big{};
    const int it = 400;
    auto timeit = [&](auto launch) {
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(a, 0);
        for (int i = 0; i < it; ++i) launch();
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        return ms * 1000.f / it;
    };
    printf("empty 192x1024 lds100K : %.2f us\n", timeit([&] { hipLaunchKernelGGL(empty1024<100000>, dim3(192), dim3(1024), 0, 0, big, buf); }));
    printf("empty 209x1024 lds37K  : %.2f us\n", timeit([&] { hipLaunchKernelGGL(empty1024<37000>, dim3(209), dim3(1024), 0, 0, big, buf); }));
    printf("empty 201x256          : %.2f us\n", timeit([&] { hipLaunchKernelGGL(empty256, dim3(201), dim3(256), 0, 0, big, buf); }));
    printf("empty  16x256          : %.2f us\n", timeit([&] { hipLaunchKernelGGL(empty256, dim3(16), dim3(256), 0, 0, big, buf); }));
    for (int mb : {0, 1, 3, 8}) {
        const int n = mb * 262144;
        float t_d = n ? timeit([&] { hipLaunchKernelGGL(dirty, dim3(256), dim3(256), 0, 0, buf, n); }) : 0.f;
        float t_pair = timeit([&] {
            if (n) hipLaunchKernelGGL(dirty, dim3(256), dim3(256), 0, 0, buf, n);
            hipLaunchKernelGGL(empty1024<37000>, dim3(209), dim3(1024), 0, 0, big, buf);
        });
        printf("dirty %d MB alone %.2f us ; dirty + empty 209x1024 : %.2f us  (empty adds %.2f)\n", mb, t_d, t_pair, t_pair - t_d);
    }
    return 0;
}
