// How fast does the chip dispatch workgroups?  Empty kernels of the grouped weight-gradient kernel's shapes: B blocks of T
// threads with L bytes of LDS, back to back on one stream.  (A grouped PPO step launches 217 x members 1024-thread blocks.)
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { char b[360]; };
template <int T, int LDS>
__global__ __launch_bounds__(T) void empty(Big a, float* p) { __shared__ float s[LDS / 4 + 1]; if (p == nullptr) s[threadIdx.x] = a.b[0]; }
int main() {
    float* buf; hipMalloc(&buf, 1 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    Big big{};
    const int it = 300;
    auto timeit = [&](auto launch) {
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(a, 0);
        for (int i = 0; i < it; ++i) launch();
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        return ms * 1000.f / it;
    };
    for (int blocks : {217, 434, 868, 1736, 3472}) {
        printf("%5d blocks: 1024 thr lds37K %.2f us | 1024 thr lds1K %.2f | 512 thr lds37K %.2f | 256 thr lds37K %.2f | 256 thr lds1K %.2f\n", blocks,
               timeit([&] { hipLaunchKernelGGL((empty<1024, 37000>), dim3(blocks), dim3(1024), 0, 0, big, buf); }),
               timeit([&] { hipLaunchKernelGGL((empty<1024, 1000>), dim3(blocks), dim3(1024), 0, 0, big, buf); }),
               timeit([&] { hipLaunchKernelGGL((empty<512, 37000>), dim3(blocks), dim3(512), 0, 0, big, buf); }),
               timeit([&] { hipLaunchKernelGGL((empty<256, 37000>), dim3(blocks), dim3(256), 0, 0, big, buf); }),
               timeit([&] { hipLaunchKernelGGL((empty<256, 1000>), dim3(blocks), dim3(256), 0, 0, big, buf); }));
    }
    return 0;
}
