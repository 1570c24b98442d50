// Does a hipGraph shorten the kernel-to-kernel boundary of a dependent chain on MI355X?  The PPO pass is 78 x 3
// dependent launches; here: chains of 234 dependent kernels of the three launch shapes (each touching a little memory so
// the dependency is real), issued (a) as stream launches, (b) as one instantiated graph captured from the same stream,
// with kernels that are (1) nearly empty and (2) about as long as the real ones (`spin` dependent loads each).
// Result (round 1): empty kernels 3.7 us -> 2.2 us per kernel in a graph, but that is the HOST's launch rate; with
// kernels of realistic length the GPU-side boundary is the same either way -- and the PPO pass replayed as graphs of 8
// steps measured 2.5 % SLOWER than plain launches (bench.py 107 vs 110 updates/s), so the pass stays on plain launches.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { char b[360]; };
template <int LDS>
__global__ __launch_bounds__(1024) void k1024(Big a, float* p, int n, int spin) {
    __shared__ float s[LDS / 4 + 1];
    int i = blockIdx.x * 1024 + threadIdx.x;
    float acc = 0.f;
    for (int k = 0; k < spin; ++k) { acc += p[(i + (int)acc) % n]; }      // dependent loads: ~1 us each
    s[threadIdx.x] = p[i % n] + acc * 1e-30f;
    __syncthreads();
    p[i % n] = s[threadIdx.x ^ 1] + a.b[0];
}
__global__ __launch_bounds__(256) void k256(Big a, float* p, int n, int spin) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (int k = 0; k < spin; ++k) { acc += p[(i + (int)acc) % n]; }
    if (i < n) p[i] = p[i] * 0.5f + a.b[0] + acc * 1e-30f;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    const int n = 1 << 18;
    float* buf; CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    Big big{};
    const int steps = 78;
  for (int spin : {0, 6}) {
    printf("-- spin %d\n", spin);
    auto chain = [&] {
        for (int s = 0; s < steps; ++s) {
            hipLaunchKernelGGL(k1024<100000>, dim3(192), dim3(1024), 0, st, big, buf, n, spin);
            hipLaunchKernelGGL(k1024<37000>, dim3(209), dim3(1024), 0, st, big, buf, n, spin);
            hipLaunchKernelGGL(k256, dim3(804), dim3(256), 0, st, big, buf, n, spin);
        }
    };
    for (int w = 0; w < 3; ++w) chain();
    CK(hipStreamSynchronize(st));
    const int reps = 10;
    CK(hipEventRecord(a, st));
    for (int r = 0; r < reps; ++r) chain();
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("stream launches : %.2f us per kernel (%.1f us per 3-kernel step)\n", ms * 1000.f / (reps * steps * 3), ms * 1000.f / (reps * steps));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    chain();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    hipEventElapsedTime(&ms, a, b);
    printf("graph launches  : %.2f us per kernel (%.1f us per 3-kernel step)\n", ms * 1000.f / (reps * steps * 3), ms * 1000.f / (reps * steps));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
    return 0;
}
