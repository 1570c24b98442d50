// Micro-benchmark: how fast can one 1024-thread workgroup per CU pull a 256 KB weight matrix
// (256x256 f32) from L2 into VGPRs, for the access patterns the tile kernels use?
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/ingest.hip -o gpurun_out/ingest ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int H = 256;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ W, float* __restrict__ out, int nets, int per_net) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
    const int net = (PAT == 6) ? blockIdx.x : (blockIdx.x / per_net) % nets;
    const float* __restrict__ P = W + (size_t)net * H * H;
    float acc = 0.f;
    if (PAT == 1 || PAT == 4) {            // forward fragment: 16 float4 per lane, rows 16w+li, 64-B pieces
        f4 v[16];
        const int rot = (PAT == 4) ? (blockIdx.x * 5 + wave) & 15 : 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int jj = (j + rot) & 15;
            v[j] = *reinterpret_cast<const f4*>(P + (size_t)(16 * wave + li) * H + 16 * jj + 4 * q);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    } else if (PAT == 2) {                 // backward fragment: 64 dword loads, 64-B pieces
        float v[64];
#pragma unroll
        for (int jc = 0; jc < 16; ++jc)
#pragma unroll
            for (int s = 0; s < 4; ++s) v[jc * 4 + s] = P[(size_t)(16 * jc + 4 * q + s) * H + 16 * wave + li];
#pragma unroll
        for (int j = 0; j < 64; ++j) acc += v[j];
    } else if (PAT == 3 || PAT == 6) {     // fully coalesced: 1 KB contiguous per wave instruction
        f4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const f4*>(P + (size_t)wave * 4096 + j * 256 + lane * 4);
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    } else if (PAT == 5) {                 // coalesced, 8 KB per wave only (half the bytes... x2 rounds)
        f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f4*>(P + (size_t)wave * 4096 + j * 256 + lane * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    } else if (PAT == 7) {                 // forward fragment with 128-B pieces: 8 lanes x 16 B contiguous per row
        f4 v[16];                          // lane -> row 8*(j&1).. : rows (lane>>3) + 8*(j&1), cols 32*(j>>1)+4*(lane&7)
#pragma unroll
        for (int j = 0; j < 16; ++j)
            v[j] = *reinterpret_cast<const f4*>(P + (size_t)(16 * wave + (lane >> 3) + 8 * (j & 1)) * H + 32 * (j >> 1) + 4 * (lane & 7));
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    }
    if (acc == 123.456f) out[blockIdx.x * 1024 + tid] = acc;
}

template <int PAT>
float run(const float* W, float* out, int grid, int nets, int per_net, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(1024), 0, 0, W, out, nets, per_net);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(1024), 0, 0, W, out, nets, per_net);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / iters;
}

int main() {
    float *W, *out;
    const int NETS = 256;
    CK(hipMalloc(&W, (size_t)NETS * H * H * 4)); CK(hipMalloc(&out, 256 * 1024 * 4));
    CK(hipMemset(W, 0, (size_t)NETS * H * H * 4));
    const int it = 500;
    for (int grid : {48, 256}) {
        const int nets = grid == 48 ? 3 : 4, per = grid / nets;
        printf("grid %3d (nets %d x %d tiles): empty %.2f | fwd-frag %.2f | fwd-frag-rot %.2f | bwd-frag %.2f | coalesced %.2f | coalesced-half %.2f | private-coalesced %.2f | fwd-128B %.2f us\n",
               grid, nets, per, run<0>(W, out, grid, nets, per, it), run<1>(W, out, grid, nets, per, it), run<4>(W, out, grid, nets, per, it),
               run<2>(W, out, grid, nets, per, it), run<3>(W, out, grid, nets, per, it), run<5>(W, out, grid, nets, per, it),
               run<6>(W, out, grid, nets, per, it), run<7>(W, out, grid, nets, per, it));
    }
    return 0;
}
