// Layout + rate check of v_mfma_f32_4x4x1_16B_f32 (gfx950).  Hypothesis: 16 independent blocks
// b = lane/4; A_b[i] in lane 4b+i, B_b[j] in lane 4b+j, D_b[i][j] in VGPR i of lane 4b+j.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[r * 64 + l] = acc[r];
}
// rate: 16 waves/block, each 64*ITER dependent-accumulate MFMAs; compare 4x4x1 against 16x16x4
template <int KIND>
__global__ __launch_bounds__(1024) void rate(float* out, int iters) {
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (KIND == 0) { acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, acc1, 0, 0, 0); }
            if (KIND == 1) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0); }
            if (KIND == 2) { acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc0, 0, 0, 0); acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, acc0, 0, 0, 0); }
        }
    }
    if (acc0[0] + acc1[0] == 123.f) out[threadIdx.x] = acc0[1];
}
int main() {
    float ha[64], hb[64], hd[256], *a, *b, *d;
    for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 100 + 3 * l; }
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int blk = l / 4, j = l % 4;
        const float want = ha[4 * blk + r] * hb[4 * blk + j];
        if (std::fabs(hd[r * 64 + l] - want) > 1e-3f) { if (bad < 8) printf("lane %d r %d got %g want %g\n", l, r, hd[r * 64 + l], want); ++bad; }
    }
    printf("layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"4x4x1 2-acc", "16x16x4 2-acc", "4x4x1 1-acc"};
    for (int kind = 0; kind < 3; ++kind) {
        const int iters = 200;
        auto launch = [&] {
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(1024), 0, 0, out, iters);
            if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(1024), 0, 0, out, iters);
            if (kind == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(1024), 0, 0, out, iters);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = 64.0 * iters * 16 * 256;     // per wave 64*iters, 16 waves, 256 blocks
        const double flop = n_mfma * (kind == 1 ? 2048.0 : 512.0);
        printf("%-14s %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", names[kind], ms, flop / ms * 1e-9,
               ms * 1e-3 * 2.4e9 / (64.0 * iters * 4));
    }
    return 0;
}
