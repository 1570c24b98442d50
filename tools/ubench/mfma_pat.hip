// How close to the fp32 MFMA peak do the GEMM inner loops of the full-batch tile kernels run IN ISOLATION (activations in LDS,
// weight fragments in registers, no global traffic inside the loop)?  VERDICT r4 item 1: "measure v_mfma_f32_32x32x2_f32 once".
//   p16      : kernels_fb.hpp mma_rows_n -- 1024 threads, wave = 16 columns, per 16-deep k-chunk and 16-row half one ds_read_b128 +
//              4 DEPENDENT v_mfma_f32_16x16x4_f32 (issue 32 cycles, dependent latency 40: MI355X_MICROARCH.md)
//   p16i     : the same with the two row halves' MFMAs interleaved (no back-to-back dependent pair inside a wave)
//   p16co    : kernels_fbco.hpp -- 512 threads, wave = two column groups one after the other, two workgroups per CU
//   p32      : 512 threads, wave = 32 columns x 32 rows on v_mfma_f32_32x32x2_f32 (issue 64 = dependent latency 64), ONE
//              accumulator chain, half the LDS fragment reads and MFMA issues per FLOP; 1 or 2 workgroups per CU
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int H = 256, LD = H + 4;

template <bool INTER>
__global__ __launch_bounds__(1024) void p16(float* out, int reps) {
    __shared__ float A[32 * LD + 30000];              // 153 KB: ONE workgroup per CU, like fb_hvp_mixed_kernel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
    for (int e = tid; e < 32 * LD; e += 1024) A[e] = 1e-3f * (float)(e % 97);
    f32x4 b[H / 16];
    for (int kc = 0; kc < H / 16; ++kc) b[kc] = f32x4{1e-3f * lane, 2e-3f * kc, 1e-3f * wave, 1e-3f};
    __syncthreads();
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    const float* arow = A + li * LD + 4 * q;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int kc = 0; kc < H / 16; ++kc) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + 16 * kc);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(arow + 16 * LD + 16 * kc);
            if constexpr (INTER) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b[kc][s], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b[kc][s], acc[1], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b[kc][s], acc[0], 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b[kc][s], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[(size_t)blockIdx.x * 1024 + tid] = acc[0][0] + acc[0][1] + acc[0][2] + acc[0][3] + acc[1][0] + acc[1][1] + acc[1][2] + acc[1][3];
}

__global__ __launch_bounds__(512, 4) void p16co(float* out, int reps) {
    __shared__ float A[2 * 32 * LD + 2048];            // 68.6 KB: two per CU, like fb_hvp_co_kernel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
    for (int e = tid; e < 32 * LD; e += 512) A[e] = 1e-3f * (float)(e % 97);
    __syncthreads();
    float total = 0.f;
    const float* arow = A + li * LD + 4 * q;
    for (int r = 0; r < reps; ++r) {
#pragma unroll 1
        for (int g = 0; g < 2; ++g) {
            f32x4 b[H / 16];
            for (int kc = 0; kc < H / 16; ++kc) b[kc] = f32x4{1e-3f * lane, 2e-3f * kc, 1e-3f * (wave + 8 * g), 1e-3f};
            f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
            for (int kc = 0; kc < H / 16; ++kc) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + 16 * kc);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(arow + 16 * LD + 16 * kc);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b[kc][s], acc[0], 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b[kc][s], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            total += acc[0][0] + acc[0][3] + acc[1][1] + acc[1][2];
        }
    }
    out[(size_t)blockIdx.x * 512 + tid] = total;
}

template <int LDSPAD>
__global__ __launch_bounds__(512) void p32(float* out, int reps) {
    __shared__ float A[32 * LD + LDSPAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, half = lane >> 5;
    for (int e = tid; e < 32 * LD; e += 512) A[e] = 1e-3f * (float)(e % 97);
    f32x4 b[H / 16];                                   // one K half of the wave's 32 columns: 16 chunks of 8 k (4 per lane half)
    for (int kc = 0; kc < H / 16; ++kc) b[kc] = f32x4{1e-3f * lane, 2e-3f * kc, 1e-3f * wave, 1e-3f};
    __syncthreads();
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* arow = A + i * LD + 4 * half;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {                // two K halves through the same 64 fragment registers
#pragma unroll
            for (int m = 0; m < H / 16; ++m) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 128 * kh + 8 * m);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[m][t], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += acc[e];
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

int main(int argc, char** argv) {
    float* buf; hipMalloc(&buf, 64 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 48;
    const double flop_wg = 2.0 * 32 * 256 * 256 * reps;
    auto timeit = [&](auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(a, 0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        return ms * 1e-3 / 10;
    };
    for (int blocks : {256, 512, 1024, 2048}) {
        const double t1 = timeit([&] { hipLaunchKernelGGL(p16<false>, dim3(blocks), dim3(1024), 0, 0, buf, reps); });
        const double t2 = timeit([&] { hipLaunchKernelGGL(p16<true>, dim3(blocks), dim3(1024), 0, 0, buf, reps); });
        const double t3 = timeit([&] { hipLaunchKernelGGL(p16co, dim3(blocks), dim3(512), 0, 0, buf, reps); });
        const double t4 = timeit([&] { hipLaunchKernelGGL(p32<30000>, dim3(blocks), dim3(512), 0, 0, buf, reps); });   // 150 KB: one per CU
        const double t5 = timeit([&] { hipLaunchKernelGGL(p32<10000>, dim3(blocks), dim3(512), 0, 0, buf, reps); });   // 73 KB: two per CU
        const double t6 = timeit([&] { hipLaunchKernelGGL(p32<0>, dim3(blocks), dim3(512), 0, 0, buf, reps); });       // 33 KB: four per CU (VGPRs allowing)
        auto tf = [&](double t) { return flop_wg * blocks / t / 1e12; };
        printf("%5d blocks | p16 %.1f TFLOP/s | p16 interleaved %.1f | p16co (2/CU) %.1f | p32 1/CU %.1f | p32 2/CU %.1f | p32 4/CU %.1f   (peak 157.3)\n",
               blocks, tf(t1), tf(t2), tf(t3), tf(t4), tf(t5), tf(t6));
    }
    return 0;
}
