// Where do the workgroups of a 2-per-CU launch land, and in which order?  512-thread workgroups with 78 KB of LDS (the
// co-resident tile kernels' footprint): every workgroup records HW_ID / XCC_ID of its wave 0 and the shader clock at entry.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(512, 4) void probe(unsigned* out, int spin) {
    __shared__ float pad[78000 / 4];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
        out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
        const unsigned long long t = __builtin_readcyclecounter();
        out[blockIdx.x * 4 + 2] = (unsigned)t; out[blockIdx.x * 4 + 3] = (unsigned)(t >> 32);
    }
    pad[threadIdx.x] = (float)spin;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    if (pad[(threadIdx.x + 1) & 511] == 123.f) out[0] = 0;
}
int main() {
    const int G = 738;
    unsigned* d; hipMalloc(&d, G * 16);
    std::vector<unsigned> h(G * 4);
    probe<<<G, 512>>>(d, 20); hipDeviceSynchronize();
    probe<<<G, 512>>>(d, 20); hipDeviceSynchronize();
    hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < G; ++b) { unsigned long long t = ((unsigned long long)h[b * 4 + 3] << 32) | h[b * 4 + 2]; if (t < t0) t0 = t; }
    std::map<unsigned, std::vector<int>> cu;
    int slot_hist[2][16] = {};
    for (int b = 0; b < G; ++b) {
        const unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 15u;
        const unsigned wave = hw & 15u, simd = (hw >> 4) & 3u, cuid = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
        if (b < 512) { cu[(xcc << 16) | (se << 8) | (sh << 4) | cuid].push_back(b); slot_hist[b >= 256][wave]++; }
        if (b < 24 || (b >= 256 && b < 264) || b >= 512 && b < 520) {
            unsigned long long t = ((unsigned long long)h[b * 4 + 3] << 32) | h[b * 4 + 2];
            printf("block %3d: xcc %u se %u sh %u cu %2u simd %u wave-slot %u  t = +%llu\n", b, xcc, se, sh, cuid, simd, wave, t - t0);
        }
    }
    int two = 0, one = 0, other = 0;
    for (auto& kv : cu) { if (kv.second.size() == 2) ++two; else if (kv.second.size() == 1) ++one; else ++other; }
    printf("first 512 blocks: %zu distinct CUs, %d with two blocks, %d with one, %d other\n", cu.size(), two, one, other);
    printf("wave-slot of wave 0, blocks 0..255:  "); for (int s = 0; s < 10; ++s) printf("%d:%d ", s, slot_hist[0][s]); printf("\n");
    printf("wave-slot of wave 0, blocks 256..511: "); for (int s = 0; s < 10; ++s) printf("%d:%d ", s, slot_hist[1][s]); printf("\n");
    int shown = 0;
    for (auto& kv : cu) if (kv.second.size() == 2 && shown++ < 8) printf("CU %06x: blocks %d and %d\n", kv.first, kv.second[0], kv.second[1]);
    return 0;
}
