// tools/trace_b42.hip — the fused block_4_2 + block_4_3 kernel (csrc/conv_b42_fused.h) alone at batch 256: time per launch and, with -DHNET_B42_TRACE, a phase timeline.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DHNET_B42_TRACE] tools/trace_b42.hip -o tools/trace_b42.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../cuahn_vio_amd/csrc/conv_b42_fused.h"
using namespace hnet;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    constexpr int B = 256;
    typedef B42Cfg C;
    const size_t i_plane = (size_t)B * B42_IMG * 16, o_plane = (size_t)B * 28 * 40 * 64;
    uint16_t *x, *w2, *w3, *out; float* bias;
    CK(hipMalloc(&x, 2 * i_plane * 2)); CK(hipMalloc(&w2, 2 * 5 * 2 * 64 * 16)); CK(hipMalloc(&w3, 4 * 9 * 2 * 64 * 16)); CK(hipMalloc(&bias, 1024)); CK(hipMalloc(&out, 2 * o_plane * 2));
    std::vector<uint16_t> h(2 * i_plane);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3000 + ((s >> 16) & 0x3FF)); }
    CK(hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(w2, h.data(), 2 * 5 * 2 * 64 * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(w3, h.data(), 4 * 9 * 2 * 64 * 16, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 1024));
    auto kern = block42_fused_kernel<2>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    const int n_tiles = B * C::TILES_X * C::TILES_Y;
    const unsigned grid = std::getenv("B42_GRID") ? std::atoi(std::getenv("B42_GRID")) : 512;
#ifdef HNET_B42_TRACE
    unsigned long long* tr;
    const size_t n = 8 * 4 * 6;
    CK(hipMalloc(&tr, n * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_b42_trace), &tr, sizeof(tr)));
#endif
    hipEvent_t a0, a1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS_BYTES, 0, x, i_plane, (const u32x4*)w2, bias, (const u32x4*)w3, bias, out, o_plane, n_tiles);
        CK(hipEventRecord(a0));
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS_BYTES, 0, x, i_plane, (const u32x4*)w2, bias, (const u32x4*)w3, bias, out, o_plane, n_tiles);
        CK(hipEventRecord(a1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a0, a1));
        std::printf("   block42_fused, grid %u: LDS %d B, %d tiles: %.4f ms per launch\n", grid, C::LDS_BYTES, n_tiles, ms / 10);
    }
#ifdef HNET_B42_TRACE
    {
        CK(hipMemset(tr, 0, n * 8));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS_BYTES, 0, x, i_plane, (const u32x4*)w2, bias, (const u32x4*)w3, bias, out, o_plane, n_tiles);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(n);
        CK(hipMemcpy(t.data(), tr, n * 8, hipMemcpyDeviceToHost));
        std::printf("shader clocks per tile (sums over the tiles of a workgroup, from its third tile on, / tiles)\n");
        std::printf("wg wave | loop top | wait patch+barrier | phase 1 | barrier + DMA issue | phase 2 | total per tile\n");
        for (int wg = 0; wg < 8; wg++)
            for (int w = 0; w < 4; w++) {
                const unsigned long long* a = &t[(size_t)(wg * 4 + w) * 6];
                const double cnt = (double)a[5];
                if (cnt < 1) continue;
                double tot = 0;
                for (int k = 0; k < 5; k++) tot += (double)a[k];
                std::printf("%2d %4d | %6.0f | %8.0f | %8.0f | %8.0f | %8.0f | %8.0f\n", wg, w, a[0] / cnt, a[1] / cnt, a[2] / cnt, a[3] / cnt, a[4] / cnt, tot / cnt);
            }
    }
#endif
    return 0;
}
