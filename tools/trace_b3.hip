// tools/trace_b3.hip — the fused block_3_0 + block_3_1 kernel (csrc/conv_b3_fused.h) alone at batch 256: time per launch and, with -DHNET_B3_TRACE, the cycles per phase.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DHNET_B3_TRACE] tools/trace_b3.hip -o tools/trace_b3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../cuahn_vio_amd/csrc/conv_b3_fused.h"
using namespace hnet;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    constexpr int B = 256;
    typedef B3Cfg C;
    const size_t n_in = (size_t)B * 112 * 160 * 2, o_plane = (size_t)B * 56 * 80 * 32;
    float *x, *bias; uint16_t *w0, *w1, *out;
    CK(hipMalloc(&x, n_in * 4)); CK(hipMalloc(&w0, C::W0_BYTES)); CK(hipMalloc(&w1, 2 * C::NSTEP1 * 2 * 64 * 16)); CK(hipMalloc(&bias, 1024)); CK(hipMalloc(&out, 2 * o_plane * 2));
    std::vector<float> h(n_in);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    CK(hipMemcpy(x, h.data(), n_in * 4, hipMemcpyHostToDevice));
    std::vector<uint16_t> hw(2 * C::NSTEP1 * 2 * 64 * 8 + C::W0_BYTES);
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x2C00 + ((s >> 16) & 0x3FF)); }
    CK(hipMemcpy(w0, hw.data(), C::W0_BYTES, hipMemcpyHostToDevice));
    CK(hipMemcpy(w1, hw.data(), (size_t)2 * C::NSTEP1 * 2 * 64 * 16, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 1024));
    auto kern = block3_fused_kernel<2>;
    const int lds = C::LDS_BYTES + C::W0_BYTES + C::SPARE_BYTES;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int n_tiles = B * C::TILES_X * C::TILES_Y;
    const unsigned grid = std::getenv("B3_GRID") ? std::atoi(std::getenv("B3_GRID")) : 512;
#ifdef HNET_B3_TRACE
    unsigned long long* tr;
    const size_t n = 8 * 4 * 5;
    CK(hipMalloc(&tr, n * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_b3_trace), &tr, sizeof(tr)));
#endif
    hipEvent_t a0, a1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, x, (const u32x4*)w0, bias, (const u32x4*)w1, bias, out, o_plane, n_tiles);
        CK(hipEventRecord(a0));
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, x, (const u32x4*)w0, bias, (const u32x4*)w1, bias, out, o_plane, n_tiles);
        CK(hipEventRecord(a1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a0, a1));
        std::printf("   block3_fused, grid %u: LDS %d B, %d tiles: %.4f ms per launch\n", grid, lds, n_tiles, ms / 10);
    }
#ifdef HNET_B3_TRACE
    {
        CK(hipMemset(tr, 0, n * 8));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, x, (const u32x4*)w0, bias, (const u32x4*)w1, bias, out, o_plane, n_tiles);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(n);
        CK(hipMemcpy(t.data(), tr, n * 8, hipMemcpyDeviceToHost));
        std::printf("shader clocks per tile (sums over the tiles of a workgroup, from its third tile on, / tiles)\n");
        std::printf("wg wave | loop top | phase 0 (patch -> LDS) | phase 1 + barrier | phase 2 | total per tile\n");
        for (int wg = 0; wg < 8; wg++)
            for (int w = 0; w < 4; w++) {
                const unsigned long long* a = &t[(size_t)(wg * 4 + w) * 5];
                const double cnt = (double)a[4];
                if (cnt < 1) continue;
                double tot = 0;
                for (int k = 0; k < 4; k++) tot += (double)a[k];
                std::printf("%2d %4d | %6.0f | %8.0f | %8.0f | %8.0f | %8.0f\n", wg, w, a[0] / cnt, a[1] / cnt, a[2] / cnt, a[3] / cnt, tot / cnt);
            }
    }
#endif
    return 0;
}
