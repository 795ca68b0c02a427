// tools/trace_b4.hip — the fused block_4_0 + block_4_1 kernel (csrc/conv_b4_fused.h) alone, at batch 256, with compile-time variants (timing only).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DHNET_B4_P2B128=0] [-DHNET_B4_ABLATE=n] [-DB4_TH1=7] tools/trace_b4.hip -o tools/trace_b4_x.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../cuahn_vio_amd/csrc/conv_b4_fused.h"
using namespace hnet;
#ifndef B4_TH1
#define B4_TH1 8
#endif
#ifndef HNET_B4_ABLATE
#define HNET_B4_ABLATE 0
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    constexpr int NP = 2, B = 256;
    typedef B4Cfg<B4_TH1, 256, NP, true> C;
    const size_t x_plane = (size_t)B * B4_HP * B4_WP;          // dwords per plane
    uint32_t* x; uint16_t *w0, *w1, *out; float* bias;
    CK(hipMalloc(&x, 3 * x_plane * 4)); CK(hipMalloc(&w0, 5 * 3 * 64 * 16)); CK(hipMalloc(&w1, 7 * 3 * 64 * 16)); CK(hipMalloc(&bias, 256));
    const size_t o_plane = (size_t)B * 112 * 160 * 16;
    CK(hipMalloc(&out, 3 * o_plane * 2));
    std::vector<uint16_t> h(3 * x_plane * 2);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3400 + ((s >> 16) & 0x3FF)); }      // fp16 in [0.25, 0.5)
    CK(hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(w0, h.data(), 5 * 3 * 64 * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(w1, h.data(), 7 * 3 * 64 * 16, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 256));
    auto kern = block4_fused_kernel<B4_TH1, 256, NP, true, true>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    const int n_tiles = B * (112 / C::TH1) * (160 / C::TW1);
    const unsigned grid = std::getenv("B4_GRID") ? std::atoi(std::getenv("B4_GRID")) : 512;      // 256: one workgroup per CU (do the two of a CU overlap?)
#ifdef HNET_B4_TRACE
    unsigned long long* tr;
    const size_t n = 8 * 4 * 7;
    CK(hipMalloc(&tr, n * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_b4_trace), &tr, sizeof(tr)));
#endif
    hipEvent_t a0, a1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS_BYTES, 0, x, x_plane, (const u32x4*)w0, bias, (const u32x4*)w1, bias, out, o_plane, n_tiles, 0);
        CK(hipEventRecord(a0));
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS_BYTES, 0, x, x_plane, (const u32x4*)w0, bias, (const u32x4*)w1, bias, out, o_plane, n_tiles, 0);
        CK(hipEventRecord(a1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a0, a1));
        std::printf("   TH1 %d, phase-2 b128 %d, ablation %d, grid %u: LDS %d B, %d tiles: %.4f ms per launch\n", B4_TH1, (int)C::P2B128, HNET_B4_ABLATE, grid, C::LDS_BYTES, n_tiles, ms / 10);
    }
#ifdef HNET_B4_TRACE
    {
        CK(hipMemset(tr, 0, n * 8));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS_BYTES, 0, x, x_plane, (const u32x4*)w0, bias, (const u32x4*)w1, bias, out, o_plane, n_tiles, 0);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(n);
        CK(hipMemcpy(t.data(), tr, n * 8, hipMemcpyDeviceToHost));
        std::printf("shader clocks per tile (sums over the tiles of a workgroup, from its third tile on, / tiles)\n");
        std::printf("wg wave | loop top | wait patch+barrier | phase 1 regular | leftover+drain | barrier | phase 2 | total per tile\n");
        for (int wg = 0; wg < 8; wg++)
            for (int w = 0; w < 4; w++) {
                const unsigned long long* a = &t[(size_t)(wg * 4 + w) * 7];
                const double cnt = (double)a[6];
                if (cnt < 1) continue;
                double tot = 0;
                for (int k = 0; k < 6; k++) tot += (double)a[k];
                std::printf("%2d %4d | %6.0f | %8.0f | %8.0f | %8.0f | %8.0f | %8.0f | %8.0f\n", wg, w, a[0] / cnt, a[1] / cnt, a[2] / cnt, a[3] / cnt, a[4] / cnt, a[5] / cnt, tot / cnt);
            }
    }
#endif
    return 0;
}
