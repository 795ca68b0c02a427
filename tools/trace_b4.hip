// tools/trace_b4.hip — the fused block_4_0 + block_4_1 kernel (csrc/conv_b4_fused.h) alone, at batch 256, with compile-time variants (timing only).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DHNET_B4_P2B128=0] [-DHNET_B4_ABLATE=n] [-DB4_TH1=7] [-DB4_WARPIN=1] [-DHNET_B4_TRACE] tools/trace_b4.hip -o tools/trace_b4_x.bin
//   B4_WARPIN=1: the form that samples its own patches from u8 images (round 6), homographies = small shifts + a little perspective per pair
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
#ifndef B4_WARPIN
#define B4_WARPIN 0
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
    auto kern = block4_fused_kernel<B4_TH1, 256, NP, true, true, B4_WARPIN != 0>;
    const int lds_bytes = C::LDS_BYTES + (B4_WARPIN ? B4W_LDS_BYTES : 0);
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    B4Warp wp = {};
#if B4_WARPIN
    {
        uint8_t *i1, *i2; float* Hd;
        CK(hipMalloc(&i1, (size_t)B * 224 * 320)); CK(hipMalloc(&i2, (size_t)B * 224 * 320)); CK(hipMalloc(&Hd, B * 9 * 4));
        std::vector<uint8_t> im((size_t)B * 224 * 320);
        for (auto& v : im) { s = s * 1664525u + 1013904223u; v = (uint8_t)(s >> 24); }
        CK(hipMemcpy(i1, im.data(), im.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(i2, im.data(), im.size(), hipMemcpyHostToDevice));
        std::vector<float> Hh(B * 9);
        for (int b = 0; b < B; b++) {
            const float a = 0.01f * (float)((b % 7) - 3);
            const float hh[9] = {1.0f + a, 0.02f * (float)((b % 5) - 2), 3.0f * (float)((b % 3) - 1), -0.015f * (float)((b % 5) - 2), 1.0f - a, 2.0f * (float)((b % 4) - 1), 1e-5f * (float)((b % 3) - 1), -1e-5f, 1.0f};
            for (int i = 0; i < 9; i++) Hh[b * 9 + i] = hh[i];
        }
        CK(hipMemcpy(Hd, Hh.data(), Hh.size() * 4, hipMemcpyHostToDevice));
        wp = B4Warp{i1, i2, Hd};
    }
#endif
    const int n_tiles = B * (112 / C::TH1) * (160 / C::TW1);
    const unsigned grid = std::getenv("B4_GRID") ? std::atoi(std::getenv("B4_GRID")) : 512;      // 256: one workgroup per CU (do the two of a CU overlap?)
#ifdef HNET_B4_TRACE
    unsigned long long* tr;
    const size_t n = 8 * 4 * 9;
    CK(hipMalloc(&tr, n * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_b4_trace), &tr, sizeof(tr)));
#endif
    hipEvent_t a0, a1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, 0, x, x_plane, (const u32x4*)w0, bias, (const u32x4*)w1, bias, out, o_plane, n_tiles, 0, wp);
        CK(hipEventRecord(a0));
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, 0, x, x_plane, (const u32x4*)w0, bias, (const u32x4*)w1, bias, out, o_plane, n_tiles, 0, wp);
        CK(hipEventRecord(a1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a0, a1));
        std::printf("   TH1 %d, phase-2 b128 %d, ablation %d, warp-in %d, grid %u: LDS %d B, %d tiles: %.4f ms per launch\n", B4_TH1, (int)C::P2B128, HNET_B4_ABLATE, B4_WARPIN, grid, lds_bytes, n_tiles, ms / 10);
    }
#ifdef HNET_B4_TRACE
    {
        CK(hipMemset(tr, 0, n * 8));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, 0, x, x_plane, (const u32x4*)w0, bias, (const u32x4*)w1, bias, out, o_plane, n_tiles, 0, wp);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(n);
        CK(hipMemcpy(t.data(), tr, n * 8, hipMemcpyDeviceToHost));
        std::printf("shader clocks per tile (sums over the tiles of a workgroup, from its third tile on, / tiles)\n");
        std::printf("wg wave | loop top | wait patch+barrier | phase 1 regular | leftover+drain | barrier | phase 2 | sampling | box + issue | total per tile\n");
        for (int wg = 0; wg < 8; wg++)
            for (int w = 0; w < 4; w++) {
                const unsigned long long* a = &t[(size_t)(wg * 4 + w) * 9];
                const double cnt = (double)a[8];
                if (cnt < 1) continue;
                double tot = 0;
                for (int k = 0; k < 8; k++) tot += (double)a[k];
                std::printf("%2d %4d | %6.0f | %8.0f | %8.0f | %8.0f | %8.0f | %8.0f | %8.0f | %8.0f | %8.0f\n", wg, w, a[0] / cnt, a[1] / cnt, a[2] / cnt, a[3] / cnt, a[4] / cnt, a[5] / cnt, a[6] / cnt, a[7] / cnt, tot / cnt);
            }
    }
#endif
    return 0;
}
