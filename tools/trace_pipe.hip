// tools/trace_pipe.hip — phase timeline of the pipelined LDS-DMA GEMM (csrc/igemm_pipe.h; s_memtime stamps, PIPE_T()).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DHNET_S3_TRACE tools/trace_pipe.hip -o tools/trace_pipe.bin
// Prints, per traced workgroup and wave, the average cycles of each phase of a K-tile: MFMA group 0 (+ fragment reads) | waitcnt |
// barrier | DMA issue | MFMA group 1 (+ fragment reads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../cuahn_vio_amd/csrc/igemm_pipe.h"
using namespace hnet;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class C, class K> static int run(const char* name, K kern, S3Params p, dim3 grid) {
    constexpr int nstamp = 5;
    unsigned long long* tr;
    const size_t n = 8 * 8 * S3T_SLOTS;
    CK(hipMalloc(&tr, n * 8));
    CK(hipMemset(tr, 0, n * 8));
    p.trace = tr;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // untraced timing of the kernel and of its ablations (p.tile: 91 no DMA in the loop, 92 no MFMAs, 93 no fragment reads; wrong results)
#ifndef HNET_PIPE_ABLATE
#define HNET_PIPE_ABLATE 0
#endif
    for (int kmul : {1, 2, 4}) {   // build with -DHNET_PIPE_ABLATE=91 (no DMA in the loop) / 92 (no MFMAs) / 93 (no fragment reads): wrong results, timing only
        const int abl = HNET_PIPE_ABLATE;
        S3Params q = p;
        q.trace = nullptr;
        q.Kp = p.Kp * kmul;                 // (timing only: the K loop runs kmul times as long over the same taps; slope = time per K-tile)
        hipEvent_t a0, a1;
        hipEventCreate(&a0); hipEventCreate(&a1);
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kern, grid, dim3(C::NT), C::LDS_BYTES, 0, q);
        hipEventRecord(a0);
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kern, grid, dim3(C::NT), C::LDS_BYTES, 0, q);
        hipEventRecord(a1);
        CK(hipDeviceSynchronize());
        float ams; hipEventElapsedTime(&ams, a0, a1);
        std::printf("   %s, ablation %d, K x %d (%d K-tiles): %.4f ms per launch (untraced)\n", name, abl, kmul, q.Kp / 64, ams / 20);
    }
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, grid, dim3(C::NT), C::LDS_BYTES, 0, p);
    CK(hipMemset(tr, 0, n * 8));
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(kern, grid, dim3(C::NT), C::LDS_BYTES, 0, p);
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(n);
    CK(hipMemcpy(h.data(), tr, n * 8, hipMemcpyDeviceToHost));
    std::printf("== %s: grid %u x %u, %.4f ms per launch\n", name, grid.x, grid.y, ms / 10);
    std::printf("blk wave |  mfma0+reads    waitcnt    barrier  dma-issue mfma1+reads |  total/K-tile  (s_memtime cycles, avg over K-tiles 2..)\n");
    for (int b = 0; b < 8; b++)
        for (int w = 0; w < C::NWAVE; w++) {
            const unsigned long long* t = &h[(size_t)(b * 8 + w) * S3T_SLOTS];
            double ph[nstamp + 1] = {};
            int cnt = 0;
            for (int it = 2; (it + 1) * nstamp < S3T_SLOTS && t[(it + 1) * nstamp]; it++) {
                for (int k = 0; k < nstamp - 1; k++) ph[k] += (double)(t[it * nstamp + k + 1] - t[it * nstamp + k]);
                ph[nstamp - 1] += (double)(t[(it + 1) * nstamp] - t[it * nstamp + nstamp - 1]);
                ph[nstamp] += (double)(t[(it + 1) * nstamp] - t[it * nstamp]);
                cnt++;
            }
            if (!cnt) continue;
            std::printf("%3d %4d | %10.0f %10.0f %10.0f %10.0f %10.0f | %10.0f   (%d tiles)\n", b, w, ph[0] / cnt, ph[1] / cnt, ph[2] / cnt,
                        ph[3] / cnt, ph[4] / cnt, ph[5] / cnt, cnt);
        }
    hipFree(tr);
    return 0;
}

int main(int argc, char** argv) {
    const int which = argc > 1 ? std::atoi(argv[1]) : 2;
    const size_t NA = (size_t)64 << 20;
    uint16_t *A, *W; float *bias, *out32; uint16_t* out16;
    CK(hipMalloc(&A, NA * 2)); CK(hipMalloc(&W, NA * 2)); CK(hipMalloc(&bias, 4096)); CK(hipMalloc(&out32, (size_t)64 << 20));
    CK(hipMalloc(&out16, (size_t)128 << 20));
    std::vector<uint16_t> h(NA);
    uint32_t s = 12345;
    for (size_t i = 0; i < NA; i++) { s = s * 1664525u + 1013904223u; h[i] = (uint16_t)(0x3C00 + ((s >> 16) & 0x1FF)); }
    CK(hipMemcpy(A, h.data(), NA * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), NA * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 4096));
    S3Params p = {};
    p.A = A; p.Wp = W; p.bias = bias; p.out32 = out32; p.out16 = out16; p.k_split = 1; p.zeros = A;
    if (which == 1) {        // block_2_2: 64 -> 128, 5x5 s2, 28x40 -> 14x20, batch 256
        typedef ConvLoaderS3<64, 5, 2, 32> L; typedef PipeCfg<5, 2, 2, 4, 140> C;
        p.H = 28; p.W = 40; p.Ho = 14; p.Wo = 20; p.M = 256 * 280; p.N = 128; p.Kp = L::KP;
        p.a_plane = (size_t)256 * 28 * 40 * 64; p.w_plane = (size_t)128 * L::KP; p.o_plane = (size_t)p.M * 128;
        return run<C>("block_2_2 160(140)x128", igemm_s3_pipe_kernel<L, C, false>, p, dim3(p.M / 140, 1));
    }
    if (which == 2) {        // block_3_4: 128 -> 256, 3x3 s2, 14x20 -> 7x10
        typedef ConvLoaderS3<128, 3, 2, 32> L; typedef PipeCfg<5, 2, 2, 4, 140> C;
        p.H = 14; p.W = 20; p.Ho = 7; p.Wo = 10; p.M = 256 * 70; p.N = 256; p.Kp = L::KP;
        p.a_plane = (size_t)256 * 14 * 20 * 128; p.w_plane = (size_t)256 * L::KP; p.o_plane = (size_t)p.M * 256;
        return run<C>("block_3_4 160(140)x128", igemm_s3_pipe_kernel<L, C, false>, p, dim3(p.M / 140, 2));
    }
    if (which == 3) {        // block_3_5: 256 -> 256, 3x3 s2, 7x10 -> 4x5 (fp32 output)
        typedef ConvLoaderS3<256, 3, 2, 32> L; typedef PipeCfg<5, 1, 1, 4, 80> C;
        p.H = 7; p.W = 10; p.Ho = 4; p.Wo = 5; p.M = 256 * 20; p.N = 256; p.Kp = L::KP;
        p.a_plane = (size_t)256 * 7 * 10 * 256; p.w_plane = (size_t)256 * L::KP;
        return run<C>("block_3_5 80x64", igemm_s3_pipe_kernel<L, C, true>, p, dim3(p.M / 80, 4));
    }
    return 0;
}
