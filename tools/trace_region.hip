// tools/trace_region.hip — what bounds the region-resident GEMM (csrc/igemm_region.h): the kernel and its ablations, timed at the batch-256 shapes.
//   for a in 0 1 2 3 4; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -DHNET_REGION_ABLATE=$a tools/trace_region.hip -o tools/trace_region_$a.bin; done
// ablations (wrong results, timing only): 1 no weight loads in the K loop, 2 one address computation per chunk, 3 no MFMAs, 4 no fragment reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../cuahn_vio_amd/csrc/igemm_region.h"
using namespace hnet;

#ifndef HNET_REGION_ABLATE
#define HNET_REGION_ABLATE 0
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class C, class K> static int run(const char* name, K kern, S3Params p) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    const int rows_tile = C::P * p.Ho * p.Wo;
    dim3 grid((p.M + rows_tile - 1) / rows_tile, p.N / C::BN, 1);
    hipEvent_t a0, a1;
    hipEventCreate(&a0); hipEventCreate(&a1);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kern, grid, dim3(C::NT), C::LDS_BYTES, 0, p);
    hipEventRecord(a0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kern, grid, dim3(C::NT), C::LDS_BYTES, 0, p);
    hipEventRecord(a1);
    CK(hipDeviceSynchronize());
    float ams; hipEventElapsedTime(&ams, a0, a1);
    std::printf("   %-28s ablation %d: grid %u x %u, %.4f ms per launch\n", name, HNET_REGION_ABLATE, grid.x, grid.y, ams / 20);
    return 0;
}

int main() {
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
    p.A = A; p.Wp = W; p.bias = bias; p.out32 = out32; p.out16 = out16; p.k_split = 1;
    {   // block_1_2: 128 -> 128, 5x5 s2, 14x20 -> 7x10, batch 256
        typedef RegionCfg<128, 5, 1, 280, false, 14, 20> C;
        p.H = 14; p.W = 20; p.Ho = 7; p.Wo = 10; p.M = 256 * 70; p.N = 128;
        p.a_plane = (size_t)256 * 14 * 20 * 128; p.o_plane = (size_t)p.M * 128;
        if (run<C>("block_1_2 70 x 128", igemm_s3_region_kernel<C, false>, p)) return 1;
    }
    {   // block_1_3: 128 -> 256, 3x3 s2, 7x10 -> 4x5
        typedef RegionCfg<128, 3, 4, 288, true, 7, 10> C;
        p.H = 7; p.W = 10; p.Ho = 4; p.Wo = 5; p.M = 256 * 20; p.N = 256;
        p.a_plane = (size_t)256 * 7 * 10 * 128; p.o_plane = (size_t)p.M * 256;
        if (run<C>("block_1_3 80 x 64, K halves", igemm_s3_region_kernel<C, false>, p)) return 1;
    }
    {   // block_2_4 / 3_5 / 4_6: 256 -> 256, 3x3 s2, 7x10 -> 4x5, fp32 out
        typedef RegionCfg<256, 3, 4, 288, true, 7, 10> C;
        p.H = 7; p.W = 10; p.Ho = 4; p.Wo = 5; p.M = 256 * 20; p.N = 256;
        p.a_plane = (size_t)256 * 7 * 10 * 256;
        if (run<C>("block_3_5 80 x 64, K halves", igemm_s3_region_kernel<C, true>, p)) return 1;
    }
    return 0;
}
