// tools/trace_s3.hip — phase timeline of the split-bf16 GEMM K loop (s_memtime stamps, see S3T() in igemm_s3.h).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DHNET_S3_TRACE tools/trace_s3.hip -o tools/trace_s3.bin
// Runs the heads GEMM shape (M 8192, N 512, K 5120) or a conv shape and prints, per traced workgroup and wave, the
// average cycles spent in each phase of a K-tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../cuahn_vio_amd/csrc/igemm_s3.h"
using namespace hnet;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class K> static int run(const char* name, K kern, S3Params p, dim3 grid, int nstamp) {
    unsigned long long* tr;
    const size_t n = 8 * 4 * S3T_SLOTS;
    CK(hipMalloc(&tr, n * 8));
    CK(hipMemset(tr, 0, n * 8));
    p.trace = tr;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, p);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, p);
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(n);
    CK(hipMemcpy(h.data(), tr, n * 8, hipMemcpyDeviceToHost));
    std::printf("== %s: grid %u x %u, %.4f ms per launch\n", name, grid.x, grid.y, ms / 10);
    std::printf("blk wave |   load-issue      mfma   barrier1  wait+store  barrier2 |   total/K-tile  (cycles of s_memtime, avg over K-tiles 4..)\n");
    for (int b = 0; b < 8; b++)
        for (int w = 0; w < 4; w++) {
            const unsigned long long* t = &h[(size_t)(b * 4 + w) * S3T_SLOTS];
            double ph[6] = {0, 0, 0, 0, 0, 0};
            int cnt = 0;
            for (int it = 4; (it + 1) * nstamp < S3T_SLOTS && t[(it + 1) * nstamp]; it++) {
                for (int k = 0; k < nstamp - 1; k++) ph[k] += (double)(t[it * nstamp + k + 1] - t[it * nstamp + k]);
                ph[nstamp - 1] += (double)(t[(it + 1) * nstamp] - t[it * nstamp]);
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
    const int which = argc > 1 ? std::atoi(argv[1]) : 0;
    // generic buffers: 64 MiB of pseudo-random bf16 for A, W; the numbers do not matter, the access pattern does
    const size_t NA = (size_t)64 << 20;
    uint16_t *A, *W; float *bias, *out32; uint16_t* out16; uint8_t* mask;
    CK(hipMalloc(&A, NA * 2)); CK(hipMalloc(&W, NA * 2)); CK(hipMalloc(&bias, 4096)); CK(hipMalloc(&out32, (size_t)64 << 20));
    CK(hipMalloc(&out16, (size_t)128 << 20)); CK(hipMalloc(&mask, (size_t)32 << 20));
    std::vector<uint16_t> h(NA);
    uint32_t s = 12345;
    for (size_t i = 0; i < NA; i++) { s = s * 1664525u + 1013904223u; h[i] = (uint16_t)(0x3C00 + ((s >> 16) & 0x1FF)); }
    CK(hipMemcpy(A, h.data(), NA * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), NA * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 4096));
    CK(hipMemset(mask, 0xFF, (size_t)32 << 20));
    S3Params p = {};
    p.A = A; p.Wp = W; p.bias = bias; p.out32 = out32; p.out16 = out16; p.k_split = 1; p.mask = mask; p.zeros = A;
    if (which == 0) {        // heads: M = 256 pairs x 32 samples, N = 512, K = 5120
        p.a_plane = (size_t)256 * 5120; p.w_plane = (size_t)512 * 5120; p.M = 8192; p.N = 512; p.Kp = argc > 3 ? std::atoi(argv[3]) : 5120; p.n_local = 32;
        return run("heads 128x64 BK64", igemm_s3_kernel<HeadLoaderS3, 128, 64, 2, true, 1, 64>, p, dim3(64, 8), 6);
    }
    if (which == 1) {        // block_2_2: 64 -> 128, 5x5 s2, 28x40 -> 14x20, batch 256
        typedef ConvLoaderS3<64, 5, 2, 32> L;
        p.H = 28; p.W = 40; p.Ho = 14; p.Wo = 20; p.M = 256 * 280; p.N = 128; p.Kp = L::KP;
        p.a_plane = (size_t)256 * 28 * 40 * 64; p.w_plane = (size_t)128 * L::KP; p.o_plane = (size_t)p.M * 128;
        return run("block_2_2 64x64 BK64", igemm_s3_kernel<L, 64, 64, 2, false, 1, 64>, p, dim3(p.M / 64, 2), 6);
    }
    if (which == 2) {        // block_3_4: 128 -> 256, 3x3 s2, 14x20 -> 7x10
        typedef ConvLoaderS3<128, 3, 2, 32> L;
        p.H = 14; p.W = 20; p.Ho = 7; p.Wo = 10; p.M = 256 * 70; p.N = 256; p.Kp = L::KP;
        p.a_plane = (size_t)256 * 14 * 20 * 128; p.w_plane = (size_t)256 * L::KP; p.o_plane = (size_t)p.M * 256;
        return run("block_3_4 64x64 BK64", igemm_s3_kernel<L, 64, 64, 2, false, 1, 64>, p, dim3(p.M / 64, 4), 6);
    }
    return 0;
}
