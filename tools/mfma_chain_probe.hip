// tools/mfma_chain_probe.hip — cycles per MFMA of a dependent accumulation chain against independent chains (one wave per SIMD, operands in registers).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_chain_probe.hip -o tools/mfma_chain_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int SHAPE>      // SHAPE 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x4 c4[NACC];
    f32x16 c16[NACC];
    for (int k = 0; k < NACC; k++) { c4[k] = f32x4{0, 0, 0, 0}; for (int i = 0; i < 16; i++) c16[k][i] = 0.f; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 24 / NACC; r++)
#pragma unroll
            for (int k = 0; k < NACC; k++) {
                if constexpr (SHAPE == 0) c4[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4[k], 0, 0, 0);
                else c16[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c16[k], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int k = 0; k < NACC; k++) { s += c4[k][0] + c4[k][3]; s += c16[k][0] + c16[k][15]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int SHAPE> static void run(const char* name, int blocks) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((probe<NACC, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((probe<NACC, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    std::printf("%-12s %d accumulators, %4d workgroups (4 waves each): %.1f cycles per MFMA and wave\n", name, NACC, blocks, (double)h[0] / (iters * 24.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {1, 256, 512}) {
        run<1, 0>("16x16x32", blocks); run<2, 0>("16x16x32", blocks); run<4, 0>("16x16x32", blocks);
        run<1, 1>("32x32x16", blocks); run<2, 1>("32x32x16", blocks); run<4, 1>("32x32x16", blocks);
    }
    return 0;
}
