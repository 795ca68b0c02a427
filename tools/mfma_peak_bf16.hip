// sustained v_mfma_f32_32x32x16_bf16 / 16x16x32 rate on random operands (what a bf16x6 split-fp32 GEMM would draw on)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k32(const bf16x8* in, float* out, int iters) {
    bf16x8 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F> void run(const char* name, F launch, double flop, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < reps; i++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-28s %.3f ms  %.1f TFLOP/s\n", name, ms, flop / (ms * 1e-3) / 1e12);
}
int main() {
    bf16x8* in; float* out; hipMalloc(&in, 512 * 16); hipMalloc(&out, 256 * 4096 * 4);
    std::vector<unsigned short> h(512 * 8);
    for (size_t i = 0; i < h.size(); i++) { float f = (float)((i * 2654435761u) % 2000) / 1000.f - 1.f; unsigned u; memcpy(&u, &f, 4); h[i] = u >> 16; }
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int iters : {2000, 20000, 100000}) {
        for (int blocks : {256, 512, 1024}) {
            printf("iters=%d blocks=%d\n", iters, blocks);
            run("32x32x16 bf16 1 acc", [&] { hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters); }, 32768.0 * iters * 4 * blocks, 5);
            run("32x32x16 bf16 2 acc", [&] { hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(256), 0, 0, in, out, iters); }, 32768.0 * 2 * iters * 4 * blocks, 5);
        }
    }
    return 0;
}
