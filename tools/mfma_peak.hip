// micro-benchmark: sustained v_mfma_f32_32x32x2_f32 / 16x16x4 rate on random data, to anchor roofline fractions
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k32(const float* in, float* out, int iters) {
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(const float* in, float* out, int iters) {
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) acc[i][r] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F> void run(const char* name, F launch, double flop) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 5; i++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-28s %.3f ms  %.1f TFLOP/s\n", name, ms, flop / (ms * 1e-3) / 1e12);
}
int main() {
    float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 256 * 4096 * 4);
    std::vector<float> h(1024); for (int i = 0; i < 1024; i++) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int blocks : {256, 512, 1024, 2048}) {
        printf("blocks=%d (x4 waves)\n", blocks);
        run("32x32x2 1 acc", [&] { hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters); }, 4096.0 * iters * 4 * blocks);
        run("32x32x2 2 acc", [&] { hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(256), 0, 0, in, out, iters); }, 4096.0 * 2 * iters * 4 * blocks);
        run("16x16x4 1 acc", [&] { hipLaunchKernelGGL(k16<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters); }, 2048.0 * iters * 4 * blocks);
        run("16x16x4 2 acc", [&] { hipLaunchKernelGGL(k16<2>, dim3(blocks), dim3(256), 0, 0, in, out, iters); }, 2048.0 * 2 * iters * 4 * blocks);
        run("16x16x4 4 acc", [&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, in, out, iters); }, 2048.0 * 4 * iters * 4 * blocks);
    }
    return 0;
}
