// tools/traffic_calib.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes the library's kernels use
// (MI355X_MICROARCH.md §HBM: FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads; other widths "uncalibrated: calibrate on
// a known byte count in your own access pattern").  Every kernel moves exactly BYTES bytes once (buffers far larger than the 256 MiB
// Infinity Cache, touched for the first time by that kernel):
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -- tools/traffic_calib.bin
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/w -- tools/traffic_calib.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr size_t BYTES = (size_t)1 << 30;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void read_b128(const u4* __restrict__ in, unsigned* __restrict__ sink) {          // global_load_dwordx4
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < BYTES / 16; i += (size_t)gridDim.x * 256) { const u4 v = in[i]; acc ^= v[0] ^ v[3]; }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void read_b64(const uint2* __restrict__ in, unsigned* __restrict__ sink) {       // global_load_dwordx2
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < BYTES / 8; i += (size_t)gridDim.x * 256) { const uint2 v = in[i]; acc ^= v.x ^ v.y; }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void read_lds_dma(const unsigned char* __restrict__ in, unsigned* __restrict__ sink) {   // global_load_lds_dwordx4
    __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
    const int wave = threadIdx.x >> 6;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < BYTES / 16; i += (size_t)gridDim.x * 256)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(in + i * 16),
                                         (void __attribute__((address_space(3)))*)(lds + wave * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (reinterpret_cast<unsigned*>(lds)[threadIdx.x] == 0x12345u) sink[0] = 1;
}
__global__ __launch_bounds__(256) void read_u8x4(const uint32_t* __restrict__ in, unsigned* __restrict__ sink) {   // 4 bytes per lane (u8 images)
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < BYTES / 4; i += (size_t)gridDim.x * 256) acc ^= in[i];
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void write_b128(u4* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < BYTES / 16; i += (size_t)gridDim.x * 256) out[i] = u4{(unsigned)i, 1u, 2u, 3u};
}
__global__ __launch_bounds__(256) void write_b64(uint2* __restrict__ out) {            // 8 bytes per lane, 512-byte runs per wave (fused kernel / patch kernels)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < BYTES / 8; i += (size_t)gridDim.x * 256) out[i] = make_uint2((unsigned)i, 7u);
}
__global__ __launch_bounds__(256) void write_b64_strided(uint2* __restrict__ out) {    // 8 bytes per lane, 32-byte pieces at 128-byte stride (conv_patch32: a quarter of each line per wave)
    // lane (m = lane & 15, g = lane >> 4) of wave w writes bytes [pixel * 128 + w * 32 + g * 8, +8): the four waves of a workgroup complete the lines
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    for (size_t t = blockIdx.x; t < BYTES / (16 * 128); t += gridDim.x)
        out[(t * 16 + m) * 16 + w * 4 + g] = make_uint2((unsigned)t, 9u);
}
int main() {
    unsigned char* a; unsigned* sink;
    hipMalloc(&a, BYTES * 8); hipMalloc(&sink, 64);
    hipMemset(a, 1, BYTES * 8);
    hipDeviceSynchronize();
    const dim3 g(2048), b(256);
    hipLaunchKernelGGL(read_b128, g, b, 0, 0, (const u4*)(a + 0 * BYTES), sink);
    hipLaunchKernelGGL(read_b64, g, b, 0, 0, (const uint2*)(a + 1 * BYTES), sink);
    hipLaunchKernelGGL(read_lds_dma, g, b, 0, 0, (const unsigned char*)(a + 2 * BYTES), sink);
    hipLaunchKernelGGL(read_u8x4, g, b, 0, 0, (const uint32_t*)(a + 3 * BYTES), sink);
    hipLaunchKernelGGL(write_b128, g, b, 0, 0, (u4*)(a + 4 * BYTES));
    hipLaunchKernelGGL(write_b64, g, b, 0, 0, (uint2*)(a + 5 * BYTES));
    hipLaunchKernelGGL(write_b64_strided, g, b, 0, 0, (uint2*)(a + 6 * BYTES));
    hipDeviceSynchronize();
    std::printf("each kernel moved %zu bytes (%.1f KiB)\n", BYTES, BYTES / 1024.0);
    return 0;
}
