// tools/lds_probe.hip — relative cost of LDS access patterns on gfx950 (not part of the library).
//   hipcc -O3 --offload-arch=gfx950 tools/lds_probe.hip -o tools/lds_probe.bin && tools/lds_probe.bin
// Every workgroup (256 threads, 4 waves) repeats one ds_read / ds_write pattern ITER x 8 times; LDS is the only busy unit, so the
// time per access relative to the contiguous pattern is the number of LDS cycles the pattern costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITER = 4096;
typedef short s4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int W>   // W = 8: ds_read_b64, 16: ds_read_b128
__global__ __launch_bounds__(256) void probe_read(const int* __restrict__ lane_off, unsigned* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    const int off = lane_off[threadIdx.x & 63] + (threadIdx.x >> 6) * 16384;
    unsigned acc = 0;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int a = off + ((it * 8 + k) & 7) * 1024;        // stays inside the wave's 16 KB
            const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + a;
            if constexpr (W == 8) { uint2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(la) : "memory"); asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(v)); acc ^= v.x; }
            else if constexpr (W == 82) { u4 v; asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(v) : "v"(la) : "memory"); asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(v)); acc ^= v[0]; }
            else { u4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(la) : "memory"); asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(v)); acc ^= v[0]; }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int W>
__global__ __launch_bounds__(256) void probe_write(const int* __restrict__ lane_off, unsigned* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int off = lane_off[threadIdx.x & 63] + (threadIdx.x >> 6) * 16384;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int a = off + ((it * 8 + k) & 7) * 1024;
            const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + a;
            if constexpr (W == 8) { const uint2 v = make_uint2(it, k); asm volatile("ds_write_b64 %0, %1" ::"v"(la), "v"(v) : "memory"); }
            else { const u4 v = u4{(unsigned)it, (unsigned)k, 0u, 1u}; asm volatile("ds_write_b128 %0, %1" ::"v"(la), "v"(v) : "memory"); }
        }
    }
    __syncthreads();
    if (reinterpret_cast<unsigned*>(lds)[threadIdx.x] == 0x12345678u) sink[0] = 1;
}

template <class K>
static float run(K kern, const std::vector<int>& off, int* d_off, unsigned* d_sink) {
    hipMemcpy(d_off, off.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d_off, d_sink);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d_off, d_sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    int* d_off; unsigned* d_sink;
    hipMalloc(&d_off, 64 * sizeof(int)); hipMalloc(&d_sink, 16);
    struct P { const char* name; int width; bool write; std::vector<int> off; };
    std::vector<P> ps;
    auto mk = [&](const char* name, int width, bool write, auto f) {
        P p{name, width, write, std::vector<int>(64)};
        for (int l = 0; l < 64; l++) p.off[l] = f(l & 15, l >> 4, l);
        ps.push_back(p);
    };
    const int XH = 21 * 16;   // group stride of conv_patch32 (bytes)
    mk("b64 contiguous (lane*8)", 8, false, [](int, int, int l) { return l * 8; });
    mk("b64 16B stride, all low halves", 8, false, [&](int m, int g, int) { return m * 16 + g * XH; });
    mk("b64 16B stride, high half for g&1", 8, false, [&](int m, int g, int) { return m * 16 + g * XH + 8 * (g & 1); });
    mk("b64 16B stride, high half for g>>1", 8, false, [&](int m, int g, int) { return m * 16 + g * XH + 8 * (g >> 1); });
    mk("b64 16B stride, high half for (g&1)^(g>>1)", 8, false, [&](int m, int g, int) { return m * 16 + g * XH + 8 * ((g & 1) ^ (g >> 1)); });
    mk("b64 16B stride, groups 256 B apart, high half g&1", 8, false, [&](int m, int g, int) { return m * 16 + g * 256 + 8 * (g & 1); });
    mk("b64 one group only distinct (others same addr)", 8, false, [&](int m, int, int) { return m * 16; });
    mk("b128 contiguous (lane*16)", 16, false, [](int, int, int l) { return l * 16; });
    mk("b128 16B stride, groups XH apart", 16, false, [&](int m, int g, int) { return m * 16 + g * XH; });
    mk("b128 16B stride, groups 256 B apart", 16, false, [&](int m, int g, int) { return m * 16 + g * 256; });
    mk("b128 16B stride, groups 272 B apart", 16, false, [&](int m, int g, int) { return m * 16 + g * 272; });
    mk("b128 16B stride, groups 320 B apart", 16, false, [&](int m, int g, int) { return m * 16 + g * 320; });
    mk("b128 16B stride, groups 256 B apart, window start +80 B", 16, false, [&](int m, int g, int) { return m * 16 + g * 256 + 80; });
    mk("b128 16B stride, groups 512 B apart", 16, false, [&](int m, int g, int) { return m * 16 + g * 512; });
    mk("b128 16B stride, groups 128 B apart", 16, false, [&](int m, int g, int) { return m * 16 + g * 128; });
    mk("b128 16B stride, groups 64 B apart", 16, false, [&](int m, int g, int) { return m * 16 + g * 64; });
    mk("b128 32B stride (m*32 + g*16): 8-channel pixels, tap pairs", 16, false, [&](int m, int g, int) { return m * 32 + (g & 1) * 16 + (g >> 1) * 2048; });
    mk("b128 GEMM-style 128B rows swizzled (row m, chunk g^((m>>1)&7))", 16, false, [&](int m, int g, int) { return m * 128 + ((g ^ ((m >> 1) & 7)) * 16); });
    mk("b128 64B rows swizzled (row m, chunk g^((m>>2)&3))", 16, false, [&](int m, int g, int) { return m * 64 + ((g ^ ((m >> 2) & 3)) * 16); });
    mk("b128 64B rows, chunk g^(3*((m>>3)&1))", 16, false, [&](int m, int g, int) { return m * 64 + ((g ^ (((m >> 3) & 1) * 3)) * 16); });
    mk("b128 64B rows, chunk g^((m>>2)&1)*2^((m>>3)&1)", 16, false, [&](int m, int g, int) { return m * 64 + ((g ^ (((m >> 2) & 1) * 2) ^ ((m >> 3) & 1)) * 16); });
    mk("b128 64B rows unswizzled", 16, false, [&](int m, int g, int) { return m * 64 + g * 16; });
    mk("read2_b64 contiguous 16B per lane (lane*16)", 82, false, [](int, int, int l) { return l * 16; });
    mk("read2_b64 16B per lane at +8 (8-byte aligned only)", 82, false, [](int, int, int l) { return l * 16 + 8; });
    mk("read2_b64 patch: m*8 + g&1 *16, rows for g>>1 (conv_first)", 82, false, [&](int m, int g, int) { return m * 8 + (g & 1) * 16 + (g >> 1) * 344; });
    mk("b64 patch: m*8 + g&1 *16, rows for g>>1 (conv_first)", 8, false, [&](int m, int g, int) { return m * 8 + (g & 1) * 16 + (g >> 1) * 344; });
    mk("b128 UNALIGNED (8-byte) conv_first pattern m*8 + g&1*16 + rows", 16, false, [&](int m, int g, int) { return m * 8 + (g & 1) * 16 + (g >> 1) * 344; });
    mk("b128 write contiguous", 16, true, [](int, int, int l) { return l * 16; });
    mk("b128 write patch32 staging (cq fastest, XH=21)", 16, true, [&](int, int, int l) { const int cq = l & 3, pc = l >> 2; return (((pc & 1) * 4 + cq) * 21 + (pc >> 1)) * 16; });
    mk("b128 write patch32 staging (XH=22)", 16, true, [&](int, int, int l) { const int cq = l & 3, pc = l >> 2; return (((pc & 1) * 4 + cq) * 22 + (pc >> 1)) * 16; });
    mk("b64 write contiguous", 8, true, [](int, int, int l) { return l * 8; });
    mk("b64 write 16B stride (m*16 + g*8 swapped?)", 8, true, [](int m, int g, int) { return m * 32 + g * 8; });
    float base64 = 0, base128 = 0, basew = 0;
    for (auto& p : ps) {
        float ms;
        if (p.write) ms = p.width == 8 ? run(probe_write<8>, p.off, d_off, d_sink) : run(probe_write<16>, p.off, d_off, d_sink);
        else ms = p.width == 8 ? run(probe_read<8>, p.off, d_off, d_sink) : p.width == 82 ? run(probe_read<82>, p.off, d_off, d_sink) : run(probe_read<16>, p.off, d_off, d_sink);
        // accesses per CU = 4 waves x ITER x 8; LDS cycles per wave-access at 2.4 GHz (one workgroup per CU)
        const double cyc = ms * 1e-3 * 2.4e9 / (4.0 * ITER * 8);
        std::printf("%-55s %8.4f ms  %6.2f clk per wave-instruction\n", p.name, ms, cyc);
    }
    (void)base64; (void)base128; (void)basew;
    return 0;
}
