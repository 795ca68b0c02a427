// tools/lds_unaligned.hip — does ds_read_b128 accept an address that is only 8-byte aligned on gfx950 (not part of the library)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 1000 + i;
    __syncthreads();
    const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)lds + threadIdx.x * 8 + 8;   // 8-byte aligned, odd lanes' neighbours
    u4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(la) : "memory");
    for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) if (h[l * 4 + j] != 1000u + 2 + l * 2 + j) bad++;
    std::printf("lane 0: %u %u %u %u (expect 1002..1005); lane 1: %u %u %u %u (expect 1004..1007); mismatches %d\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], bad);
    return 0;
}
