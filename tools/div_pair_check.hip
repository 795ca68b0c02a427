// div_pair_check.hip — X / Z and Y / Z with ONE reciprocal refinement, bit for bit the IEEE quotients?
// hipcc expands an fp32 division into v_div_scale x 2, v_rcp, five FMAs, v_div_fmas, v_div_fixup (11 instructions); the warp kernels divide
// X and Y by the same Z for every pixel.  When no scaling is needed (operands far from the ends of the exponent range: Z ~ 1, |X| < 1e6
// here) the expansion reduces to the FMA chain below, whose first three steps depend on Z only.  This probe compares the shared form with
// the compiler's division on 2^32 operand pairs in the range the warp sees, plus values next to powers of two and exact ties.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/_build/div_pair_check tools/div_pair_check.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float rcp_refined(float z) {
    const float r = __builtin_amdgcn_rcpf(z);
    const float e = __builtin_fmaf(-z, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div_with(float x, float z, float r1) {
    const float m = x * r1;
    const float f2 = __builtin_fmaf(-z, m, x);
    const float f3 = __builtin_fmaf(f2, r1, m);
    const float f4 = __builtin_fmaf(-z, f3, x);
    return __builtin_fmaf(f4, r1, f3);
}
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void check(unsigned long long* bad, float* ex, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nbad = 0;
    for (uint32_t it = 0; it < 4096; it++) {
        const uint32_t h0 = mix(tid * 4096u + it + seed), h1 = mix(h0 ^ 0x9e3779b9u), h2 = mix(h1 + 0x85ebca6bu);
        // Z: a homography's third row on pixel coordinates: around 1, both signs, sometimes far from 1; X: pixel coordinates times Z
        float z = 1.0f + ((int)(h0 & 0xffffff) - 0x800000) * (1.0f / 0x800000) * ((h2 & 7) == 0 ? 0.9999f : (h2 & 7) == 1 ? 0.5f : 0.02f);
        if ((h2 & 0x30) == 0) z = -z;
        if ((h2 & 0xc0) == 0) z *= __builtin_bit_cast(float, 0x3f800000u + ((h2 >> 8) % 40 - 20) * 0x800000u);     // powers of two 2^-20 .. 2^19
        float x = ((int)(h1 & 0xfffff) - 0x80000) * (1.0f / 1024.0f) + (h1 >> 20) * (1.0f / 4096.0f);               // |x| < 513, fine grid
        if ((h2 & 0x300) == 0) x = __builtin_bit_cast(float, h1 & 0x4fffffffu);                                       // any magnitude up to 2^32
        if ((h2 & 0xc00) == 0) x = z * (float)(int)(h1 % 640);                                                         // exact quotients
        const float r1 = rcp_refined(z);
        const float q = div_with(x, z, r1), ref = x / z;
        const bool usable = __builtin_fabsf(z) > 1e-12f && __builtin_fabsf(z) < 1e12f && (x == 0.0f || (__builtin_fabsf(x) > 1e-12f && __builtin_fabsf(x) < 1e12f));
        if (usable && __builtin_bit_cast(uint32_t, q) != __builtin_bit_cast(uint32_t, ref)) {
            if (!(x == 0.0f)) { nbad++; ex[0] = x; ex[1] = z; ex[2] = q; ex[3] = ref; }
            else if (q != ref) { nbad++; ex[0] = x; ex[1] = z; ex[2] = q; ex[3] = ref; }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    unsigned long long* d_bad; float* d_ex;
    hipMalloc(&d_bad, 8); hipMalloc(&d_ex, 16);
    hipMemset(d_bad, 0, 8);
    unsigned long long total = 0;
    for (int round = 0; round < 4; round++) {
        check<<<1024, 256>>>(d_bad, d_ex, 0x1234567u * (round + 1));     // 2^18 threads x 4096 = 2^30 per round
        total += 1ull << 30;
    }
    unsigned long long bad; float ex[4];
    if (hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("device error\n"); return 1; }
    hipMemcpy(ex, d_ex, 16, hipMemcpyDeviceToHost);
    printf("%llu operand pairs, %llu quotients differ from x / z", total, bad);
    if (bad) printf("  (e.g. %a / %a: shared %a, compiler %a)", ex[0], ex[1], ex[2], ex[3]);
    printf("\n");
    return bad != 0;
}
