// scratch: operand layout of v_mfma_f32_16x16x16_bf16 (bf16_1k builtin) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
__global__ void k(const uint16_t* A, const uint16_t* B, float* D) {   // A [16 i][16 k], B [16 k][16 j] row-major bf16
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    s4 a, b;
    for (int e = 0; e < 4; e++) { a[e] = (short)A[r * 16 + 4 * g + e]; b[e] = (short)B[(4 * g + e) * 16 + r]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    for (int e = 0; e < 4; e++) D[(4 * g + e) * 16 + r] = c[e];      // row 4g+e, col r
}
int main() {
    uint16_t hA[256], hB[256]; float fA[256], fB[256];
    for (int i = 0; i < 256; i++) { fA[i] = (float)((i * 7) % 13 - 6); fB[i] = (float)((i * 5) % 11 - 5); hA[i] = f2bf(fA[i]); hB[i] = f2bf(fB[i]); }
    uint16_t *dA, *dB; float* dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    float hD[256]; hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { float s = 0; for (int kk = 0; kk < 16; kk++) s += fA[i * 16 + kk] * fB[kk * 16 + j]; if (s != hD[i * 16 + j]) bad++; }
    printf("mismatches: %d of 256 (D[0][0] = %g)\n", bad, hD[0]);
    return 0;
}
