// f16x2_probe.hip — can an fp32 product be carried on the fp16 matrix cores with THREE MFMAs instead of the six of the split-bf16 format?
//   activation a = A0 + A1 / 4096   (A0 = f16(a), A1 = f16((a - A0) * 4096): 11 + 11 significand bits, the residual scaled into the normal range)
//   weight planes W0 = f16(4096 w), W1 = f16(4096 w - W0), W0' = W0 / 4096
//   4096 * a * w ~= W0 A0 + W1 A0 + W0' A1        (dropped: A1 W1 / 4096 <= 2^-22 |a w| 4096)
// Checks on the device: (1) the three builtins exist for gfx950, (2) fp16 subnormal inputs are not flushed by the MFMA, (3) the error of the
// scheme against double, next to the six-product split-bf16 scheme and a serial fp32 FMA chain, on conv-like data.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/f16x2_probe tools/f16x2_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int K = 3200, STEPS = K / 32;

// A: [16 rows][K] fp32, B: [16 cols][K] fp32 -> D[16][16] (row-major) for the three schemes
__global__ void probe(const float* A, const float* B, float* Dh, float* Db, float* Df, float* misc) {
    const int lane = threadIdx.x, m = lane & 15, g = lane >> 4;
    f4 acch = {0, 0, 0, 0}, accb = {0, 0, 0, 0};
    for (int st = 0; st < STEPS; st++) {
        h8 a0, a1, w0, w1, w0p;
        b8 ab[3], wb[3];
        for (int j = 0; j < 8; j++) {
            const int k = st * 32 + 8 * g + j;
            const float a = A[m * K + k], w = B[m * K + k];
            a0[j] = (_Float16)a;
            a1[j] = (_Float16)((a - (float)a0[j]) * 4096.f);
            w0[j] = (_Float16)(w * 4096.f);
            w1[j] = (_Float16)((w * 4096.f - (float)w0[j]));
            w0p[j] = (_Float16)((float)w0[j] * (1.f / 4096.f));
            float r = a;
            for (int p = 0; p < 3; p++) { ab[p][j] = (__bf16)r; r -= (float)ab[p][j]; }
            r = w;
            for (int p = 0; p < 3; p++) { wb[p][j] = (__bf16)r; r -= (float)wb[p][j]; }
        }
        // weights as the A operand (transposed tile, as in the library): D row 4g + r = weight row, column m = activation row
        acch = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0p, a1, acch, 0, 0, 0);
        acch = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, a0, acch, 0, 0, 0);
        acch = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, a0, acch, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], ab[2], accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[2], ab[0], accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[1], ab[1], accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], ab[1], accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[1], ab[0], accb, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], ab[0], accb, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) {
        Dh[(4 * g + r) * 16 + m] = acch[r] * (1.f / 4096.f);
        Db[(4 * g + r) * 16 + m] = accb[r];
    }
    // serial fp32 FMA chain
    for (int q = 0; q < 4; q++) {
        const int wr = 4 * g + q;
        float s = 0.f;
        for (int k = 0; k < K; k++) s = fmaf(B[wr * K + k], A[m * K + k], s);
        Df[wr * 16 + m] = s;
    }
    // subnormal inputs: 2^-20 (fp16 subnormal) x 2^10, K = 32 ones -> 32 * 2^-10 per element if not flushed
    {
        h8 x, y;
        for (int j = 0; j < 8; j++) { x[j] = (_Float16)9.5367431640625e-07f; y[j] = (_Float16)1024.f; }
        f4 z = {0, 0, 0, 0};
        z = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, z, 0, 0, 0);
        h4 x4 = {x[0], x[1], x[2], x[3]}, y4 = {y[0], y[1], y[2], y[3]};
        f4 z4 = {0, 0, 0, 0};
        z4 = __builtin_amdgcn_mfma_f32_16x16x16f16(x4, y4, z4, 0, 0, 0);
        f16v z32 = {0};
        z32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, z32, 0, 0, 0);
        if (lane == 0) { misc[0] = z[0]; misc[1] = z4[0]; misc[2] = z32[0]; }
    }
}

int main() {
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> A(16 * K), B(16 * K);
    for (int i = 0; i < 16 * K; i++) {
        float a = nd(rng);
        a = a > 0 ? a : 0.1f * a;
        const int row = i / K;
        if (row >= 12) a *= 1e-4f;                 // rows of tiny activations
        if (row == 11) a *= 3000.f;               // a row of large ones
        A[i] = a;
        float w = 0.05f * nd(rng);
        if (row == 15) w *= 1e-3f;
        if (row == 14) w *= 50.f;                  // |w| up to ~12: inside the |w| < 16 range of the 4096 w planes
        B[i] = w;
    }
    float *dA, *dB, *dh, *db, *df, *dm;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4);
    hipMalloc(&dh, 1024); hipMalloc(&db, 1024); hipMalloc(&df, 1024); hipMalloc(&dm, 64);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dh, db, df, dm);
    std::vector<float> H(256), Bf(256), F(256), M(3);
    hipMemcpy(H.data(), dh, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(Bf.data(), db, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(F.data(), df, 1024, hipMemcpyDeviceToHost);
    if (hipMemcpy(M.data(), dm, 12, hipMemcpyDeviceToHost) != hipSuccess) { printf("device error\n"); return 1; }
    printf("subnormal fp16 input 2^-20 x 2^10, expected 32 * 2^-10 = %.6g (16-deep: %.6g): 16x16x32 %.6g  16x16x16 %.6g  32x32x16 %.6g\n",
           32.0 / 1024, 16.0 / 1024, M[0], M[1], M[2]);
    // error relative to sum |w a| (the scale rounding errors live on), per scheme
    double worst[3] = {0, 0, 0}, rms[3] = {0, 0, 0};
    for (int wr = 0; wr < 16; wr++)
        for (int m = 0; m < 16; m++) {
            double ref = 0, scale = 0;
            for (int k = 0; k < K; k++) { ref += (double)B[wr * K + k] * A[m * K + k]; scale += std::fabs((double)B[wr * K + k] * A[m * K + k]); }
            const double e[3] = {std::fabs(H[wr * 16 + m] - ref) / scale, std::fabs(Bf[wr * 16 + m] - ref) / scale, std::fabs(F[wr * 16 + m] - ref) / scale};
            for (int s = 0; s < 3; s++) { worst[s] = std::max(worst[s], e[s]); rms[s] += e[s] * e[s]; }
        }
    int nonfinite = 0;
    for (int i = 0; i < 256; i++) nonfinite += !std::isfinite(H[i]);
    printf("non-finite fp16x2 results: %d of 256\n", nonfinite);
    const char* name[3] = {"fp16x2, 3 MFMAs", "bf16x3, 6 MFMAs", "fp32 FMA chain"};
    for (int s = 0; s < 3; s++) printf("%-16s |err| / sum|w a|: worst %.3e  rms %.3e\n", name[s], worst[s], std::sqrt(rms[s] / 256));
    return 0;
}
