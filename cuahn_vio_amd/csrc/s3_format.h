// s3_format.h — the "S3" number formats: an fp32 value carried as 16-bit planes for the matrix cores.
//   3 planes: three bf16 planes (exact 3-way split), six bf16 MFMAs per product                         (HNET_PREC_BF16X3)
//   1 plane:  plain bf16                                                                                 (HNET_PREC_BF16)
//   2 planes: two fp16 planes, THREE fp16 MFMAs per product                                              (HNET_PREC_F16X2)
//             activation a = A0 + A1 / 4096:  A0 = f16(a), A1 = f16((a - A0) * 4096)   (11 + 11 significand bits + sign: fp32's 24)
//             weight planes W0 = f16(4096 w), W1 = f16(4096 w - W0), W2 = W0 / 4096
//             4096 a w = W0 A0 + W1 A0 + W2 A1      (dropped: A1 W1 / 4096 <= 2^-24 |a w|; fp16 x fp16 products are exact in fp32)
//             the accumulator carries 4096 x the sum (bias enters as 4096 b) and is scaled back (exactly) in the epilogue.
//             Range: |a| < 32768 guaranteed, |w| < 16 (hnet_create checks the weights).  In [32768, 65520) the first plane is finite but 0.04 % of the
//             fp32 values (those within 2^-8 ulp of an fp16 rounding tie) scale their residual to >= 65520: the second plane becomes an infinity, the
//             result is non-finite and is DETECTED (hnet_overflow_flag / demotion), never silently wrong; from 65520 on A0 overflows as well
//             (tests/cpp/s3_format_check.cpp walks both bands exhaustively); fp16 subnormals are not flushed by the gfx950 MFMAs
//             (tools/f16x2_probe.hip), so small values only lose ABSOLUTE precision below 2^-37.
// Host + device helpers shared by the kernels (igemm_s3.h) and the weight packer (hnet_capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hnet {

__host__ __device__ inline uint16_t f32_to_bf16_rn(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(uint16_t, (__bf16)f);      // v_cvt_pk_bf16_f32: round to nearest even
#else
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);     // round to nearest even (finite inputs), same result as the device instruction
    return (uint16_t)(u >> 16);
#endif
}
__host__ __device__ inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
// exact 3-way split of an fp32 value into bf16 planes
__host__ __device__ inline void split3(float v, uint16_t& a, uint16_t& b, uint16_t& c) {
    a = f32_to_bf16_rn(v);
    const float r = v - bf16_to_f32(a);
    b = f32_to_bf16_rn(r);
    const float r2 = r - bf16_to_f32(b);
    c = f32_to_bf16_rn(r2);
}


// ---- fp16 planes (HNET_PREC_F16X2)
constexpr float S3_F16_SCALE = 4096.0f, S3_F16_INV = 1.0f / 4096.0f;
__host__ __device__ inline uint16_t f32_to_f16_rn(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(uint16_t, (_Float16)f);
#else
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7FFFFFFFu;
    if (u >= 0x7F800000u) return (uint16_t)(sign | (u > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (u >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                 // rounds to >= 65520: infinity
    if (u < 0x38800000u) {                                                   // below 2^-14: subnormal result, multiples of 2^-24
        if (u < 0x33000000u) return (uint16_t)sign;                          // < 2^-25: zero (ties at exactly 2^-25 round to even = 0)
        const int shift = 126 - (int)(u >> 23);                              // 14 .. 24
        const uint32_t mant = (u & 0x7FFFFFu) | 0x800000u;
        const uint32_t q = mant >> shift, rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
        return (uint16_t)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
    }
    const uint32_t r = u + 0xFFFu + ((u >> 13) & 1u);                        // round to nearest even on bit 13
    return (uint16_t)(sign | ((r - 0x38000000u) >> 13));
#endif
}
__host__ __device__ inline float f16_to_f32(uint16_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (float)__builtin_bit_cast(_Float16, h);
#else
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else { float f = (float)m * 5.9604644775390625e-08f; __builtin_memcpy(&u, &f, 4); u |= sign; }
    } else if (e == 31) u = sign | 0x7F800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}
// activation split: a = A0 + A1 / 4096
__host__ __device__ inline void split2h(float v, uint16_t& a, uint16_t& b) {
    a = f32_to_f16_rn(v);
    b = f32_to_f16_rn((v - f16_to_f32(a)) * S3_F16_SCALE);
}
__host__ __device__ inline float join2h(uint16_t a, uint16_t b) { return f16_to_f32(a) + f16_to_f32(b) * S3_F16_INV; }
// weight planes (see the header)
__host__ __device__ inline void wsplit2h(float w, uint16_t& w0, uint16_t& w1, uint16_t& w2) {
    const float ws = w * S3_F16_SCALE;
    w0 = f32_to_f16_rn(ws);
    w1 = f32_to_f16_rn(ws - f16_to_f32(w0));
    w2 = f32_to_f16_rn(f16_to_f32(w0) * S3_F16_INV);
}
// run-time plane count (np = 3 / 1: bf16 planes, plane 0 of the 3-way split is bf16(v); np = 2: fp16 planes, c unused)
__host__ __device__ inline void split_np(float v, int np, uint16_t& a, uint16_t& b, uint16_t& c) {
    if (np == 2) { split2h(v, a, b); c = 0; }
    else split3(v, a, b, c);
}
__host__ __device__ inline void wsplit_np(float w, int np, uint16_t& a, uint16_t& b, uint16_t& c) {
    if (np == 2) wsplit2h(w, a, b, c);
    else split3(w, a, b, c);
}
__host__ __device__ inline float join_np(uint16_t a, uint16_t b, uint16_t c, int np) {
    if (np == 2) return join2h(a, b);
    return np == 3 ? (bf16_to_f32(a) + bf16_to_f32(b)) + bf16_to_f32(c) : bf16_to_f32(a);
}

}  // namespace hnet
