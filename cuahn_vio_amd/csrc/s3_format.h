// s3_format.h — the "S3" number format: an fp32 value carried as three bf16 planes (exact 3-way split).
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

}  // namespace hnet
