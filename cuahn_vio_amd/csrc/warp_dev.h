// warp_dev.h - device helpers of the projective bilinear warp (WarpImg.warpSingleImage_H_Mtrx, warp.py:60-79) shared by the prep / error-map kernels
// (kernels.hip) and by the block-4 kernel that samples its own input patches (conv_b4_fused.h, round 6).  One definition, so that the paths agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "geom.h"

namespace hnet {

// ---------------------------------------------------------------------------------------------
// pixel access: u8 -> float exactly as `toType(kFloat) / 255.0` (HomographyNet.cpp:141,146) through a
// 256-entry table in LDS (one IEEE division per entry per workgroup instead of one per tap)
// ---------------------------------------------------------------------------------------------
// (float)b / 255.0f without the divide or a table: q = b * (1/255) is off by at most one ulp, one Newton step on the
// remainder lands on the correctly rounded quotient — verified for all 256 bytes (test_u8_scaling_is_exact)
__device__ __forceinline__ float u8_to_unit(float f) {
    constexpr float r = 1.0f / 255.0f;
    const float q = f * r;
    return fmaf(fmaf(-255.0f, q, f), r, q);
}
template <typename PIX> struct PixRead;
template <> struct PixRead<uint8_t> {
    static constexpr bool kNeedLut = true;
    __device__ static inline float get(const uint8_t* img, int idx, const float* lut) { return lut[img[idx]]; }
    // the same value without the table (the tiled kernel's rare per-pixel fallback)
    __device__ static inline float get_direct(const uint8_t* img, int idx) { return u8_to_unit((float)img[idx]); }
    __device__ static inline float cvt(uint8_t v) { return u8_to_unit((float)v); }
};
template <> struct PixRead<float> {
    static constexpr bool kNeedLut = false;
    __device__ static inline float get(const float* img, int idx, const float*) { return img[idx]; }
    __device__ static inline float get_direct(const float* img, int idx) { return img[idx]; }
    __device__ static inline float cvt(float v) { return v; }
};

__device__ inline void fill_lut(float* lut) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) lut[i] = (float)i / 255.0f;
    __syncthreads();
}

// WarpImg.warpSingleImage_H_Mtrx (warp.py:60-79) for one output pixel (u, v):
//   (X,Y,Z) = H (u,v,1); x = X/Z, y = Y/Z; g = x * 2/(W-1) - 1; grid_sample(bilinear, zeros,
//   align_corners=True) un-normalises i = ((g+1)/2)*(W-1) and blends the 4 neighbours; taps outside the
//   image contribute 0.  fp32 throughout like the reference; fp32 division is IEEE (hipcc default).
// sampling position of output pixel (u, v) in img2, in pixels: the normalise / un-normalise round trip of
// grid_sample(align_corners=True) is kept (it is not the identity in fp32)
// X / Z and Y / Z share their denominator.  hipcc expands an IEEE fp32 division into v_div_scale x 2, v_rcp, five FMAs, v_div_fmas and
// v_div_fixup; when the operands are far from the ends of the exponent range (no scaling: the case of a homography's Z ~ 1) that is the FMA
// chain below, whose first three steps depend on Z only.  tools/div_pair_check.hip: 2^32 operand pairs, every quotient bit-identical to
// x / z.  ZSAFE says |Z| is known to lie in [2^-60, 2^60] (the tiled kernel proves it once per tile from the tile's corners: Z is affine
// in the pixel position); otherwise the test is made here and the compiler's division used outside the range.  Where the two forms
// could differ at all - quotients that are denormal or overflow - the sampled value does not depend on the quotient's low bits (the
// coordinate is then -1 after the normalisation, or the tap is outside the image).
__device__ __forceinline__ float warp_rcp_refined(float z) {
    const float r = __builtin_amdgcn_rcpf(z);
    return fmaf(fmaf(-z, r, 1.0f), r, r);
}
__device__ __forceinline__ float warp_div_with(float x, float z, float r1) {
    const float m = x * r1;
    const float f3 = fmaf(fmaf(-z, m, x), r1, m);
    return fmaf(fmaf(-z, f3, x), r1, f3);
}
__device__ __forceinline__ bool warp_z_safe(float Z) { return fabsf(Z) > 8.7e-19f && fabsf(Z) < 1.15e18f; }     // 2^-60 .. 2^60; false for NaN
template <bool ZSAFE = false>
__device__ inline void warp_coords(const float* h, int u, int v, float& ix, float& iy, float& Z) {
    const float fu = (float)u, fv = (float)v;
    const float X = fmaf(h[0], fu, fmaf(h[1], fv, h[2]));
    const float Y = fmaf(h[3], fu, fmaf(h[4], fv, h[5]));
    Z = fmaf(h[6], fu, fmaf(h[7], fv, h[8]));
    float qx, qy;
    if (ZSAFE || warp_z_safe(Z)) {
        const float r1 = warp_rcp_refined(Z);
        qx = warp_div_with(X, Z, r1);
        qy = warp_div_with(Y, Z, r1);
    } else {
        qx = X / Z;
        qy = Y / Z;
    }
    const float gx = qx * (float)(2.0 / (IMG_W - 1)) - 1.0f;
    const float gy = qy * (float)(2.0 / (IMG_H - 1)) - 1.0f;
    ix = ((gx + 1.0f) * 0.5f) * (float)(IMG_W - 1);
    iy = ((gy + 1.0f) * 0.5f) * (float)(IMG_H - 1);
}

// bilinear blend of the four taps around (ix, iy) read from global memory; out-of-image taps contribute 0
template <typename PIX, bool LUT = true>
__device__ inline float warp_taps_global(const PIX* img, float ix, float iy, const float* lut) {
    const float x0f = floorf(ix), y0f = floorf(iy);
    // NaN / far-out coordinates: all taps out of range -> 0 (comparisons with NaN are false)
    if (!(x0f >= -1.0f && x0f <= (float)IMG_W && y0f >= -1.0f && y0f <= (float)IMG_H)) return 0.0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float wx1 = ix - x0f, wx0 = 1.0f - wx1, wy1 = iy - y0f, wy0 = 1.0f - wy1;
    const bool xin0 = x0 >= 0 && x0 < IMG_W, xin1 = x0 + 1 >= 0 && x0 + 1 < IMG_W;
    const bool yin0 = y0 >= 0 && y0 < IMG_H, yin1 = y0 + 1 >= 0 && y0 + 1 < IMG_H;
    float s = 0.0f;
    if (yin0 && xin0) s = fmaf((LUT ? PixRead<PIX>::get(img, y0 * IMG_W + x0, lut) : PixRead<PIX>::get_direct(img, y0 * IMG_W + x0)), wx0 * wy0, s);
    if (yin0 && xin1) s = fmaf((LUT ? PixRead<PIX>::get(img, y0 * IMG_W + x0 + 1, lut) : PixRead<PIX>::get_direct(img, y0 * IMG_W + x0 + 1)), wx1 * wy0, s);
    if (yin1 && xin0) s = fmaf((LUT ? PixRead<PIX>::get(img, (y0 + 1) * IMG_W + x0, lut) : PixRead<PIX>::get_direct(img, (y0 + 1) * IMG_W + x0)), wx0 * wy1, s);
    if (yin1 && xin1) s = fmaf((LUT ? PixRead<PIX>::get(img, (y0 + 1) * IMG_W + x0 + 1, lut) : PixRead<PIX>::get_direct(img, (y0 + 1) * IMG_W + x0 + 1)), wx1 * wy1, s);
    return s;
}

// minimum / maximum over the four lanes of a quad (DPP quad_perm: no LDS crossbar, no wait)
template <int CTRL> __device__ __forceinline__ float quad_perm_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_min(float v) { v = fminf(v, quad_perm_f<0xB1>(v)); return fminf(v, quad_perm_f<0x4E>(v)); }
__device__ __forceinline__ float quad_max(float v) { v = fmaxf(v, quad_perm_f<0xB1>(v)); return fmaxf(v, quad_perm_f<0x4E>(v)); }
__device__ __forceinline__ float uniform_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

}  // namespace hnet
