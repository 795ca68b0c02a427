// kernels_conv.hip — instantiations and dispatch of the implicit-GEMM kernel (igemm.h) for the 20
// convolution layers of HomographyNet (reference model_to_trace.py:88-113, :210-216) and the heads' first FC.
#include "s3_dispatch.h"

namespace hnet {

// execution order; seg = K-segment of the packed weights: largest of {32,16,8} dividing KS*CIN, else 16 (padded)
const ConvDesc kConvs[20] = {
    {"block_1_1", 2, 128, 7, 2, 16, 1}, {"block_1_2", 128, 128, 5, 2, 32, 1}, {"block_1_3", 128, 256, 3, 2, 32, 1},
    {"block_2_1", 2, 64, 7, 2, 16, 2},  {"block_2_2", 64, 128, 5, 2, 32, 2},  {"block_2_3", 128, 256, 3, 2, 32, 2},
    {"block_2_4", 256, 256, 3, 2, 32, 2},
    {"block_3_0", 2, 16, 7, 1, 16, 3},  {"block_3_1", 16, 32, 5, 2, 16, 3},   {"block_3_2", 32, 64, 3, 2, 32, 3},
    {"block_3_3", 64, 128, 3, 2, 32, 3}, {"block_3_4", 128, 256, 3, 2, 32, 3}, {"block_3_5", 256, 256, 3, 2, 32, 3},
    {"block_4_0", 2, 8, 7, 1, 16, 4},   {"block_4_1", 8, 16, 5, 2, 8, 4},     {"block_4_2", 16, 32, 3, 2, 16, 4},
    {"block_4_3", 32, 64, 3, 2, 32, 4}, {"block_4_4", 64, 128, 3, 2, 32, 4},  {"block_4_5", 128, 256, 3, 2, 32, 4},
    {"block_4_6", 256, 256, 3, 2, 32, 4},
};

bool conv_is_first_direct(int layer) { return layer == 7 || layer == 13; }   // block_3_0, block_4_0: conv_first.h

int conv_padded_k(int layer) {
    const ConvDesc& d = kConvs[layer];
    if (conv_is_first_direct(layer)) return 0;
    const int rl = d.ks * d.cin;
    const int spr = (rl + d.seg - 1) / d.seg;
    return d.ks * spr * d.seg;
}

// Small-M launches (batch-1 latency: a 4x5 feature map is 20 GEMM rows) would run a handful of workgroups through a
// long serial K loop (72 K-tiles ~ 72 us).  They are cut along K into gridDim.z slices whose raw partial sums are
// combined in a fixed order by splitk_reduce_kernel: deterministic, one extra launch.
// split-K policy of the small-M launches: at least `min_iters` K-tiles per slice, about `target` workgroups in total.
// Measured (HNET_SPLITK_MIN_ITERS / HNET_SPLITK_BLOCKS sweeps): with a handful of tiles (batch 1-4) fewer, longer slices win
// (4 K-tiles, 192 workgroups: batch-1 p50 0.289 -> 0.278 ms: less partial-sum traffic for the reduce kernel), with 16+ tiles
// 3 K-tiles and 384 workgroups do (block_1_2 at batch 256: 0.109 vs 0.142 ms).
int splitk_min_iters(long tiles) { return tiles < 16 ? 4 : 3; }
int splitk_target_blocks(long tiles) { return tiles < 16 ? 192 : 384; }

template <class L, int BM, int BN, int WGM, int MF>
static hipError_t run(IgemmParams p, hipStream_t s, float* ws, size_t ws_floats) {
    dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, 1);
    const long tiles = (long)grid.x * grid.y;
    const int n_iter = (p.Kp + IG_BK - 1) / IG_BK;
    int split = 1;
    if (ws && tiles < 192 && n_iter >= 8 && (p.N % 4) == 0) {
        split = (int)std::min<long>(std::min<long>(n_iter / splitk_min_iters(tiles), (splitk_target_blocks(tiles) + tiles - 1) / tiles), 64);
        const size_t per = (size_t)p.M * p.N;
        if ((size_t)split * per > ws_floats) split = (int)(ws_floats / per);
        if (split < 2) split = 1;
    }
    p.k_split = split;
    p.partial = ws;
    grid.z = split;

    hipLaunchKernelGGL((igemm_kernel<L, BM, BN, WGM, MF>), grid, dim3(256), 0, s, p);
    if (split > 1) {
        const size_t total4 = (size_t)p.M * p.N / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, ws, split, p.M, p.N, p.bias, p.out);
    }
    return hipGetLastError();
}

// tile choice: Cout <= 16 -> 16x16x4 MFMA, BN = 16;  Cout = 32 -> BN = 32;  else BN = 64.
// BM = 128 when that still gives >= 2 workgroups per CU worth of tiles, else 64.
template <int CIN, int KS, int STRIDE, int SEG, int COUT>
static hipError_t run_conv(const IgemmParams& p, hipStream_t s, float* ws, size_t wsn) {
    typedef ConvLoader<CIN, KS, STRIDE, SEG> L;
    if constexpr (COUT <= 16) {
        return run<L, 128, 16, 4, 16>(p, s, ws, wsn);
    } else if constexpr (COUT == 32) {
        return run<L, 128, 32, 4, 32>(p, s, ws, wsn);
    } else {
        return run<L, 64, 64, 2, 32>(p, s, ws, wsn);   // 64x64: 4 workgroups/CU; measured faster than 128x64 / 128x128 on every layer
    }
}

hipError_t launch_conv(int layer, const float* in, int batch, int h, int w, const float* wpacked,
                       const float* bias, float* out, hipStream_t s, float* ws, size_t wsn, uint16_t* out16, size_t o_plane) {
    if (layer < 0 || layer >= 20) return hipErrorInvalidValue;
    const ConvDesc& d = kConvs[layer];
    IgemmParams p = {};
    p.A = in; p.Wp = wpacked; p.bias = bias; p.out = out;
    p.out16 = out16; p.o_plane = o_plane;
    if (out16) { ws = nullptr; wsn = 0; }   // the S3 epilogue is not combined with split-K (only block_1_1 / block_2_1 use it)
    p.H = h; p.W = w;
    p.Ho = conv_out_dim(h, d.ks, d.stride);
    p.Wo = conv_out_dim(w, d.ks, d.stride);
    p.M = batch * p.Ho * p.Wo;
    p.N = d.cout;
    p.Kp = conv_padded_k(layer);
    if (conv_is_first_direct(layer)) {   // wpacked = MFMA B-fragments [NFRAG][64] (pack_first_weights)
        if (d.cout == 8) {
            const int tx = (w + 63) / 64, ty = (h + 15) / 16;
            hipLaunchKernelGGL(conv7_c2_s1_kernel<8>, dim3((unsigned)(batch * tx * ty)), dim3(256), 0, s, in, wpacked, bias, out, out16, o_plane, h, w, tx, ty);
        } else {
            const int tx = (w + 31) / 32, ty = (h + 15) / 16;
            hipLaunchKernelGGL(conv7_c2_s1_kernel<16>, dim3((unsigned)(batch * tx * ty)), dim3(256), 0, s, in, wpacked, bias, out, out16, o_plane, h, w, tx, ty);
        }
        return hipGetLastError();
    }
    switch (layer) {
        case 0:  return run_conv<2, 7, 2, 16, 128>(p, s, ws, wsn);
        case 1:  return run_conv<128, 5, 2, 32, 128>(p, s, ws, wsn);
        case 2: case 5: case 11: case 18: return run_conv<128, 3, 2, 32, 256>(p, s, ws, wsn);
        case 3:  return run_conv<2, 7, 2, 16, 64>(p, s, ws, wsn);
        case 4:  return run_conv<64, 5, 2, 32, 128>(p, s, ws, wsn);
        case 6: case 12: case 19: return run_conv<256, 3, 2, 32, 256>(p, s, ws, wsn);
        case 8:  return run_conv<16, 5, 2, 16, 32>(p, s, ws, wsn);
        case 9: case 16: return run_conv<32, 3, 2, 32, 64>(p, s, ws, wsn);
        case 10: case 17: return run_conv<64, 3, 2, 32, 128>(p, s, ws, wsn);
        case 14: return run_conv<8, 5, 2, 8, 16>(p, s, ws, wsn);
        case 15: return run_conv<16, 3, 2, 16, 32>(p, s, ws, wsn);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// bf16-matrix-core path (igemm_s3.h, conv_patch_s2.h, conv_first.h, conv_b4_fused.h): s3_dispatch.h holds the variant
// selection as templates on the plane count; NP = 3 is instantiated here, NP = 1 in kernels_conv_bf16.hip
// ---------------------------------------------------------------------------------------------
HNET_S3_DISPATCH_INSTANCES(, 3)
HNET_S3_DISPATCH_INSTANCES(extern, 1)
HNET_S3_DISPATCH_INSTANCES(extern, 2)

bool conv_is_s3_layer(int layer) { return kConvs[layer].cin >= 8; }
bool conv_region_layer(int layer) { return layer == 1 || layer == 2 || layer == 6 || layer == 12 || layer == 19; }
int conv_region_taps_padded(int layer) { return layer == 1 ? 25 : 10; }      // RegionCfg::NTAP_PAD: block_1_2 all 25 taps per wave; the 3 x 3 layers 5 + 5 (K-split)
bool conv_is_patch_layer(int layer) { return layer == 8 || layer == 15; }   // block_3_1 (5x5), block_4_2 (3x3)
bool conv_is_patch32_layer(int layer) { return layer == 9 || layer == 16; }   // block_3_2, block_4_3 (3x3, 32 -> 64)

hipError_t conv_kernels_init_device() {
    hipError_t e = conv_kernels_init_device_np<3>();
    if (e == hipSuccess) e = conv_kernels_init_device_np<1>();
    return e != hipSuccess ? e : conv_kernels_init_device_np<2>();
}

// arithmetic mode of the bf16 / fp16 matrix-core layers = number of activation planes (s3_format.h)
#define HNET_NP(fn, ...) (n_planes == 1 ? fn<1>(__VA_ARGS__) : n_planes == 2 ? fn<2>(__VA_ARGS__) : fn<3>(__VA_ARGS__))

hipError_t launch_block4_fused(const void* x_in, size_t x_plane, const void* w0frag, const float* bias0, const void* w1frag, const float* bias1,
                               uint16_t* out16, size_t o_plane, int batch, hipStream_t s, int flags, int n_planes, const B4Warp* warp) {
    return HNET_NP(launch_block4_fused_np, x_in, x_plane, w0frag, bias0, w1frag, bias1, out16, o_plane, batch, s, flags, warp);
}

hipError_t launch_block42_fused(const uint16_t* in16, size_t i_plane, const void* w2frag, const float* bias2, const void* w3frag, const float* bias3,
                                uint16_t* out16, size_t o_plane, int batch, hipStream_t s, int n_planes) {
    return HNET_NP(launch_block42_fused_np, in16, i_plane, w2frag, bias2, w3frag, bias3, out16, o_plane, batch, s);
}

hipError_t launch_block3_fused(const float* x_in, const void* w0frag, const float* bias0, const void* w1frag, const float* bias1, uint16_t* out16,
                               size_t o_plane, int batch, hipStream_t s, int n_planes) {
    return HNET_NP(launch_block3_fused_np, x_in, w0frag, bias0, w1frag, bias1, out16, o_plane, batch, s);
}

hipError_t launch_conv_first_s3(const float* x_in, const void* wfrag, const float* bias, uint16_t* out16, size_t o_plane, int batch,
                                int h, int w, hipStream_t s, int n_planes) {
    return HNET_NP(launch_conv_first_s3_np, x_in, wfrag, bias, out16, o_plane, batch, h, w, s);
}

hipError_t launch_conv_first_s2(int layer, const float* x_in, const void* wfrag, const float* bias, uint16_t* out16, size_t o_plane, int batch,
                                hipStream_t s, int n_planes) {
    return HNET_NP(launch_conv_first_s2_np, layer, x_in, wfrag, bias, out16, o_plane, batch, s);
}

hipError_t launch_conv_patch(int layer, const uint16_t* in, size_t i_plane, int batch, int h, int w, const void* wfrag,
                             const float* bias, uint16_t* out16, size_t o_plane, hipStream_t s, int n_planes, bool b128, int rb5) {
    return HNET_NP(launch_conv_patch_np, layer, in, i_plane, batch, h, w, wfrag, bias, out16, o_plane, s, b128, rb5);
}

hipError_t launch_heads_fc1_s3(const float* feat, int batch, int n_local, int s_begin, float p_drop, uint64_t mc_seed,
                               uint64_t pair_seq0, const uint16_t* w1planes, const float* b1, float* hidden,
                               uint16_t* feat16, size_t f_plane, uint8_t* mask, hipStream_t s, float* ws, size_t wsn,
                               const uint64_t* seq_dev, int n_planes, int tile, LatIO* lat) {
    return HNET_NP(launch_heads_fc1_s3_np, feat, batch, n_local, s_begin, p_drop, mc_seed, pair_seq0, w1planes, b1, hidden, feat16, f_plane, mask, s, ws, wsn, seq_dev, tile, lat);
}

bool heads_fc1_one_launch(int batch, int n_local, int n_planes) { return n_planes == 2 && batch <= 8 && n_local <= 16 * HL_MAXG; }

hipError_t launch_conv_s3(int layer, const uint16_t* in, size_t in_plane, int batch, int h, int w, const uint16_t* wplanes,
                          size_t w_plane, const float* bias, uint16_t* out16, size_t o_plane, float* out32, hipStream_t s,
                          float* ws, size_t wsn, const uint16_t* wfrag, int n_planes, int tile, LatIO* lat) {
    return HNET_NP(launch_conv_s3_np, layer, in, in_plane, batch, h, w, wplanes, w_plane, bias, out16, o_plane, out32, s, ws, wsn, wfrag, tile, lat);
}

// 16-byte chunks of [np][B][h][w][c] planes to / from the interior of a bordered [np][B][hp][wp][c] array
__global__ void s3_repitch_kernel(const uint16_t* __restrict__ src, size_t src_plane, uint16_t* __restrict__ dst, size_t dst_plane, int batch, int h, int w, int cch,
                                  int hp, int wp, int pady, int padx, int to_padded, int n_planes) {
    const long n = (long)n_planes * batch * h * w * cch;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ch = (int)(i % cch);
    long r = i / cch;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h); r /= h;
    const int b = (int)(r % batch);
    const int pl = (int)(r / batch);
    const size_t plain = (size_t)pl * (to_padded ? src_plane : dst_plane) + ((((size_t)b * h + y) * w + x) * cch + ch) * 8;
    const size_t padded = (size_t)pl * (to_padded ? dst_plane : src_plane) + ((((size_t)b * hp + y + pady) * wp + x + padx) * cch + ch) * 8;
    if (to_padded) *reinterpret_cast<u32x4*>(dst + padded) = *reinterpret_cast<const u32x4*>(src + plain);
    else *reinterpret_cast<u32x4*>(dst + plain) = *reinterpret_cast<const u32x4*>(src + padded);
}
hipError_t launch_s3_repitch(const uint16_t* src, size_t src_plane, uint16_t* dst, size_t dst_plane, int batch, int h, int w, int c, int hp, int wp, int pady,
                             int padx, bool to_padded, hipStream_t s, int n_planes) {
    if (c % 8) return hipErrorInvalidValue;
    const long n = (long)n_planes * batch * h * w * (c / 8);
    hipLaunchKernelGGL(s3_repitch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, src_plane, dst, dst_plane, batch, h, w, c / 8, hp, wp, pady, padx,
                       to_padded ? 1 : 0, n_planes);
    return hipGetLastError();
}

hipError_t launch_nchw_f32_to_nhwc_s3(const float* in, uint16_t* out, size_t o_plane, int batch, int c, int h, int w, hipStream_t s, int n_planes) {
    const long n = (long)batch * c * h * w;
    hipLaunchKernelGGL(nchw_f32_to_nhwc_s3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, o_plane, batch, c, h * w, n_planes);
    return hipGetLastError();
}
hipError_t launch_nhwc_s3_to_nchw_f32(const uint16_t* in, size_t i_plane, float* out, int batch, int c, int h, int w, hipStream_t s, int n_planes) {
    const long n = (long)batch * c * h * w;
    hipLaunchKernelGGL(nhwc_s3_to_nchw_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, i_plane, out, batch, c, h * w, n_planes);
    return hipGetLastError();
}

hipError_t launch_heads_fc1(const float* feat, int batch, int n_local, int s_begin, float p_drop, uint64_t mc_seed,
                            uint64_t pair_seq0, const float* w1packed, const float* b1, float* hidden, hipStream_t s,
                            float* ws, size_t wsn, const uint64_t* seq_dev) {
    IgemmParams p = {};
    p.A = feat; p.Wp = w1packed; p.bias = b1; p.out = hidden;
    p.M = batch * n_local;
    p.N = 512;
    p.Kp = 5120;
    p.n_local = n_local;
    p.s_begin = s_begin;
    p.thr = hnet_drop_threshold(p_drop);
    p.scale = 1.0f / (1.0f - p_drop);
    p.mc_seed = mc_seed;
    p.pair_seq0 = pair_seq0;
    p.seq_dev = seq_dev;
    return run<HeadLoader, 64, 64, 2, 32>(p, s, ws, wsn);
}

}  // namespace hnet
