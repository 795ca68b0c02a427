// kernels_conv.hip — instantiations and dispatch of the implicit-GEMM kernel (igemm.h) for the 20
// convolution layers of HomographyNet (reference model_to_trace.py:88-113, :210-216) and the heads' first FC.
#include "igemm.h"
#include "conv_first.h"
#include "igemm_s3.h"
#include "conv_b4_fused.h"
#include "conv_patch_s2.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>

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
static int splitk_min_iters(long tiles) {
    static const int v = std::getenv("HNET_SPLITK_MIN_ITERS") ? std::max(1, std::atoi(std::getenv("HNET_SPLITK_MIN_ITERS"))) : 0;
    return v ? v : (tiles < 16 ? 4 : 3);
}
static int splitk_target_blocks(long tiles) {
    static const int v = std::getenv("HNET_SPLITK_BLOCKS") ? std::max(64, std::atoi(std::getenv("HNET_SPLITK_BLOCKS"))) : 0;
    return v ? v : (tiles < 16 ? 192 : 384);
}

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
        static const int force = std::getenv("HNET_TILE") ? std::atoi(std::getenv("HNET_TILE")) : -1;   // experiments
        const long tiles128 = (long)((p.M + 127) / 128) * (COUT / 64);
        if (force == 0) return run<L, 64, 64, 2, 32>(p, s, ws, wsn);
        if (force == 1) return run<L, 128, 64, 2, 32>(p, s, ws, wsn);
        if constexpr (COUT >= 128) { if (force == 2) return run<L, 128, 128, 2, 32>(p, s, ws, wsn); }
        (void)tiles128;
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
// split-bf16 (S3) path, igemm_s3.h
// ---------------------------------------------------------------------------------------------
template <bool OUT32>
static hipError_t finish_split_impl(const S3Params& p, int split, float* ws, hipStream_t s) {
    if (split > 1) {
        if (OUT32) {
            const size_t total4 = (size_t)p.M * p.N / 4;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, ws, split, p.M, p.N, p.bias, p.out32);
        } else {
            const size_t total = (size_t)p.M * p.N;
            hipLaunchKernelGGL(splitk_reduce_s3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ws, split, p.M, p.N, p.bias, p.out16, p.o_plane);
        }
    }
    return hipGetLastError();
}
#define finish_split(p, split, ws, s) finish_split_impl<OUT32>(p, split, ws, s)

template <class L, int BM, int BN, int WGM, bool OUT32>
static hipError_t run_s3(S3Params p, hipStream_t s, float* ws, size_t ws_floats) {
    static const int nbuf = std::getenv("HNET_S3_NBUF") ? std::atoi(std::getenv("HNET_S3_NBUF")) : 1;   // experiments
    dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, 1);
    const long tiles = (long)grid.x * grid.y;
    const int n_iter = (p.Kp + IG_BK - 1) / IG_BK;
    int split = 1;
    if (ws && tiles < 192 && n_iter >= 8) {
        split = (int)std::min<long>(std::min<long>(n_iter / splitk_min_iters(tiles), (splitk_target_blocks(tiles) + tiles - 1) / tiles), 64);
        const size_t per = (size_t)p.M * p.N;
        if ((size_t)split * per > ws_floats) split = (int)(ws_floats / per);
        if (split < 2) split = 1;
    }
    p.k_split = split;
    p.partial = ws;
    grid.z = split;
    // XCD-aware tile mapping (igemm_s3.h): -2..-4 % on the >= 64-channel layers, +3 % on the 32-channel LDS-DMA layers -> wide taps only
    static const int xcd = std::getenv("HNET_XCD_REMAP") ? std::atoi(std::getenv("HNET_XCD_REMAP")) : -1;
    p.xcd_remap = xcd >= 0 ? xcd : (L::WIDE_TAPS ? 1 : 0);
    // LDS-DMA ring (3 stages) by default: 2-8 % faster than register staging on the 64x64 tiles (HNET_S3_DMA=0 disables)
    static const int dma = std::getenv("HNET_S3_DMA") ? std::atoi(std::getenv("HNET_S3_DMA")) : 3;
    if constexpr (!L::HAS_MASK && BM % 64 == 0 && BN % 64 == 0 && BM * BN <= 128 * 64 && !L::WIDE_TAPS) {
        if (dma && split == 1 && p.zeros) {
            if constexpr (BM * BN <= 128 * 64) {
                if (dma == 4) { hipLaunchKernelGGL((igemm_s3_dma_kernel<L, BM, BN, WGM, OUT32, 4>), grid, dim3(256), 0, s, p); return hipGetLastError(); }
            }
            // the 16x16x32 shape gains nothing on these short-K (288) 32-channel layers (0.110 vs 0.109 ms): 32x32x16 unless HNET_S3_MF16=2
            static const int mf16d = std::getenv("HNET_S3_MF16") ? std::atoi(std::getenv("HNET_S3_MF16")) : 0;
            if (mf16d == 2) hipLaunchKernelGGL((igemm_s3_dma_kernel<L, BM, BN, WGM, OUT32, 3, 16>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((igemm_s3_dma_kernel<L, BM, BN, WGM, OUT32, 3>), grid, dim3(256), 0, s, p);
            return hipGetLastError();
        }
    }
    // MFMA shape 16x16x32 (transposed tiles) by default: -4..-9 % per layer against 32x32x16 at the same LDS traffic (HNET_S3_MF16=0: 32x32x16)
    static const int mf16 = std::getenv("HNET_S3_MF16") ? std::atoi(std::getenv("HNET_S3_MF16")) : 1;
    // 64-wide K tiles (full 128-byte lines per staged row) for layers whose taps hold >= 64 channels: ~10 % faster than the
    // 32-wide tiles there (the texture addresser is the busy unit); HNET_S3_BK64=0 disables
    static const int bk64 = std::getenv("HNET_S3_BK64") ? std::atoi(std::getenv("HNET_S3_BK64")) : 1;
    static const int t96 = std::getenv("HNET_S3_TILE") ? std::atoi(std::getenv("HNET_S3_TILE")) : 0;
    const bool bk64_ok = BM != 96 || t96 == 6;               // 96-row tiles: 32-wide K tiles keep three workgroups per CU (61 KB of LDS at BK 64)
    if constexpr (BM * BN <= 128 * 64 && L::SEGMENT >= 32 && L::WIDE_TAPS) {
        if (bk64 && mf16 && bk64_ok) { hipLaunchKernelGGL((igemm_s3_kernel<L, BM, BN, WGM, OUT32, 1, 64, 16>), grid, dim3(256), 0, s, p); return finish_split(p, split, ws, s); }
        if constexpr (BM != 96) {
            if (bk64) { hipLaunchKernelGGL((igemm_s3_kernel<L, BM, BN, WGM, OUT32, 1, 64>), grid, dim3(256), 0, s, p); return finish_split(p, split, ws, s); }
        }
    }
    if constexpr (BM == 96) {     // 96-row tiles exist only in the 16x16x32 form (wave tile 48 x 32)
        hipLaunchKernelGGL((igemm_s3_kernel<L, BM, BN, WGM, OUT32, 1, 32, 16>), grid, dim3(256), 0, s, p);
    } else {
        if (nbuf == 2) hipLaunchKernelGGL((igemm_s3_kernel<L, BM, BN, WGM, OUT32, 2>), grid, dim3(256), 0, s, p);
        else if (mf16) hipLaunchKernelGGL((igemm_s3_kernel<L, BM, BN, WGM, OUT32, 1, 32, 16>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((igemm_s3_kernel<L, BM, BN, WGM, OUT32, 1>), grid, dim3(256), 0, s, p);
    }
    if (split > 1) {
        if (OUT32) {
            const size_t total4 = (size_t)p.M * p.N / 4;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, ws, split, p.M, p.N, p.bias, p.out32);
        } else {
            const size_t total = (size_t)p.M * p.N;
            hipLaunchKernelGGL(splitk_reduce_s3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ws, split, p.M, p.N, p.bias, p.out16, p.o_plane);
        }
    }
    return hipGetLastError();
}

template <int CIN, int KS, int STRIDE, int SEG, int COUT, bool OUT32>
static hipError_t run_conv_s3(const S3Params& p, hipStream_t s, float* ws, size_t wsn) {
    typedef ConvLoaderS3<CIN, KS, STRIDE, SEG> L;
    static const int tile = std::getenv("HNET_S3_TILE") ? std::atoi(std::getenv("HNET_S3_TILE")) : 0;   // experiments
    if constexpr (COUT <= 32) return run_s3<L, 128, 32, 4, OUT32>(p, s, ws, wsn);
    else {
        if (tile == 1) return run_s3<L, 128, 64, 2, OUT32>(p, s, ws, wsn);
        if constexpr (COUT >= 128) { if (tile == 2) return run_s3<L, 128, 128, 2, OUT32>(p, s, ws, wsn); }
        // long-K layers amortise a bigger tile (measured at batch 256): 256->256 3x3 (K 2304) 128x64, 128->128 5x5 (K 3200) 128x128
        const bool big_m = p.M >= 4096;
        if constexpr (CIN == 128 && KS == 3) { if (big_m && (tile == 5 || tile == 6)) return run_s3<L, 96, 64, 2, OUT32>(p, s, ws, wsn); }
        if constexpr (CIN == 256) { if (big_m) return run_s3<L, 128, 64, 2, OUT32>(p, s, ws, wsn); }
        if constexpr (CIN == 128 && KS == 5) { if (big_m && tile != 4) return run_s3<L, 128, 128, 2, OUT32>(p, s, ws, wsn); }
        return run_s3<L, 64, 64, 2, OUT32>(p, s, ws, wsn);
    }
}

bool conv_is_s3_layer(int layer) { return kConvs[layer].cin >= 8; }

hipError_t conv_kernels_init_device() {
    hipError_t e = hipFuncSetAttribute((const void*)block4_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, b4f::LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_patch_s2_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, PatchS2Cfg<5>::LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_patch_s2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, PatchS2Cfg<3>::LDS_BYTES);
    return e;
}

// block_4_0 + block_4_1 fused (conv_b4_fused.h): x_in fp32 [B][224][320][2] -> out16 S3 planes [3][B][112][160][16]
hipError_t launch_block4_fused(const float* x_in, const void* w0frag, const float* bias0, const void* w1frag, const float* bias1,
                               uint16_t* out16, size_t o_plane, int batch, hipStream_t s, int flags) {
    const int n_tiles = batch * (112 / b4f::TH1) * (160 / b4f::TW1);
    const unsigned blocks = (unsigned)std::min(n_tiles, 256);      // persistent: one 512-thread workgroup per CU (85 KB of LDS)
    int dbg = 0;
#ifdef HNET_B4_ABLATE   // profiling build only (make FLAGS+=-DHNET_B4_ABLATE): HNET_B4_DBG drops phases of the kernel, results are wrong
    static const int dbg_env = std::getenv("HNET_B4_DBG") ? std::atoi(std::getenv("HNET_B4_DBG")) : 0;
    dbg = dbg_env & 7;
#endif
    hipLaunchKernelGGL(block4_fused_kernel, dim3(blocks), dim3(b4f::THREADS), b4f::LDS_BYTES, s, x_in, (const u32x4*)w0frag, bias0,
                       (const u32x4*)w1frag, bias1, out16, o_plane, n_tiles, dbg | ((flags & 1) ? 8 : 0));
    return hipGetLastError();
}

// block_3_0 on the split-bf16 path (conv_first.h): x_in fp32 [B][h][w][2] -> out16 S3 planes [3][B][h][w][16]
hipError_t launch_conv_first_s3(const float* x_in, const void* wfrag, const float* bias, uint16_t* out16, size_t o_plane, int batch,
                                int h, int w, hipStream_t s) {
    const int tx = (w + 31) / 32, ty = (h + 15) / 16;
    hipLaunchKernelGGL(conv7_c2_s1_s3_kernel, dim3((unsigned)(batch * tx * ty)), dim3(256), 0, s, x_in, (const u32x4*)wfrag, bias, out16,
                       o_plane, h, w, tx, ty);
    return hipGetLastError();
}

// block_3_1 (5x5) / block_4_2 (3x3): 16 -> 32 channels, stride 2, from an LDS-resident patch (conv_patch_s2.h)
bool conv_is_patch_layer(int layer) { return layer == 8 || layer == 15; }   // block_3_1 (5x5), block_4_2 (3x3)

template <int KS>
static hipError_t run_patch(const uint16_t* in, size_t i_plane, const void* wfrag, const float* bias, uint16_t* out16,
                            size_t o_plane, int batch, int h, int w, hipStream_t s) {
    typedef PatchS2Cfg<KS> C;
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    const int n_tiles = batch * ((ho + C::TH - 1) / C::TH) * ((wo + C::TW - 1) / C::TW);
    const unsigned blocks = (unsigned)std::min(n_tiles, 512);      // persistent, 2 workgroups per CU
    // bit 0: reverse the 5x5 kernel (block_3_1), bit 1: reverse the 3x3 kernel (block_4_2)
    static const int rev = std::getenv("HNET_PATCH_REV") ? std::atoi(std::getenv("HNET_PATCH_REV")) : 3;
    hipLaunchKernelGGL(conv_patch_s2_kernel<KS>, dim3(blocks), dim3(256), C::LDS_BYTES, s, in, i_plane, (const u32x4*)wfrag, bias,
                       out16, o_plane, h, w, n_tiles, KS == 5 ? (rev & 1) : ((rev >> 1) & 1));
    return hipGetLastError();
}

hipError_t launch_conv_patch(int layer, const uint16_t* in, size_t i_plane, int batch, int h, int w, const void* wfrag,
                             const float* bias, uint16_t* out16, size_t o_plane, hipStream_t s) {
    if (layer == 8) return run_patch<5>(in, i_plane, wfrag, bias, out16, o_plane, batch, h, w, s);
    if (layer == 15) return run_patch<3>(in, i_plane, wfrag, bias, out16, o_plane, batch, h, w, s);
    return hipErrorInvalidValue;
}

// first FC of both heads on the split-bf16 path.  feat fp32 [B][5120]; w1planes [3][512][5120] bf16;
// scratch: feat16 [3][B][5120] bf16 and mask [B][n_local][2][640] bytes (context-owned)
hipError_t launch_heads_fc1_s3(const float* feat, int batch, int n_local, int s_begin, float p_drop, uint64_t mc_seed,
                               uint64_t pair_seq0, const uint16_t* w1planes, const float* b1, float* hidden,
                               uint16_t* feat16, size_t f_plane, uint8_t* mask, hipStream_t s, float* ws, size_t wsn,
                               const uint64_t* seq_dev) {
    const size_t nwork = std::max((size_t)batch * 5120, (size_t)batch * n_local * 2 * 640);
    hipLaunchKernelGGL(heads_prep_kernel, dim3((unsigned)((nwork + 255) / 256)), dim3(256), 0, s, feat, batch, n_local, s_begin,
                       hnet_drop_threshold(p_drop), 1.0f / (1.0f - p_drop), mc_seed, pair_seq0, seq_dev, feat16, f_plane, mask);
    S3Params p = {};
    p.A = feat16; p.a_plane = f_plane; p.Wp = w1planes; p.w_plane = (size_t)512 * 5120; p.bias = b1;
    p.out32 = hidden;
    p.M = batch * n_local; p.N = 512; p.Kp = 5120;
    p.mask = mask; p.n_local = n_local;
    // K = 5120 (160 K-tiles): the 128x64 tile amortises better (0.317 vs 0.353 ms at batch 256); small M keeps 64x64 + split-K
    if (p.M >= 4096) return run_s3<HeadLoaderS3, 128, 64, 2, true>(p, s, ws, wsn);
    return run_s3<HeadLoaderS3, 64, 64, 2, true>(p, s, ws, wsn);
}

hipError_t launch_conv_s3(int layer, const uint16_t* in, size_t in_plane, int batch, int h, int w, const uint16_t* wplanes,
                          size_t w_plane, const float* bias, uint16_t* out16, size_t o_plane, float* out32, hipStream_t s,
                          float* ws, size_t wsn, const uint16_t* zeros) {
    if (layer < 0 || layer >= 20 || !conv_is_s3_layer(layer)) return hipErrorInvalidValue;
    const ConvDesc& d = kConvs[layer];
    S3Params p = {};
    p.A = in; p.a_plane = in_plane; p.Wp = wplanes; p.w_plane = w_plane; p.bias = bias;
    p.out16 = out16; p.o_plane = o_plane; p.out32 = out32;
    p.zeros = zeros;
    p.H = h; p.W = w;
    p.Ho = conv_out_dim(h, d.ks, d.stride);
    p.Wo = conv_out_dim(w, d.ks, d.stride);
    p.M = batch * p.Ho * p.Wo;
    p.N = d.cout;
    p.Kp = conv_padded_k(layer);
    const bool o32 = out32 != nullptr;
    switch (layer) {
        case 1:  return run_conv_s3<128, 5, 2, 32, 128, false>(p, s, ws, wsn);
        case 2: case 5: case 11: case 18:
            return o32 ? run_conv_s3<128, 3, 2, 32, 256, true>(p, s, ws, wsn) : run_conv_s3<128, 3, 2, 32, 256, false>(p, s, ws, wsn);
        case 4:  return run_conv_s3<64, 5, 2, 32, 128, false>(p, s, ws, wsn);
        case 6: case 12: case 19:
            return o32 ? run_conv_s3<256, 3, 2, 32, 256, true>(p, s, ws, wsn) : run_conv_s3<256, 3, 2, 32, 256, false>(p, s, ws, wsn);
        case 8:  return run_conv_s3<16, 5, 2, 16, 32, false>(p, s, ws, wsn);
        case 9: case 16: return run_conv_s3<32, 3, 2, 32, 64, false>(p, s, ws, wsn);
        case 10: case 17: return run_conv_s3<64, 3, 2, 32, 128, false>(p, s, ws, wsn);
        case 14: return run_conv_s3<8, 5, 2, 8, 16, false>(p, s, ws, wsn);
        case 15: return run_conv_s3<16, 3, 2, 16, 32, false>(p, s, ws, wsn);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_nchw_f32_to_nhwc_s3(const float* in, uint16_t* out, size_t o_plane, int batch, int c, int h, int w, hipStream_t s) {
    const long n = (long)batch * c * h * w;
    hipLaunchKernelGGL(nchw_f32_to_nhwc_s3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, o_plane, batch, c, h * w);
    return hipGetLastError();
}
hipError_t launch_nhwc_s3_to_nchw_f32(const uint16_t* in, size_t i_plane, float* out, int batch, int c, int h, int w, hipStream_t s) {
    const long n = (long)batch * c * h * w;
    hipLaunchKernelGGL(nhwc_s3_to_nchw_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, i_plane, out, batch, c, h * w);
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
    static const int force = std::getenv("HNET_TILE") ? std::atoi(std::getenv("HNET_TILE")) : -1;   // experiments
    if (force == 1) return run<HeadLoader, 128, 64, 2, 32>(p, s, ws, wsn);
    return run<HeadLoader, 64, 64, 2, 32>(p, s, ws, wsn);
}

}  // namespace hnet
