// s3_dispatch.h — tile / kernel-variant selection of the bf16-matrix-core layers, templated on the number of bf16 planes
// NP (3 = split-bf16, the fp32-grade default; 1 = plain bf16 operands, HNET_PREC_BF16).  Included by kernels_conv.hip
// (instantiates NP = 3) and kernels_conv_bf16.hip (NP = 1) so that the two sets of kernels compile in parallel.
#pragma once
#include "igemm.h"
#include "conv_first.h"
#include "igemm_s3.h"
#include "igemm_pipe.h"
#include "igemm_region.h"
#include "heads_lat.h"
#include "conv_b4_fused.h"
#include "conv_b3_fused.h"
#include "conv_b42_fused.h"
#include "conv_patch_s2.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>

namespace hnet {

int splitk_min_iters(long tiles);        // kernels_conv.hip
int splitk_target_blocks(long tiles);

template <bool OUT32, int NP>
static hipError_t finish_split_impl(const S3Params& p, int split, float* ws, hipStream_t s) {
    if (split > 1) {
        if (OUT32) {
            const size_t total4 = (size_t)p.M * p.N / 4;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, ws, split, p.M, p.N, p.bias, p.out32);
        } else {
            const size_t total = (size_t)p.M * p.N;
            hipLaunchKernelGGL(splitk_reduce_s3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ws, split, p.M, p.N, p.bias, p.out16, p.o_plane, NP);
        }
    }
    return hipGetLastError();
}
#define finish_split(p, split, ws, s) (((split) > 1 && lat) ? (void)(lat->kernels = 2) : (void)0, finish_split_impl<OUT32, NP>(p, split, ws, s))

constexpr int LEAN8_LDS_BYTES = 2 * (2 * 128 * 64 + 2 * 128 * 64) * 2 + 256 * 16;      // igemm_s3_lean8_kernel: two buffers of tiles + the mask table

// LEAN8: the eight-wave double-buffered kernel (igemm_s3_lean8_kernel; 128 x 128 tiles, fp16-plane mode, layers whose K-tiles hold 64 channels)
template <class L, int BM, int BN, int WGM, bool OUT32, int NP, bool LEAN8 = false>
static hipError_t run_s3(S3Params p, hipStream_t s, float* ws, size_t ws_floats, LatIO* lat = nullptr) {
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
    if (lat) lat->kernels = 1;
    // round 5: a split-K launch of the lean kernels with at most 40 GEMM rows (the 4 x 5 layers of one or two pairs) is finished by the last workgroup of
    // each tile to arrive (lat->tickets, igemm_s3.h s3_splitk_last_arriver) instead of a splitk_reduce* launch: 13.3 -> 11.3 us per layer at batch 1.  From 70
    // rows on, one workgroup summing a 64 x 64 tile is slower than the reduce launch that spreads it (block_2_2 12.9 -> 16.1 us): those keep the launch.
    // hnet_config.variant 30: the reduce launches everywhere (A/B, bitwise tests)
    uint32_t* const tickets = split > 1 && lat && lat->tickets && p.M <= 40 && tiles <= SPLITK_TICKETS ? lat->tickets : nullptr;
    // XCD-aware tile mapping (igemm_s3.h): -2..-4 % on the >= 64-channel layers, +3 % on the 32-channel ones -> wide taps only
    p.xcd_remap = L::WIDE_TAPS ? 1 : 0;

    if constexpr (NP == 2 && BM == 128 && BN == 128 && L::WIDE_TAPS && LEAN8) {      // the eight-wave double-buffered kernel of round 3 (heads; since round 4 only as the A/B reference of igemm_heads_pipe_kernel)
        hipLaunchKernelGGL((igemm_s3_lean8_kernel<L, OUT32, NP>), grid, dim3(512), LEAN8_LDS_BYTES, s, p);
        return finish_split(p, split, ws, s);
    }
    // The lean kernel (buffer loads with scalar tap offsets, igemm_s3.h) on every layer it covers, with 64-wide K tiles where a tap holds >= 64
    // channels; the register-staged kernel in its 16x16x32 form for the rest (ragged operator-level shapes).  Round-2 / round-3 measurements that
    // chose this (all removed from the tree in round 4, profiles/r0*_experiments_not_shipped.log): 32x32x16 MFMA shape, double-buffered register
    // staging, the LDS-DMA ring (igemm_s3_dma_kernel), 96 / 64x128 / 128x64 tiles for the 128-channel layers, the eight-wave kernel on the conv layers.
    if constexpr (NP != 1 && BM * BN <= 128 * 64 && L::SEGMENT >= 32 && L::WIDE_TAPS) {      // (plain bf16, reported only: the register-staged kernel throughout)
        if constexpr (L::template lean_ok<64>()) {
            p.tickets = tickets;
            hipLaunchKernelGGL((igemm_s3_lean_kernel<L, BM, BN, WGM, OUT32, 64, NP>), grid, dim3(256), 0, s, p);
            return tickets ? hipGetLastError() : finish_split(p, split, ws, s);
        }
    }
    if constexpr (NP != 1 && L::template lean_ok<32>()) {
        p.tickets = tickets;
        hipLaunchKernelGGL((igemm_s3_lean_kernel<L, BM, BN, WGM, OUT32, 32, NP>), grid, dim3(256), 0, s, p);
        if (tickets) return hipGetLastError();
    } else if constexpr (BM * BN <= 128 * 64 && L::SEGMENT >= 32 && L::WIDE_TAPS)
        hipLaunchKernelGGL((igemm_s3_kernel<L, BM, BN, WGM, OUT32, 1, 64, 16, NP>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((igemm_s3_kernel<L, BM, BN, WGM, OUT32, 1, 32, 16, NP>), grid, dim3(256), 0, s, p);
    return finish_split(p, split, ws, s);
}

// igemm_pipe.h: the software-pipelined LDS-DMA kernel on whole-pair tiles (fp16-plane mode).  Per layer shape the tile that makes the grid
// exactly 256 or 512 workgroups at batch 256 (70 / 280 / 20 GEMM rows per pair):
//   128 -> 256 3x3 (block_2_3 / 3_4 / 4_5, 70/pair)   160 (140) x 128, eight waves of 80 x 32                    2 pairs x 2 channel halves: 256 workgroups
//   64 -> 128 5x5 / 3x3 (block_2_2, 3_3, 4_4, 280)    160 (140) x 128, eight waves                               half a pair: 512 workgroups
template <class L, class C, bool OUT32>
static hipError_t run_pipe(S3Params p, hipStream_t s) {
    dim3 grid((p.M + C::BMV - 1) / C::BMV, p.N / C::BN, 1);
    p.k_split = 1;
    hipLaunchKernelGGL((igemm_s3_pipe_kernel<L, C, OUT32>), grid, dim3(C::NT), C::LDS_BYTES, s, p);
    return hipGetLastError();
}
typedef PipeCfg<3, 2, 3, 4, 140> PipeCfg144;     // 144 x 128, 768 threads (three waves per SIMD): 2.8 % instead of 12.5 % padded rows
// the layers it serves, measured in process against the four-wave lean kernels at batch 256 (profiles/r04_ab_pipe*.log): the 160 (140) x 128 tile
// wins on block_2_2 (- 18 %), block_2_3 / 3_4 / 4_5 (- 22 %) and block_3_3 / 4_4 (- 2 ... - 6 %).  Four-wave tiles of 80 x 128 / 80 x 64 for
// block_1_2 and the 4 x 5 layers LOST (+ 4 % / + 21 %): they need 55 / 77 B / clk / CU from the texture addresser, whose limit is 64
template <int CIN, int KS, int COUT>
static bool pipe_ok(const S3Params& p) {
    const int rows = p.Ho * p.Wo;                 // GEMM rows per pair
    // one workgroup per CU: it pays from three quarters of a round of tiles on (measured, prior-3 / N = 16, ms pipelined / lean: batch 64 - 64 tiles of the
    // 128 -> 256 layers 0.0349 / 0.0244, 128 tiles of the 64 -> 128 ones 0.0446 / 0.0427; batch 128 - 128 tiles 0.0355 / 0.0320, 256 tiles 0.0502 / 0.0657)
    const long tiles = (long)((p.M + 139) / 140) * (p.N / 128);
    if ((tiles < 192 && p.tile != 21) || p.tile == 20) return false;    // (HNET_S3_TILE=20: the four-wave lean kernels, A/B; 21: this kernel at any M, tests)
    if (CIN == 128 && KS == 3 && COUT == 256 && rows % 70 == 0) return true;
    if (CIN == 64 && COUT == 128 && rows % 140 == 0) return true;
    return false;
}

// igemm_region.h: the input region of the tile's pairs resident in LDS, the weights straight into registers (layers bound by operand delivery).
// p.wfrag carries the packed weight fragments (hnet_create; nullptr = not packed: the lean kernels)
template <class C, bool OUT32>
static hipError_t run_region(S3Params p, hipStream_t s) {
    const int rows_tile = C::P * p.Ho * p.Wo;
    dim3 grid((p.M + rows_tile - 1) / rows_tile, p.N / C::BN, 1);
    p.k_split = 1;
    p.Wp = p.wfrag;
    hipLaunchKernelGGL((igemm_s3_region_kernel<C, OUT32>), grid, dim3(C::NT), C::LDS_BYTES, s, p);
    return hipGetLastError();
}
typedef RegionCfg<128, 5, 1, 280, false, 14, 20> RegionCfg12;     // block_1_2: one pair (14 x 20 x 128 in, 7 x 10 out) x 128 channels per workgroup: 256 workgroups at batch 256
typedef RegionCfg<128, 3, 4, 288, true, 7, 10> RegionCfg13;       // block_1_3: four pairs (7 x 10 x 128 in, 4 x 5 out) x 64 channels, K halves over the wave halves
typedef RegionCfg<256, 3, 4, 288, true, 7, 10> RegionCfgT;        // block_2_4 / 3_5 / 4_6: the same with 256 input channels
template <class C>
static bool region_ok(const S3Params& p) {
    // (variant 25: the lean kernels, A/B; 21: at any batch, tests)
    // (a one-workgroup-per-CU kernel as well: from three quarters of a round on - batch 128, 128 workgroups of the 4 x 5 layers: 0.0318 against 0.0235 ms lean)
    const long wgs = (long)((p.M + C::P * p.Ho * p.Wo - 1) / (C::P * p.Ho * p.Wo)) * (p.N / C::BN);
    return p.wfrag && p.H == C::HI && p.W == C::WI && p.Ho == C::HO && p.Wo == C::WO && p.tile != 20 && p.tile != 25 && (wgs >= 192 || p.tile == 21) && 2 * C::P * (((p.Ho * p.Wo + 1) & ~1) + (((p.H >> 1) * p.Wo + 1) & ~1)) <= C::RP && C::P * p.Ho * p.Wo <= 80 &&
           (p.M % (p.Ho * p.Wo)) == 0 && p.W == 2 * p.Wo && ((p.H + 1) >> 1) == p.Ho && p.N % C::BN == 0;
}

template <int CIN, int KS, int STRIDE, int SEG, int COUT, bool OUT32, int NP>
static hipError_t run_conv_s3(const S3Params& p, hipStream_t s, float* ws, size_t wsn, LatIO* lat) {
    typedef ConvLoaderS3<CIN, KS, STRIDE, SEG> L;
    if constexpr (NP == 2 && !OUT32 && CIN == 128 && KS == 5 && COUT == 128) {
        if (region_ok<RegionCfg12>(p)) return run_region<RegionCfg12, OUT32>(p, s);
    }
    if constexpr (NP == 2 && KS == 3 && COUT == 256) {
        if (p.Ho * p.Wo == 20) {                                  // the 4 x 5 layers
            if constexpr (CIN == 128) { if (region_ok<RegionCfg13>(p)) return run_region<RegionCfg13, OUT32>(p, s); }
            if constexpr (CIN == 256) { if (region_ok<RegionCfgT>(p)) return run_region<RegionCfgT, OUT32>(p, s); }
        }
    }
    if constexpr (NP == 2 && !OUT32 && ((CIN == 128 && KS == 3 && COUT == 256) || (CIN == 64 && COUT == 128))) {
        if (pipe_ok<CIN, KS, COUT>(p)) return run_pipe<L, PipeCfg144, OUT32>(p, s);      // (the 160 x 128 / 512-thread tile of the first build lost to it and was removed in round 5)
    }
    if constexpr (COUT <= 32) return run_s3<L, 128, 32, 4, OUT32, NP>(p, s, ws, wsn, lat);
    else {
        // long-K layers amortise a bigger tile (measured at batch 256): 256 -> 256 3x3 (K 2304) 128 x 64; 128 -> 128 5x5 (K 3200, N = 128: the im2col tile
        // staged once for all of N) 128 x 128 - in the fp16 mode from M = 8192 (64 x 64 / 128 x 128 at batch 64: 0.0291 / 0.0344, 128: 0.0537 / 0.0448,
        // 256: 0.0734 / 0.0676 ms); every other layer 64 x 64 = four workgroups per CU (profiles/r03_experiments_not_shipped.log, items 7, 9, 17)
        const bool big_m = p.M >= 4096;
        if constexpr (CIN == 256) { if (big_m) return run_s3<L, 128, 64, 2, OUT32, NP>(p, s, ws, wsn); }
        if constexpr (CIN == 128 && KS == 5) { if (NP == 2 ? p.M >= 8192 : big_m) return run_s3<L, 128, 128, 2, OUT32, NP>(p, s, ws, wsn); }
        return run_s3<L, 64, 64, 2, OUT32, NP>(p, s, ws, wsn, lat);
    }
}

// block_4_0 + block_4_1 fused (conv_b4_fused.h): x_in = the padded 16-bit planes of kernels.h B4_* (x_plane dwords per plane) -> out16 planes [NP][B][112][160][16].
// 7 x 32 tiles, two 256-thread workgroups per CU, LDS-DMA staging, phase-1 fragment reuse: the winner of the round-2 / round-3 measurements
// (8 x 32 / 512 threads, fp32 input without DMA, no fragment reuse and the v2 kernel were removed in round 4; DESIGN.md section 3.4 keeps the table)
template <int TH1, int NP>
static hipError_t run_block4_fused(const void* x_in, size_t x_plane, const void* w0frag, const float* bias0, const void* w1frag, const float* bias1,
                                   uint16_t* out16, size_t o_plane, int batch, hipStream_t s, int flags, const B4Warp* warp = nullptr) {
    typedef B4Cfg<TH1, 256, NP, true> C;
    const int n_tiles = batch * (112 / C::TH1) * (160 / C::TW1);
    const int per_cu = std::max(1, std::min(2, std::min(2048 / 256, (160 * 1024) / C::LDS_BYTES)));   // two waves per SIMD (launch bounds)
    const unsigned blocks = (unsigned)std::min(n_tiles, 256 * per_cu);        // persistent
    if constexpr (NP == 2 && TH1 == 8) {
        static_assert(2 * (C::LDS_BYTES + B4W_LDS_BYTES) <= 160 * 1024, "two workgroups per CU with the warp box");
        if (warp) {                                                           // the kernel samples its own patches (conv_b4_fused.h WARPIN): x_in is not read
            hipLaunchKernelGGL((block4_fused_kernel<TH1, 256, NP, true, true, true>), dim3(blocks), dim3(256), C::LDS_BYTES + B4W_LDS_BYTES, s, x_in, x_plane,
                               (const u32x4*)w0frag, bias0, (const u32x4*)w1frag, bias1, out16, o_plane, n_tiles, flags, *warp);
            return hipGetLastError();
        }
    } else if (warp) return hipErrorInvalidValue;
    hipLaunchKernelGGL((block4_fused_kernel<TH1, 256, NP, true, true>), dim3(blocks), dim3(256), C::LDS_BYTES, s, x_in, x_plane,
                       (const u32x4*)w0frag, bias0, (const u32x4*)w1frag, bias1, out16, o_plane, n_tiles, flags, B4Warp{});
    return hipGetLastError();
}
template <int NP>
hipError_t launch_block4_fused_np(const void* x_in, size_t x_plane, const void* w0frag, const float* bias0, const void* w1frag, const float* bias1,
                                  uint16_t* out16, size_t o_plane, int batch, hipStream_t s, int flags, const B4Warp* warp) {
    flags &= 113;
    // fp16-plane mode: 8 x 32 tiles (57 KB of LDS: still two workgroups per CU; 16 phase-2 M-tiles = four per wave exactly, 19 / 16 rows of halo instead of 17 / 14).
    // (the 7 x 32 tiles of rounds 2 - 3 remain the tile of the three-plane modes, which need them for two workgroups per CU)
    if constexpr (NP == 2) return run_block4_fused<8, NP>(x_in, x_plane, w0frag, bias0, w1frag, bias1, out16, o_plane, batch, s, flags & ~32, warp);
    else return warp ? hipErrorInvalidValue : run_block4_fused<7, NP>(x_in, x_plane, w0frag, bias0, w1frag, bias1, out16, o_plane, batch, s, flags);
}

// block_3_0 + block_3_1 in one kernel (conv_b3_fused.h; fp16-plane mode only): x_in fp32 [B][112][160][2] -> out16 fp16 planes [2][B][56][80][32]
template <int NP>
hipError_t launch_block3_fused_np(const float* x_in, const void* w0frag, const float* bias0, const void* w1frag, const float* bias1, uint16_t* out16,
                                  size_t o_plane, int batch, hipStream_t s) {
    if constexpr (NP != 2) return hipErrorInvalidValue;
    else {
        const int n_tiles = batch * B3Cfg::TILES_X * B3Cfg::TILES_Y;
        hipLaunchKernelGGL(block3_fused_kernel<NP>, dim3((unsigned)std::min(n_tiles, 512)), dim3(256), B3Cfg::LDS_BYTES + B3Cfg::W0_BYTES + B3Cfg::SPARE_BYTES, s, x_in, (const u32x4*)w0frag, bias0,
                           (const u32x4*)w1frag, bias1, out16, o_plane, n_tiles);
        return hipGetLastError();
    }
}

// block_4_2 + block_4_3 in one kernel (conv_b42_fused.h; fp16-plane mode only): in16 [2][B][112][160][16] -> out16 [2][B][28][40][64]
template <int NP>
hipError_t launch_block42_fused_np(const uint16_t* in16, size_t i_plane, const void* w2frag, const float* bias2, const void* w3frag, const float* bias3,
                                   uint16_t* out16, size_t o_plane, int batch, hipStream_t s) {
    if constexpr (NP != 2) return hipErrorInvalidValue;
    else {
        const int n_tiles = batch * B42Cfg::TILES_X * B42Cfg::TILES_Y;
        hipLaunchKernelGGL(block42_fused_kernel<NP>, dim3((unsigned)std::min(n_tiles, 512)), dim3(256), B42Cfg::LDS_BYTES, s, in16, i_plane,
                           (const u32x4*)w2frag, bias2, (const u32x4*)w3frag, bias3, out16, o_plane, n_tiles);
        return hipGetLastError();
    }
}

// block_3_0 on the bf16 matrix cores (conv_first.h): x_in fp32 [B][h][w][2] -> out16 S3 planes [3][B][h][w][16]
template <int NP>
hipError_t launch_conv_first_s3_np(const float* x_in, const void* wfrag, const float* bias, uint16_t* out16, size_t o_plane, int batch,
                                   int h, int w, hipStream_t s) {
    const int tx = (w + 31) / 32, ty = (h + 15) / 16;
    const int n_tiles = batch * tx * ty;
    constexpr int per_cu = 2;    // measured 0.113 (2) / 0.119 (3) / 0.114 (4) ms at batch 256
    hipLaunchKernelGGL(conv7_c2_s1_s3_kernel<NP>, dim3((unsigned)std::min(n_tiles, 256 * per_cu)), dim3(256), 0, s, x_in, (const u32x4*)wfrag, bias,
                       out16, o_plane, h, w, tx, ty, n_tiles);
    return hipGetLastError();
}

// block_1_1 (layer 0: 2 -> 128 @28x40) / block_2_1 (layer 3: 2 -> 64 @56x80), 7x7 stride 2 (conv_first.h conv7_c2_s2_s3_kernel)
template <int NP>
hipError_t launch_conv_first_s2_np(int layer, const float* x_in, const void* wfrag, const float* bias, uint16_t* out16, size_t o_plane,
                                   int batch, hipStream_t s) {
    // bands of 7 output rows amortise the weight loads at large batches; at small ones (the batch-1 latency path) 2-row bands give
    // 7 / 14 workgroups per pair instead of 2 / 4
    const bool small = batch <= 16;
    if (layer == 0) {
        if (small)
            hipLaunchKernelGGL((conv7_c2_s2_s3_kernel<128, 14, 20, 2, NP>), dim3((unsigned)(batch * 7)), dim3(256), 0, s, x_in, (const u32x4*)wfrag,
                               bias, out16, o_plane);
        else
            hipLaunchKernelGGL((conv7_c2_s2_s3_kernel<128, 14, 20, 7, NP>), dim3((unsigned)(batch * 2)), dim3(256), 0, s, x_in, (const u32x4*)wfrag,
                               bias, out16, o_plane);
    } else if (layer == 3) {
        if (small)
            hipLaunchKernelGGL((conv7_c2_s2_s3_kernel<64, 28, 40, 2, NP>), dim3((unsigned)(batch * 14)), dim3(256), 0, s, x_in, (const u32x4*)wfrag,
                               bias, out16, o_plane);
        else
            hipLaunchKernelGGL((conv7_c2_s2_s3_kernel<64, 28, 40, 7, NP>), dim3((unsigned)(batch * 4)), dim3(256), 0, s, x_in, (const u32x4*)wfrag,
                               bias, out16, o_plane);
    }
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// block_3_1 (5x5) / block_4_2 (3x3): 16 -> 32 channels, stride 2, from an LDS-resident patch (conv_patch_s2.h)
template <int KS, int NP>
static hipError_t run_patch(const uint16_t* in, size_t i_plane, const void* wfrag, const float* bias, uint16_t* out16,
                            size_t o_plane, int batch, int h, int w, hipStream_t s, bool b128, int rb5) {
    typedef PatchS2Cfg<KS, NP> C;
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    const int n_tiles = batch * ((ho + C::TH - 1) / C::TH) * ((wo + C::TW - 1) / C::TW);
    const unsigned blocks = (unsigned)std::min(n_tiles, 512);      // persistent, 2 workgroups per CU
    // the tiles are walked from the end of the batch (Infinity-Cache order): bit 0 the 5x5 kernel (block_3_1), bit 1 the 3x3 kernel (block_4_2)
    constexpr int rev = 3;
    const int r = KS == 5 ? (rev & 1) : ((rev >> 1) & 1);
    // block_3_1 / block_4_2 read their fragments with ds_read_b128 from the interleaved layout; the 5 x 5 kernel stages as many region rows per batch of loads as
    // fit next to its weight registers (5 in the fp16 mode, 3 in split-bf16: 0.156 (1) / 0.142 (2) / 0.123 ms (5) at batch 256).  The half-major b64 form and
    // the 1 / 2-row forms (the A/B references of rounds 2 - 4) were removed in round 5: `b128` must be true, `rb5` is ignored.
    if (!b128) return hipErrorInvalidValue;
    (void)rb5;
    if constexpr (KS == 5 && NP != 1) {
        constexpr int RBMAX = NP == 2 ? 5 : 3;   // what fits the 256 registers next to the 156 weight registers
        hipLaunchKernelGGL((conv_patch_s2_kernel<KS, NP, true, RBMAX>), dim3(blocks), dim3(256), C::LDS_BYTES, s, in, i_plane, (const u32x4*)wfrag, bias,
                           out16, o_plane, h, w, n_tiles, r);
    } else {
        hipLaunchKernelGGL((conv_patch_s2_kernel<KS, NP, true>), dim3(blocks), dim3(256), C::LDS_BYTES, s, in, i_plane, (const u32x4*)wfrag, bias,
                           out16, o_plane, h, w, n_tiles, r);
    }
    return hipGetLastError();
}

template <int NP>
hipError_t launch_conv_patch_np(int layer, const uint16_t* in, size_t i_plane, int batch, int h, int w, const void* wfrag,
                                const float* bias, uint16_t* out16, size_t o_plane, hipStream_t s, bool b128, int rb5) {
    if (layer == 8) return run_patch<5, NP>(in, i_plane, wfrag, bias, out16, o_plane, batch, h, w, s, b128, rb5);
    if (layer == 15) return run_patch<3, NP>(in, i_plane, wfrag, bias, out16, o_plane, batch, h, w, s, b128, rb5);
    if ((layer == 9 || layer == 16) && h == 56 && w == 80) {     // block_3_2 / block_4_3
        typedef Patch32Cfg<NP> C;
        const int n_tiles = batch * (28 / C::TH) * (40 / C::TW);
        constexpr int rev = 3;      // (bit 2 = conv_patch32: forward walk)
        if constexpr (NP == 2)      // the next tile's region in flight under the MFMAs (fp16-plane mode; the kernel without prefetch serves the other modes)
            hipLaunchKernelGGL((conv_patch32_s2_pf_kernel<NP>), dim3((unsigned)std::min(n_tiles, 512)), dim3(256), C::LDS_BYTES, s, in, i_plane,
                               (const u32x4*)wfrag, bias, out16, o_plane, n_tiles, (rev >> 2) & 1);
        else
            hipLaunchKernelGGL((conv_patch32_s2_kernel<NP>), dim3((unsigned)std::min(n_tiles, 512)), dim3(256), C::LDS_BYTES, s, in, i_plane,
                               (const u32x4*)wfrag, bias, out16, o_plane, n_tiles, (rev >> 2) & 1);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

// first FC of both heads on the bf16 matrix cores.  feat fp32 [B][5120]; w1planes [3][512][5120] bf16;
// scratch: feat16 [3][B][5120] bf16 and mask [B][n_local][2][640] bytes (context-owned)
template <int NP>
hipError_t launch_heads_fc1_s3_np(const float* feat, int batch, int n_local, int s_begin, float p_drop, uint64_t mc_seed,
                                  uint64_t pair_seq0, const uint16_t* w1planes, const float* b1, float* hidden,
                                  uint16_t* feat16, size_t f_plane, uint8_t* mask, hipStream_t s, float* ws, size_t wsn,
                                  const uint64_t* seq_dev, int tile, LatIO* lat) {
    if constexpr (NP == 2) {
        // latency path (round 5): keep bits, then ONE launch that owns four hidden units per workgroup over the whole K (heads_lat.h) instead of
        // feature planes + split-K GEMM + reduce
        if (lat && lat->heads_one_launch && batch <= 8 && n_local <= 16 * HL_MAXG) {
            lat->kernels = lat->mask_ready ? 1 : 2;
            if (!lat->mask_ready) {
                const size_t nmw = (size_t)batch * n_local * 2 * 160;
                hipLaunchKernelGGL(heads_prep_kernel, dim3((unsigned)((nmw + 255) / 256)), dim3(256), 0, s, feat, batch, n_local, s_begin,
                                   hnet_drop_threshold(p_drop), 1.0f / (1.0f - p_drop), mc_seed, pair_seq0, seq_dev, (uint16_t*)nullptr, f_plane, mask, NP, 0);
            }
            // pairs per workgroup: two y-groups of 128 workgroups = one round of the 256 CUs (this is a one-workgroup-per-CU kernel), the weights fetched once per workgroup
            const int ppw = (batch + 1) / 2, gy = (batch + ppw - 1) / ppw;
            if (n_local <= 32)
                hipLaunchKernelGGL(heads_fc1_lat_kernel<2>, dim3(512 / HL_UN, (unsigned)gy), dim3(HL_NT), HL_LDS_BYTES, s, feat, w1planes, (size_t)512 * 5120, b1, mask,
                                   n_local, 1.0f / (1.0f - p_drop), hidden, batch, ppw);
            else
                hipLaunchKernelGGL(heads_fc1_lat_kernel<HL_MAXG>, dim3(512 / HL_UN, (unsigned)gy), dim3(HL_NT), HL_LDS_BYTES, s, feat, w1planes, (size_t)512 * 5120, b1, mask,
                                   n_local, 1.0f / (1.0f - p_drop), hidden, batch, ppw);
            return hipGetLastError();
        }
    }
    const size_t nwork = std::max((size_t)batch * 5120, (size_t)batch * n_local * 2 * 160);      // one feature element, four mask bytes per thread
    // heads_prep_kernel forms its byte and row indices in 32 bits (i0 = blockIdx.x * 1024, row = i0 / 640): refuse what would wrap
    // (batch x n_local beyond ~3.3 M rows, or more than ~200 k pairs; hnet_create rejects such a max_batch x N as well)
    if (4 * nwork + 1024 >= ((size_t)1 << 32)) return hipErrorInvalidValue;
    // whole rounds of 128 x 128 tiles on the 256 CUs: the pipelined kernel (igemm_pipe.h), which reads its keep bits K-tile major.  It is a
    // one-workgroup-per-CU kernel, so it pays when its 4 x M / 128 tiles fill whole rounds (heads_fc1, ms, four-wave 128 x 64 / eight waves, N = 32, round 3:
    // batch 128 0.116 / 0.112, 192 0.162 / 0.144, 256 0.180 / 0.158, 320 0.255 / 0.276, 384 0.264 / 0.278, 512 0.359 / 0.319) -> up to one round, or when
    // the last round is at least three quarters full.  tile (hnet_config.variant): 13 = the four-wave kernel, 22 = the eight-wave kernel of round 3 (A/B, bitwise tests)
    const int Mh = batch * n_local;
    const long t8h = (long)((Mh + 127) / 128) * 4;
    const bool one_per_cu = Mh >= 4096 && (t8h <= 256 || t8h % 256 == 0 || t8h % 256 >= 192) && tile != 13;
    const bool pipe = NP == 2 && one_per_cu && tile != 22;
    hipLaunchKernelGGL(heads_prep_kernel, dim3((unsigned)((nwork + 255) / 256)), dim3(256), 0, s, feat, batch, n_local, s_begin,
                       hnet_drop_threshold(p_drop), 1.0f / (1.0f - p_drop), mc_seed, pair_seq0, seq_dev, feat16, f_plane, mask, NP, pipe ? 1 : 0);
    LatIO lat_count = {};                      // (feature planes + keep bits, the GEMM, its reduce launch if it splits K)
    if (!lat) lat = &lat_count;
    lat->tickets = nullptr;
    S3Params p = {};
    p.A = feat16; p.a_plane = f_plane; p.Wp = w1planes; p.w_plane = (size_t)512 * 5120; p.bias = b1;
    p.out32 = hidden;
    p.M = batch * n_local; p.N = 512; p.Kp = 5120;
    p.mask = mask; p.n_local = n_local; p.tile = tile;
    if constexpr (NP == 2) {
        if (pipe) {
            p.k_split = 1;
            hipLaunchKernelGGL(igemm_heads_pipe_kernel<NP>, dim3((unsigned)((p.M + 127) / 128), 4, 1), dim3(HeadsPipeCfg::NT), HeadsPipeCfg::LDS_BYTES, s, p);
            lat->kernels = 2;
            return hipGetLastError();
        }
    }
    hipError_t e = hipSuccess;
    bool done = false;
    if constexpr (NP == 2) {
        if (one_per_cu) { e = run_s3<HeadLoaderS3, 128, 128, 2, true, NP, true>(p, s, ws, wsn, lat); done = true; }
    }
    // K = 5120 (160 K-tiles): the 128x64 tile amortises better (0.317 vs 0.353 ms at batch 256); small M keeps 64x64 + split-K
    if (!done) e = p.M >= 4096 ? run_s3<HeadLoaderS3, 128, 64, 2, true, NP>(p, s, ws, wsn, lat) : run_s3<HeadLoaderS3, 64, 64, 2, true, NP>(p, s, ws, wsn, lat);
    lat->kernels += 1;                         // + heads_prep_kernel
    return e;
}

template <int NP>
hipError_t launch_conv_s3_np(int layer, const uint16_t* in, size_t in_plane, int batch, int h, int w, const uint16_t* wplanes,
                             size_t w_plane, const float* bias, uint16_t* out16, size_t o_plane, float* out32, hipStream_t s,
                             float* ws, size_t wsn, const uint16_t* wfrag, int tile, LatIO* lat) {
    if (layer < 0 || layer >= 20 || !conv_is_s3_layer(layer)) return hipErrorInvalidValue;
    const ConvDesc& d = kConvs[layer];
    S3Params p = {};
    p.A = in; p.a_plane = in_plane; p.Wp = wplanes; p.w_plane = w_plane; p.bias = bias;
    p.out16 = out16; p.o_plane = o_plane; p.out32 = out32;
    p.wfrag = wfrag;       // igemm_region.h layers: the packed weight fragments (nullptr: the lean kernels)
    p.tile = tile;
    p.H = h; p.W = w;
    p.Ho = conv_out_dim(h, d.ks, d.stride);
    p.Wo = conv_out_dim(w, d.ks, d.stride);
    p.M = batch * p.Ho * p.Wo;
    p.N = d.cout;
    p.Kp = conv_padded_k(layer);
    const bool o32 = out32 != nullptr;
    switch (layer) {
        case 1:  return run_conv_s3<128, 5, 2, 32, 128, false, NP>(p, s, ws, wsn, lat);
        case 2: case 5: case 11: case 18:
            return o32 ? run_conv_s3<128, 3, 2, 32, 256, true, NP>(p, s, ws, wsn, lat) : run_conv_s3<128, 3, 2, 32, 256, false, NP>(p, s, ws, wsn, lat);
        case 4:  return run_conv_s3<64, 5, 2, 32, 128, false, NP>(p, s, ws, wsn, lat);
        case 6: case 12: case 19:
            return o32 ? run_conv_s3<256, 3, 2, 32, 256, true, NP>(p, s, ws, wsn, lat) : run_conv_s3<256, 3, 2, 32, 256, false, NP>(p, s, ws, wsn, lat);
        case 8:  return run_conv_s3<16, 5, 2, 16, 32, false, NP>(p, s, ws, wsn, lat);
        case 9: case 16: return run_conv_s3<32, 3, 2, 32, 64, false, NP>(p, s, ws, wsn, lat);
        case 10: case 17: return run_conv_s3<64, 3, 2, 32, 128, false, NP>(p, s, ws, wsn, lat);
        case 14: return run_conv_s3<8, 5, 2, 8, 16, false, NP>(p, s, ws, wsn, lat);
        case 15: return run_conv_s3<16, 3, 2, 16, 32, false, NP>(p, s, ws, wsn, lat);
    }
    return hipErrorInvalidValue;
}

// dynamic-LDS limits of the kernels that use more than 64 KB, for this NP; once per device
template <int NP>
hipError_t conv_kernels_init_device_np() {
    hipError_t e = hipSuccess;
    if constexpr (NP != 2) e = hipFuncSetAttribute((const void*)block4_fused_kernel<7, 256, NP, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, B4Cfg<7, 256, NP, true>::LDS_BYTES);
    if constexpr (NP == 2) {
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)block4_fused_kernel<8, 256, NP, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, B4Cfg<8, 256, NP, true>::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)block4_fused_kernel<8, 256, NP, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, B4Cfg<8, 256, NP, true>::LDS_BYTES + B4W_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)igemm_s3_lean8_kernel<HeadLoaderS3, true, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, LEAN8_LDS_BYTES);
#define HNET_PIPE_ATTR(L_, C_, O_) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)igemm_s3_pipe_kernel<L_, C_, O_>, hipFuncAttributeMaxDynamicSharedMemorySize, C_::LDS_BYTES)
        typedef ConvLoaderS3<128, 3, 2, 32> L1283; typedef ConvLoaderS3<64, 5, 2, 32> L645; typedef ConvLoaderS3<64, 3, 2, 32> L643;
        HNET_PIPE_ATTR(L1283, PipeCfg144, false); HNET_PIPE_ATTR(L645, PipeCfg144, false); HNET_PIPE_ATTR(L643, PipeCfg144, false);
#undef HNET_PIPE_ATTR
#define HNET_REGION_ATTR(C_, O_) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)igemm_s3_region_kernel<C_, O_>, hipFuncAttributeMaxDynamicSharedMemorySize, C_::LDS_BYTES)
        HNET_REGION_ATTR(RegionCfg12, false);
        HNET_REGION_ATTR(RegionCfg13, false); HNET_REGION_ATTR(RegionCfg13, true); HNET_REGION_ATTR(RegionCfgT, false); HNET_REGION_ATTR(RegionCfgT, true);
#undef HNET_REGION_ATTR
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)igemm_heads_pipe_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, HeadsPipeCfg::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)heads_fc1_lat_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, HL_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)heads_fc1_lat_kernel<HL_MAXG>, hipFuncAttributeMaxDynamicSharedMemorySize, HL_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)block42_fused_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, B42Cfg::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)block3_fused_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, B3Cfg::LDS_BYTES + B3Cfg::W0_BYTES + B3Cfg::SPARE_BYTES);
    }
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_patch_s2_kernel<3, NP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PatchS2Cfg<3, NP>::LDS_BYTES);
    if constexpr (NP != 1) {
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_patch_s2_kernel<5, NP, true, (NP == 2 ? 5 : 3)>, hipFuncAttributeMaxDynamicSharedMemorySize, PatchS2Cfg<5, NP>::LDS_BYTES);
    } else {
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_patch_s2_kernel<5, NP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PatchS2Cfg<5, NP>::LDS_BYTES);
    }
    if constexpr (NP != 2) {
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_patch32_s2_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, Patch32Cfg<NP>::LDS_BYTES);
    }
    return e;
}

// the explicit instantiations live in kernels_conv.hip (NP = 3), kernels_conv_bf16.hip (NP = 1) and kernels_conv_f16x2.hip (NP = 2)
#define HNET_S3_DISPATCH_INSTANCES(KW, NP)                                                                                               \
    KW template hipError_t launch_block4_fused_np<NP>(const void*, size_t, const void*, const float*, const void*, const float*,         \
                                                      uint16_t*, size_t, int, hipStream_t, int, const B4Warp*);                          \
    KW template hipError_t launch_block42_fused_np<NP>(const uint16_t*, size_t, const void*, const float*, const void*, const float*, uint16_t*, size_t, int, hipStream_t); \
    KW template hipError_t launch_block3_fused_np<NP>(const float*, const void*, const float*, const void*, const float*, uint16_t*, size_t, int, hipStream_t); \
    KW template hipError_t launch_conv_first_s3_np<NP>(const float*, const void*, const float*, uint16_t*, size_t, int, int, int, hipStream_t); \
    KW template hipError_t launch_conv_first_s2_np<NP>(int, const float*, const void*, const float*, uint16_t*, size_t, int, hipStream_t); \
    KW template hipError_t launch_conv_patch_np<NP>(int, const uint16_t*, size_t, int, int, int, const void*, const float*, uint16_t*,   \
                                                    size_t, hipStream_t, bool, int);                                                           \
    KW template hipError_t launch_heads_fc1_s3_np<NP>(const float*, int, int, int, float, uint64_t, uint64_t, const uint16_t*,           \
                                                      const float*, float*, uint16_t*, size_t, uint8_t*, hipStream_t, float*, size_t,    \
                                                      const uint64_t*, int, LatIO*);                                                     \
    KW template hipError_t launch_conv_s3_np<NP>(int, const uint16_t*, size_t, int, int, int, const uint16_t*, size_t, const float*,     \
                                                 uint16_t*, size_t, float*, hipStream_t, float*, size_t, const uint16_t*, int, LatIO*);  \
    KW template hipError_t conv_kernels_init_device_np<NP>();

}  // namespace hnet
