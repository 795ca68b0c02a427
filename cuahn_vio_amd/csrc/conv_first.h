// conv_first.h — direct 7x7 / stride-1 / Cin = 2 convolution on fp32 MFMA for the two full-resolution first
// layers (block_4_0: 2->8 @224x320, block_3_0: 2->16 @112x160; reference model_to_trace.py:107,210 via conv() :7-15).
//
// Why not the generic implicit GEMM: with Cin = 2 its im2col staging is address-arithmetic bound (12 % of the MFMA
// peak measured) and Cout = 8 wastes half of a 16-wide MFMA.  Here instead:
//   * the workgroup stages its input patch (tile + 3-pixel halo) ONCE in LDS, NHWC, zero filled outside the image;
//   * GEMM view: M = pairs of horizontally adjacent output pixels, N = (pixel-in-pair dx, cout) = 2*Cout,
//     K = (kh, kw', ci) with kw' in 0..7: the 8-tap window both pixels of a pair need.  W'[kh][kw'][ci][(dx,co)] =
//     W[co][ci][kh][kw'-dx] (zero outside 0..6).  N is exactly 16 (Cout 8, v_mfma_f32_16x16x4_f32) or 32 (Cout 16,
//     v_mfma_f32_32x32x2_f32) and 7 of 8 taps are useful: 87.5 % MFMA efficiency instead of 43.75 %;
//   * for a fixed kh the 16 K-values of a pair are 16 CONTIGUOUS floats of the patch row: one ds_read_b128 per lane
//     feeds 4 MFMAs (K-permutation as in igemm.h: lane group g reads floats 4g..4g+3, element i goes to step i);
//   * the whole W' (28 resp. 56 fragment registers per lane) lives in VGPRs for the lifetime of the workgroup.
#pragma once
#include <hip/hip_runtime.h>
#include "igemm.h"
#include "igemm_s3.h"

namespace hnet {

template <int COUT> struct FirstCfg;
template <> struct FirstCfg<8> {
    static constexpr int TH = 16, TW = 64;          // output tile
    static constexpr int NFRAG = 28;                // 7 rows x 4 k-steps (16x16x4)
};
template <> struct FirstCfg<16> {
    static constexpr int TH = 16, TW = 32;
    static constexpr int NFRAG = 56;                // 7 rows x 8 k-steps (32x32x2)
};

// wfrag: [NFRAG][64] floats, fragment t of lane l (host-packed, see pack_first_weights in hnet_capi.hip)
// out16 != nullptr: the output is written as three bf16 planes (S3, igemm_s3.h) instead of fp32
template <int COUT>
__global__ __launch_bounds__(256) void conv7_c2_s1_kernel(const float* __restrict__ in, const float* __restrict__ wfrag,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          uint16_t* __restrict__ out16, size_t o_plane,
                                                          int H, int W, int tiles_x, int tiles_y) {
    typedef FirstCfg<COUT> C;
    constexpr int TH = C::TH, TW = C::TW, PH = TH + 6, PW = TW + 8;   // patch: 3-px halo + 2 columns for kw' = 7 / padding
    constexpr int PROW = PW * 2;                                       // floats per patch row
    __shared__ __attribute__((aligned(16))) float patch[PH * PROW];
    // wave-private staging of one split output tile: [plane][rows][cols] bf16 (16x16 for Cout 8, 32x32 for Cout 16)
    constexpr int ST_ROWS = COUT == 8 ? 16 : 32;
    __shared__ __attribute__((aligned(16))) uint16_t stage[4 * 3 * ST_ROWS * ST_ROWS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- weights fragments -> registers (coalesced: 64 consecutive floats per fragment)
    float wreg[C::NFRAG];
#pragma unroll
    for (int t = 0; t < C::NFRAG; t++) wreg[t] = wfrag[t * 64 + lane];

    // ---- stage the patch: unconditional float2 loads from a clamped address, zero selected afterwards
    const float* inb = in + (size_t)b * H * W * 2;
    for (int i = tid; i < PH * PW; i += 256) {
        const int pr = i / PW, pc = i - pr * PW;
        const int iy = y0 - 3 + pr, ix = x0 - 3 + pc;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const float2 v = *reinterpret_cast<const float2*>(inb + (ok ? ((size_t)iy * W + ix) * 2 : 0));
        *reinterpret_cast<float2*>(&patch[pr * PROW + pc * 2]) = ok ? v : make_float2(0.f, 0.f);
    }
    __syncthreads();

    if constexpr (COUT == 8) {
        // M-tile = 16 pixel pairs of one output row; 2 M-tiles per row, TH rows -> 32 M-tiles, 8 per wave
        const int m = lane & 15, g = lane >> 4;
        const float bv = bias[lane & 7];
#pragma unroll 2
        for (int j = 0; j < (TH * (TW / 32)) / 4; j++) {
            const int mt = wave + 4 * j;
            const int row = mt / (TW / 32), half = mt % (TW / 32);
            const int pm = half * 16 + m;                       // pixel-pair index within the tile row
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 7; kh++) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&patch[(row + kh) * PROW + pm * 4 + 4 * g]);
#pragma unroll
                for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], wreg[kh * 4 + e], acc, 0, 0, 0);
            }
            // D: col n = lane&15 = (dx, co); row = 4*(lane>>4) + r = pixel pair
            const int y = y0 + row;
            if (out16 == nullptr) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int pp = half * 16 + 4 * g + r;
                    const int x = x0 + 2 * pp + (m >> 3);
                    if (y < H && x < W) {
                        const float v = acc[r] + bv;
                        out[(((size_t)b * H + y) * W + x) * 8 + (m & 7)] = v > 0.f ? v : v * 0.1f;
                    }
                }
            } else {
                // tile = 16 pixel pairs x 16 (dx,co): per plane 16 rows of 32 bytes = 2 pieces of 16 B (one pixel each)
                uint16_t* st = stage + wave * (3 * 16 * 16);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = acc[r] + bv;
                    v = v > 0.f ? v : v * 0.1f;
                    uint16_t sa, sb, sc;
                    split3(v, sa, sb, sc);
                    const int trow = 4 * g + r;
                    st[(0 * 16 + trow) * 16 + m] = sa;
                    st[(1 * 16 + trow) * 16 + m] = sb;
                    st[(2 * 16 + trow) * 16 + m] = sc;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int piece = q * 64 + lane;             // 96 pieces: [plane][row][dx]
                    if (piece < 96) {
                        const int pl = piece >> 5, rem = piece & 31, trow = rem >> 1, dx = rem & 1;
                        const int x = x0 + 2 * (half * 16 + trow) + dx;
                        const u32x4 v = *reinterpret_cast<const u32x4*>(&st[(pl * 16 + trow) * 16 + dx * 8]);
                        if (y < H && x < W) *reinterpret_cast<u32x4*>(out16 + pl * o_plane + (((size_t)b * H + y) * W + x) * 8) = v;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
            }
        }
    } else {
        // COUT == 16: M-tile = 2 output rows x 16 pixel pairs (32 px wide tile); TH/2 = 8 M-tiles, 2 per wave
        const int m = lane & 31, h = lane >> 5;
        const int mrow = m >> 4, mp = m & 15;
        const float bv = bias[lane & 15];
#pragma unroll 1
        for (int j = 0; j < (TH / 2) / 4; j++) {
            const int mt = wave + 4 * j;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
            for (int kh = 0; kh < 7; kh++) {
                const float* prow = &patch[(mt * 2 + mrow + kh) * PROW + mp * 4 + 4 * h];
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(prow);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(prow + 8);
#pragma unroll
                for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], wreg[kh * 8 + e], acc, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], wreg[kh * 8 + 4 + e], acc, 0, 0, 0);
            }
            // D: col n = lane&31 = (dx, co); row = (r&3) + 8*(r>>2) + 4*(lane>>5) = (row-in-pair-of-rows, pixel pair)
            if (out16 == nullptr) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int mr = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int y = y0 + mt * 2 + (mr >> 4);
                    const int x = x0 + 2 * (mr & 15) + (m >> 4);
                    if (y < H && x < W) {
                        const float v = acc[r] + bv;
                        out[(((size_t)b * H + y) * W + x) * 16 + (m & 15)] = v > 0.f ? v : v * 0.1f;
                    }
                }
            } else {
                // tile = 32 (row-of-pair, pixel pair) x 32 (dx,co): per plane 32 rows of 64 bytes = 4 pieces (dx, co-half)
                uint16_t* st = stage + wave * (3 * 32 * 32);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int mr = (r & 3) + 8 * (r >> 2) + 4 * h;
                    float v = acc[r] + bv;
                    v = v > 0.f ? v : v * 0.1f;
                    uint16_t sa, sb, sc;
                    split3(v, sa, sb, sc);
                    st[(0 * 32 + mr) * 32 + m] = sa;
                    st[(1 * 32 + mr) * 32 + m] = sb;
                    st[(2 * 32 + mr) * 32 + m] = sc;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const int piece = q * 64 + lane;             // 384 pieces: [plane][32 rows][4 chunks]
                    const int pl = piece >> 7, rem = piece & 127, mr = rem >> 2, ch = rem & 3;
                    const int y = y0 + mt * 2 + (mr >> 4);
                    const int x = x0 + 2 * (mr & 15) + (ch >> 1);
                    const u32x4 v = *reinterpret_cast<const u32x4*>(&st[(pl * 32 + mr) * 32 + ch * 8]);
                    if (y < H && x < W)
                        *reinterpret_cast<u32x4*>(out16 + pl * o_plane + (((size_t)b * H + y) * W + x) * 16 + (ch & 1) * 8) = v;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// block_3_0 (2 -> 16 @112x160) on the bf16 matrix cores, split-bf16 x3 (igemm_s3.h): the same pixel-pair GEMM, but the
// patch is split into three bf16 planes while it is staged and a kernel row is ONE v_mfma_f32_32x32x16_bf16 step
// (K = 8 taps x 2 channels) instead of eight 32x32x2 fp32 steps: 42 bf16 MFMAs per 64 pixels against 56 fp32 ones at a
// sixteenth of the rate.  The MFMA is issued with the weights as A operand (transposed tile, conv_b4_fused.h): a lane
// holds 4 x 4 consecutive channels of its pixel pair and writes them as 8-byte pieces of the S3 planes.
// wfrag: [7 kernel rows][3 planes][64 lanes] x 16 B; lane (n = l&31 = (dx, co), hh = l>>5) holds kk = 8hh .. 8hh+7 of
// W'[kh][kk = 2 kw' + ci][n] = W[co][ci][kh][kw' - dx] (pack in hnet_capi.hip).
// ---------------------------------------------------------------------------------------------
// NP = number of bf16 planes (3 = split-bf16, 1 = plain bf16 operands)
template <int NP>
__global__ __launch_bounds__(256) void conv7_c2_s1_s3_kernel(const float* __restrict__ in, const u32x4* __restrict__ wfrag,
                                                             const float* __restrict__ bias, uint16_t* __restrict__ out16, size_t o_plane,
                                                             int H, int W, int tiles_x, int tiles_y, int n_tiles) {
    constexpr int TH = 16, TW = 32, PH = TH + 6, PW = TW + 8, PROW = PW * 2, PPLANE = PH * PROW;
    __shared__ __attribute__((aligned(16))) uint16_t patch[NP * PPLANE];
    __shared__ __attribute__((aligned(16))) uint16_t stage[4 * NP * 256 * 4];      // per wave and plane: 256 units of 8 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // persistent workgroups (round 2): the 84 weight registers are loaded once, not once per 16 x 32 tile (86 KB per tile from L2), and
    // the next tile's patch is in flight while the current one is computed
    bf16x8 wv[7][3];
#pragma unroll
    for (int kh = 0; kh < 7; kh++)
#pragma unroll
        for (int pl = 0; pl < s3_wplanes<NP>; pl++) wv[kh][pl] = __builtin_bit_cast(bf16x8, wfrag[(kh * 3 + pl) * 64 + lane]);

    constexpr int PPT = (PH * PW + 255) / 256;
    float2 px[PPT];
    uint32_t okbits = 0;
    // unconditional float2 loads from a clamped address; zero is selected when the registers are consumed
    auto patch_load = [&](int t) {
        int bid = s3p::xcd_tile(t, n_tiles, gridDim.x);
        const int tx = bid % tiles_x; bid /= tiles_x;
        const int ty = bid % tiles_y;
        const int b = bid / tiles_y;
        const float* inb = in + (size_t)b * H * W * 2;
        okbits = 0;
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = min(tid + q * 256, PH * PW - 1);
            const int pr = i / PW, pc = i - pr * PW;
            const int iy = ty * TH - 3 + pr, ix = tx * TW - 3 + pc;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            px[q] = *reinterpret_cast<const float2*>(inb + (ok ? ((size_t)iy * W + ix) * 2 : 0));
            okbits |= ok ? (1u << q) : 0u;
        }
    };
    if ((int)blockIdx.x < n_tiles) patch_load(blockIdx.x);

    const int pcol = lane & 31, hh = lane >> 5;
    const int prow = pcol >> 4, pair = pcol & 15;
    int hi4 = 4;                                                // opaque: two ds_read_b64 (2 + 2 LDS cycles) instead of one ds_read2_b64 (8)
    asm volatile("" : "+v"(hi4));
    float bv[4][4];                                             // D row (r&3) + 8(r>>2) + 4hh = n = (dx, co): group q = r>>2
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < 4; i++) bv[q][i] = bias[(8 * q + 4 * hh + i) & 15] * s3_acc_scale<NP>;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    int bid = s3p::xcd_tile(tile, n_tiles, gridDim.x);
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- the prefetched patch -> bf16 planes in LDS
    __syncthreads();                                            // the previous tile's MFMA phase is done with the patch
#pragma unroll
    for (int q = 0; q < PPT; q++) {
        const int i = tid + q * 256;
        if (i < PH * PW) {
            const bool ok = (okbits >> q) & 1u;
            uint32_t pk[3];
            s3p::split_pair<NP>(ok ? px[q].x : 0.f, ok ? px[q].y : 0.f, pk);
            const int e = i * 2;                                // [row][column][2 ch], rows of PROW elements: i = pr*PW + pc
#pragma unroll
            for (int pl = 0; pl < NP; pl++) *reinterpret_cast<uint32_t*>(&patch[pl * PPLANE + e]) = pk[pl];
        }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < n_tiles) patch_load(tile + gridDim.x);

    // M-tile = 2 output rows x 16 pixel pairs; lane column = (row-of-pair, pixel pair), lane half hh = taps 4hh..4hh+3
#pragma unroll 1
    for (int j = 0; j < (TH / 2) / 4; j++) {
        const int mt = wave + 4 * j;
        f32x16 acc;                                            // bias = initial accumulator
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = bv[r >> 2][r & 3];
        const int abase = (mt * 2 + prow) * PROW + pair * 4 + 8 * hh;
#pragma unroll
        for (int kh = 0; kh < 7; kh++) {
            bf16x8 a[3];
#pragma unroll
            for (int pl = 0; pl < NP; pl++) {                   // 8 bf16 = 4 taps x 2 ch, 8-byte aligned
                const uint16_t* src = &patch[pl * PPLANE + abase + kh * PROW];
                typedef short bf16x4_t __attribute__((ext_vector_type(4)));
                const bf16x4_t lo = *reinterpret_cast<const bf16x4_t*>(src);
                const bf16x4_t hi = *reinterpret_cast<const bf16x4_t*>(src + hi4);
                a[pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            acc = s3_mfma32<NP>(acc, wv[kh], a);
        }
        // D row 4 (2 q' + hh) + i of group q = (dx, co half c): a lane holds channels 8 c + 4 hh .. + 3 of pixel 2 pair + dx.  The wave's
        // 2 rows x 32 pixels x 16 channels are staged per plane in LDS as 8-byte units and read back so that a lane stores 16 bytes and
        // one store instruction covers 1 KiB of contiguous global memory (direct 8-byte stores at 64-byte stride were measured 35 %
        // slower: 4x the cache lines per instruction).  Unit index U = ((c 2 + hh) 2 + prow) 32 + 16 (prow ^ dx) + pair: the 32 lanes of a
        // write (fixed dx, c, hh; all prow, pair) and of a read (fixed prow, hh, c; all pair, dx) each fall on 32 different units mod 32
        // = all 64 banks once.  (Round 1 used the plain [row][x][16 ch] order: pairs 0, 4, 8, 12 on the same bank, 58 % of the LDS cycles.)
        uint16_t* st = stage + wave * (NP * 256 * 4);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int dx = q >> 1, c = q & 1;
            uint32_t pa[3], pb[3];
            s3p::act_split<NP>(acc[4 * q], acc[4 * q + 1], pa);
            s3p::act_split<NP>(acc[4 * q + 2], acc[4 * q + 3], pb);
            const int u = ((c * 2 + hh) * 2 + prow) * 32 + 16 * (prow ^ dx) + pair;
#pragma unroll
            for (int pl = 0; pl < NP; pl++) *reinterpret_cast<uint2*>(&st[(pl * 256 + u) * 4]) = make_uint2(pa[pl], pb[pl]);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): the wave's own LDS writes have landed
        {
            const int rp = lane & 15, rdx = (lane >> 4) & 1, rc = lane >> 5;      // this lane's piece of a row: pixel 2 rp + rdx, channel half rc
            const int x = x0 + 2 * rp + rdx;
#pragma unroll
            for (int pr = 0; pr < 2; pr++) {
                const int y = y0 + mt * 2 + pr;
                const int u0 = ((rc * 2) * 2 + pr) * 32 + 16 * (pr ^ rdx) + rp;
#pragma unroll
                for (int pl = 0; pl < NP; pl++) {
                    const uint2 lo = *reinterpret_cast<const uint2*>(&st[(pl * 256 + u0) * 4]);
                    const uint2 hi = *reinterpret_cast<const uint2*>(&st[(pl * 256 + u0 + 64) * 4]);
                    if (y < H && x < W)
                        *reinterpret_cast<u32x4*>(out16 + pl * o_plane + (((size_t)b * H + y) * W + x) * 16 + rc * 8) = u32x4{lo.x, lo.y, hi.x, hi.y};
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    }   // persistent tile loop
}


// ---------------------------------------------------------------------------------------------
// block_1_1 (2 -> 128 @28x40 -> 14x20) and block_2_1 (2 -> 64 @56x80 -> 28x40): 7x7, stride 2, Cin 2, on the bf16 matrix
// cores (round 2).  In round 1 these two layers were the last ones on the fp32 MFMA (1/16 of the bf16 rate): 0.9 % of the
// network's MACs took 4.8 % of the step (block_1_1 0.046 ms, block_2_1 0.082 ms, 14.9 VALU instructions per MFMA in the
// generic im2col GEMM).  Here a workgroup stages the input band of TH output rows (all columns) ONCE into LDS as bf16 planes
// and every wave multiplies all the band's pixels with ITS slice of output channels, whose weights stay in registers:
//   M-tile = 16 consecutive output pixels of the band (row major), lane column m = pixel;
//   K = (kh, kw' 0..7, ci): a kernel row is 8 taps x 2 channels = 16 values (tap 7 has zero weights), two rows per
//   32-deep step, four steps (row 7 zero): lane group g reads row 2 st + (g>>1), taps 4 (g&1) .. + 3 = 16 contiguous bytes
//   of the patch (8-byte aligned: the first tap of output column ox is input column 2 ox);
//   N = 16 output channels per MFMA, weights as A operand (transposed tile: a lane holds 4 channels of one pixel).
// wfrag: [COUT / 16 n-tiles][4 steps][3 planes][64 lanes] x 16 B (pack in hnet_capi.hip).  out16: S3 planes [B][HO][WO][COUT].
// ---------------------------------------------------------------------------------------------
template <int COUT, int HO, int WO, int TH, int NP>
__global__ __launch_bounds__(256) void conv7_c2_s2_s3_kernel(const float* __restrict__ in, const u32x4* __restrict__ wfrag,
                                                             const float* __restrict__ bias, uint16_t* __restrict__ out16, size_t o_plane) {
    constexpr int H = 2 * HO, W = 2 * WO;
    constexpr int PH = 2 * TH + 5, PW = 2 * WO + 6, PPLANE = PH * PW * 2;      // patch rows x columns (x 2 channels, bf16 elements)
    constexpr int BANDS = HO / TH, NPIX_T = TH * WO, M_TILES = (NPIX_T + 15) / 16;
    constexpr int NT = COUT / 16, NTW = NT / 4;                                 // n-tiles per wave
    static_assert(HO % TH == 0 && NT % 4 == 0, "band / channel split");
    __shared__ __attribute__((aligned(16))) uint16_t patch[NP * PPLANE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int band = blockIdx.x % BANDS, b = blockIdx.x / BANDS;
    const int oy0 = band * TH;

    bf16x8 wv[NTW][4][3];
    float bv[NTW][4];
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const int nt = wave + 4 * t;
#pragma unroll
        for (int st = 0; st < 4; st++)
#pragma unroll
            for (int pl = 0; pl < s3_wplanes<NP>; pl++) wv[t][st][pl] = __builtin_bit_cast(bf16x8, wfrag[((nt * 4 + st) * 3 + pl) * 64 + lane]);
#pragma unroll
        for (int r = 0; r < 4; r++) bv[t][r] = bias[nt * 16 + 4 * g + r] * s3_acc_scale<NP>;
    }

    // ---- stage the band: input rows 2 oy0 - 3 .. + PH, columns -3 .. + PW, zero outside the image, split into planes
    const float* inb = in + (size_t)b * H * W * 2;
    {
        // all of a thread's pixels in flight before the first one is consumed (round 4: written as one loop, every load was followed by vmcnt(0) - up to seven
        // dependent global round trips per workgroup, most of the kernel's time)
        constexpr int NPX = (PH * PW + 255) / 256;
        float2 f[NPX];
        uint32_t okb = 0;
#pragma unroll
        for (int q = 0; q < NPX; q++) {
            const int i = min(tid + 256 * q, PH * PW - 1);
            const int pr = i / PW, pc = i - pr * PW;
            const int iy = 2 * oy0 - 3 + pr, ix = pc - 3;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            f[q] = *reinterpret_cast<const float2*>(inb + (ok ? ((size_t)iy * W + ix) * 2 : 0));
            okb |= ok ? (1u << q) : 0u;
        }
#pragma unroll
        for (int q = 0; q < NPX; q++) {
            const int i = tid + 256 * q;
            if (i < PH * PW) {
                const bool ok = (okb >> q) & 1u;
                uint16_t a[3], c[3];
                s3p::split1<NP>(ok ? f[q].x : 0.f, a[0], a[1], a[2]);
                s3p::split1<NP>(ok ? f[q].y : 0.f, c[0], c[1], c[2]);
#pragma unroll
                for (int pl = 0; pl < NP; pl++) *reinterpret_cast<uint32_t*>(&patch[pl * PPLANE + i * 2]) = (uint32_t)a[pl] | ((uint32_t)c[pl] << 16);
            }
        }
    }
    __syncthreads();

    typedef short bf16x4_t __attribute__((ext_vector_type(4)));
#pragma unroll 1
    for (int mt = 0; mt < M_TILES; mt++) {
        const int p = min(mt * 16 + m, NPIX_T - 1);                            // pixels beyond the band are computed on a clamped address, not stored
        const int oy = p / WO, ox = p - oy * WO;
        const int abase = ((2 * oy + (g >> 1)) * PW + 2 * ox + 4 * (g & 1)) * 2;
        bf16x8 a[4][3];
#pragma unroll
        for (int st = 0; st < 4; st++)
#pragma unroll
            for (int pl = 0; pl < NP; pl++) {
                const uint16_t* src = &patch[pl * PPLANE + abase + min(2 * st, 6 - (g >> 1)) * PW * 2];     // kernel row 7 has zero weights
                const bf16x4_t lo = *reinterpret_cast<const bf16x4_t*>(src);
                const bf16x4_t hi = *reinterpret_cast<const bf16x4_t*>(src + 4);
                a[st][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        const size_t opix = ((size_t)b * HO + oy0 + oy) * WO + ox;
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            f32x4 acc = {bv[t][0], bv[t][1], bv[t][2], bv[t][3]};
#pragma unroll
            for (int st = 0; st < 4; st++) {
                acc = s3_mfma16<NP>(acc, wv[t][st], a[st]);
            }
            // D (transposed): row 4g + r = channel 4g + r of n-tile wave + 4t, column m = pixel: 8 bytes per lane and plane.  The packed LeakyReLU + split of the
            // GEMM epilogues (round 4; one value at a time it was twice the vector instructions - and this kernel's vector issue is as long as its stores)
            uint32_t pa[3], pb[3];
            s3p::act_split<NP>(acc[0], acc[1], pa);
            s3p::act_split<NP>(acc[2], acc[3], pb);
            {   // (no branch - it would end the scheduling region: lanes beyond the band repeat its last pixel, same address, same value)
                uint16_t* o = out16 + opix * COUT + (wave + 4 * t) * 16 + 4 * g;
#pragma unroll
                for (int pl = 0; pl < NP; pl++) *reinterpret_cast<uint2*>(o + pl * o_plane) = make_uint2(pa[pl], pb[pl]);
            }
        }
        // all of the M-tile's fragment reads first (the two 8-byte halves of a fragment are one ds_read2_b64), then its MFMAs: left to itself the scheduler put
        // every read in front of its MFMA with a wait - eight LDS round trips per M-tile
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * NP, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NTW * 4 * (NP == 3 ? 6 : NP == 2 ? 3 : 1), 0);
    }
}

}  // namespace hnet
