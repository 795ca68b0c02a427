// conv_b42_fused.h — block_4_2 (3x3 s2, 16 -> 32 @112x160 -> 56x80) and block_4_3 (3x3 s2, 32 -> 64 -> 28x40) in ONE kernel
// (reference model_to_trace.py:212-213 via conv() :7-15).  Round 3; fp16-plane arithmetic (HNET_PREC_F16X2) only.
//
// Unfused, the two layers are the patch kernels of conv_patch_s2.h: 0.101 + 0.056 ms per 256 pairs at 0.16 - 0.28 matrix-pipe busy - they
// wait for memory: block_4_2 reads 294 MB and writes 147 MB that block_4_3 reads straight back.  Here a workgroup owns a 4 x 8 tile of
// block_4_3 outputs (35 tiles per pair):
//   phase 0  the 19 x 35-pixel patch of block_4_1's output the tile needs (two fp16 planes, 16 channels = 32 bytes per pixel and plane) is
//            copied into LDS as [plane][row][column parity][column / 2][16 ch], zero outside the image (= block_4_2's zero padding)
//   phase 1  block_4_2 on the 9 x 17 region: transposed 16x16x32 tiles (weights as the A operand, in VGPRs), an M-tile = 16 pixels of one
//            region row (+ one M-tile for column 16 of all rows), a 32-deep K step = two filter taps x 16 channels (5 steps for 9 taps);
//            bias + LeakyReLU, zero outside the 56 x 80 image (= block_4_3's zero padding), split into planes, written to LDS as 64-byte
//            pixels [plane][row][parity][column / 2][32 ch] (rows padded by 16 bytes: see the bank note below)
//   phase 2  block_4_3 from that image: M-tile = 2 output rows x 8 pixels, a K step = ONE tap x 32 channels (9 steps), wave w owns output
//            channels 16 w .. + 15 and both M-tiles; output as fp16 planes.
// The 32-channel intermediate (147 MB per 256 pairs) never touches HBM.  Arithmetic: the two-plane / two-accumulator form of igemm_s3.h
// (w = W0 + W1 / 4096, hi += W0 A0, lo += W1 A0 + W0 A1, result hi + lo / 4096), as conv_b3_fused.h.
//
// Bank notes (a ds_read_b128 is served in four passes of 16 lanes: pixels {0-3, 12-15} of lane group g with pixels {4-11} of group g + 1).
// Phase 1 reads 32-byte pixels: group g takes even 16-byte slots, group g + 1 (other channel half, or the next tap) odd ones - conflict free as
// in conv_b3_fused.h.  The image has 64-byte pixels; written plainly, the 16 lanes of a phase-1 store group (16 consecutive columns, the same
// 8 bytes of each pixel) fall on TWO bank positions (8-way: 45 % of the kernel's LDS cycles were conflicts).  The four 16-byte chunks of a
// pixel are therefore ROTATED by (x / 2 >> 1) & 3: physical chunk = (channel chunk + rot) & 3.  Stores are then 2-way, and the phase-2 reads
// (lane group g = channels 8 g .. + 7 of one tap, lanes = 2 rows x 8 columns) stay conflict free (checked with a bank model of the four
// 16-lane passes over all nine taps; without the rotation they needed a 16-byte row pad instead).
#pragma once
#include <hip/hip_runtime.h>
#include "igemm_s3.h"
#include "kernels.h"

namespace hnet {

struct B42Cfg {
    static constexpr int TH = 4, TW = 8, THREADS = 256;
    static constexpr int H1 = 112, W1 = 160, H2 = 56, W2 = 80, H3 = 28, W3 = 40, C1 = 16, C2 = 32, C3 = 64;
    static constexpr int RH = 2 * TH + 1, RW = 2 * TW + 1;       // block_4_2 region of a tile: 9 x 17
    static constexpr int PH = 2 * RH + 1, PW = 2 * RW + 1;       // block_4_1 patch: 19 x 35
    static constexpr int XHP = 18;                               // patch pixels per (row, parity)
    static constexpr int PROWB = 2 * XHP * 32 + 16, PPLANEB = PH * PROWB;   // + 16: the rows of the column-16 M-tile (one lane per row) spread over the banks
    static constexpr int XHR = 9;                                // region pixels per (row, parity)
    static constexpr int IROWB = 2 * XHR * 64, IPLANEB = RH * IROWB;
    static constexpr int LDS_BYTES = 2 * (PPLANEB + IPLANEB);
    static constexpr int TILES_X = W3 / TW, TILES_Y = H3 / TH;   // 5 x 7 tiles per pair
    static constexpr int N_MT1 = RH + 1;                         // phase-1 M-tiles: one per region row (columns 0..15) + one for column 16
    static constexpr int NST1 = 5, NST2 = 9;
    static_assert(W3 % TW == 0 && H3 % TH == 0, "tiles cover the 28 x 40 output exactly");
};

// in16:   block_4_1 output, fp16 planes [2][B][112][160][16] (i_plane elements per plane)
// w2frag: block_4_2, [2 n-tiles][5 steps][2 planes][64 lanes] x 16 B: lane (i = l & 15, g = l >> 4): channel 16 nt + i, tap 2 st + (g >> 1), ci 8 (g & 1) + j
// w3frag: block_4_3, [4 n-tiles][9 steps][2 planes][64 lanes] x 16 B: lane (i, g): channel 16 nt + i, tap st, ci 8 g + j
template <int NP>
__global__ __launch_bounds__(256, 2) void block42_fused_kernel(const uint16_t* __restrict__ in16, size_t i_plane, const u32x4* __restrict__ w2frag,
                                                               const float* __restrict__ bias2, const u32x4* __restrict__ w3frag,
                                                               const float* __restrict__ bias3, uint16_t* __restrict__ out16, size_t o_plane,
                                                               int n_tiles) {
    static_assert(NP == 2, "two fp16 planes (HNET_PREC_F16X2): the other modes run the two layers unfused");
    typedef B42Cfg C;
    constexpr int H1 = C::H1, W1 = C::W1, H2 = C::H2, W2 = C::W2, H3 = C::H3, W3 = C::W3, TH = C::TH, TW = C::TW, RH = C::RH, PH = C::PH;
    constexpr int XHP = C::XHP, PROWB = C::PROWB, PPLANEB = C::PPLANEB, XHR = C::XHR, IROWB = C::IROWB, IPLANEB = C::IPLANEB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char* const patch = lds_raw;                               // [2 planes][PH][2][XHP] x 32 B
    unsigned char* const img = lds_raw + 2 * PPLANEB;                   // [2 planes][RH] x IROWB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, g = lane >> 4;

    // ---- weights -> registers, once per (persistent) workgroup: phase 1 n-tile = wave & 1, phase 2 n-tile = wave
    const int nt1 = wave & 1;
    f16x8 w2[C::NST1][2], w3[C::NST2][2];
#pragma unroll
    for (int st = 0; st < C::NST1; st++)
#pragma unroll
        for (int pl = 0; pl < 2; pl++) w2[st][pl] = __builtin_bit_cast(f16x8, w2frag[((nt1 * C::NST1 + st) * 2 + pl) * 64 + lane]);
#pragma unroll
    for (int st = 0; st < C::NST2; st++)
#pragma unroll
        for (int pl = 0; pl < 2; pl++) w3[st][pl] = __builtin_bit_cast(f16x8, w3frag[((wave * C::NST2 + st) * 2 + pl) * 64 + lane]);
    float bv2[4], bv3[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { bv2[r] = bias2[16 * nt1 + 4 * g + r]; bv3[r] = bias3[16 * wave + 4 * g + r]; }

    // phase-1 tap offsets: step st, tap t = 2 st + (g >> 1) = (kh, kw) of the 3 x 3 window: patch pixel (2 row + kh, 2 col + kw) -> parity kw & 1,
    // x/2 = col + (kw >> 1); two compile-time constants per step, selected by g >> 1 (t = 9 has zero weights: any valid address)
    auto p1tap = [](int t) constexpr { const int tt = t < 8 ? t : 8; const int kh = tt / 3, kw = tt - 3 * kh; return kh * C::PROWB + ((kw & 1) * C::XHP + (kw >> 1)) * 32; };
    auto p2tap = [](int t) constexpr { const int kh = t / 3, kw = t - 3 * kh; return kh * C::IROWB + ((kw & 1) * C::XHR + (kw >> 1)) * 64; };
    const bool ghi = (g >> 1) != 0;
    const uint32_t p1lane = (uint32_t)(m * 32 + 16 * (g & 1));          // regular M-tile: column m, channel half g & 1
    // phase 2: lane m = (output row m >> 3 of the M-tile, column m & 7), lane group g = channels 8 g .. + 7 = chunk (g + rot) & 3 of its pixel;
    // the pixel is x / 2 = (m & 7) for the taps kw = 0, 1 and (m & 7) + 1 for kw = 2: two lane offsets
    const uint32_t p2row = (uint32_t)((2 * (m >> 3)) * IROWB + (m & 7) * 64);
    const uint32_t p2lane0 = p2row + 16u * ((g + (((m & 7) >> 1) & 3)) & 3), p2lane1 = p2row + 16u * ((g + ((((m & 7) + 1) >> 1) & 3)) & 3);

    auto tile_origin = [&](int t, int& b, int& ty, int& tx) {
        int bid = s3p::xcd_tile(t, n_tiles, gridDim.x);
        tx = bid % C::TILES_X; bid /= C::TILES_X;
        ty = bid % C::TILES_Y;
        b = bid / C::TILES_Y;
    };

    // ---- patch prefetch (registers).  The patch rows are contiguous in memory (35 pixels x 32 bytes per plane): chunk q of a thread is
    // (row-plane rp = idx / 70, 16-byte chunk cc = idx % 70 of that row), idx = tid + 256 q - consecutive lanes read consecutive chunks; the
    // LDS side scatters them to the parity-split layout (8-lane write groups stay conflict free: pixels 0..3 -> slots 0, 36, 2, 38 + halves)
    constexpr int CPR = 2 * C::PW, NCH = 2 * PH * CPR, PPT = (NCH + 255) / 256;      // 70 chunks per row, 2660 in all, 11 per thread
    u32x4 pre[PPT];
    auto patch_load = [&](int t) {
        int b, ty, tx;
        tile_origin(t, b, ty, tx);
        const int Py0 = 8 * ty * TH / 2 - 3, Px0 = 8 * tx * TW / 2 - 3;      // = 2 (2 ty0 - 1) - 1, 2 (2 tx0 - 1) - 1
        const uint16_t* inb = in16 + (size_t)b * H1 * W1 * C::C1;
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int idx = min(tid + 256 * q, NCH - 1);
            const int rp = idx / CPR, cc = idx - rp * CPR;
            const int pl = rp >= PH ? 1 : 0, prow = rp - pl * PH;
            const int iy = Py0 + prow, ix = Px0 + (cc >> 1);
            const bool ok = (unsigned)iy < (unsigned)H1 && (unsigned)ix < (unsigned)W1;
            // unconditional load from a clamped address (zero is selected when the registers are consumed)
            const size_t e = ok ? ((size_t)iy * W1 + ix) * C::C1 + 8 * (cc & 1) : 0;
            pre[q] = *reinterpret_cast<const u32x4*>(inb + (size_t)pl * i_plane + e);
            if (!ok) pre[q] = u32x4{0u, 0u, 0u, 0u};
        }
    };
    if ((int)blockIdx.x < n_tiles) patch_load(blockIdx.x);

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int b, ty, tx;
        tile_origin(tile, b, ty, tx);
        const int ty0 = ty * TH, tx0 = tx * TW;
        const int Ry0 = 2 * ty0 - 1, Rx0 = 2 * tx0 - 1;                 // block_4_2 coordinates of region pixel (0, 0)

        // ---- phase 0: the prefetched patch -> LDS, [plane][row][parity][x / 2][16 ch]
        __syncthreads();                                                // the previous tile is done with the patch and the image
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int idx = tid + 256 * q;
            if (idx < NCH) {
                const int rp = idx / CPR, cc = idx - rp * CPR, px = cc >> 1;
                *reinterpret_cast<u32x4*>(patch + rp * PROWB + ((px & 1) * XHP + (px >> 1)) * 32 + 16 * (cc & 1)) = pre[q];
            }
        }
        // (column 35 of the odd-parity half is never read: 2 col + kw <= 34)
        __syncthreads();
        if (tile + (int)gridDim.x < n_tiles) patch_load(tile + gridDim.x);   // in flight during phases 1 and 2

        // ---- phase 1: block_4_2 over the region -> LDS image.  This wave: channels 16 nt1 .. + 15, M-tiles (wave >> 1) + 2 j
#pragma unroll 1
        for (int j = 0; j < C::N_MT1 / 2; j++) {
            const int mt = (wave >> 1) + 2 * j;                         // wave-uniform; mt < RH: region row mt, columns 0..15; mt = RH: column 16, row m
            const bool reg = mt < RH;
            const int row = reg ? mt : min(m, RH - 1), col = reg ? m : 16;
            const unsigned char* abase = patch + (2 * row) * PROWB + (reg ? p1lane : (uint32_t)(16 * 32 + 16 * (g & 1)));
            f32x4_m16 hi = f32x4_m16{bv2[0], bv2[1], bv2[2], bv2[3]}, lo = f32x4_m16{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < C::NST1; st++) {
                const uint32_t off = ghi ? (uint32_t)p1tap(2 * st + 1) : (uint32_t)p1tap(2 * st);
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(abase + off);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(abase + off + PPLANEB);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[st][0], a1, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[st][1], a0, lo, 0, 0, 0);
                hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[st][0], a0, hi, 0, 0, 0);
            }
            // D (transposed): row 4 g + r = channel 16 nt1 + 4 g + r, column m = this lane's region pixel; zero outside the 56 x 80 image
            const bool ok = (unsigned)(Ry0 + row) < (unsigned)H2 && (unsigned)(Rx0 + col) < (unsigned)W2;
            uint32_t pa[3], pb[3];
            s3p::act_split<2>(fmaf(hi[0], S3_F16_SCALE, lo[0]), fmaf(hi[1], S3_F16_SCALE, lo[1]), pa, ok);
            s3p::act_split<2>(fmaf(hi[2], S3_F16_SCALE, lo[2]), fmaf(hi[3], S3_F16_SCALE, lo[3]), pb, ok);
            if (reg || m < RH) {
                const int xh = col >> 1;
                unsigned char* dst = img + row * IROWB + ((col & 1) * XHR + xh) * 64 + 16 * ((2 * nt1 + (g >> 1) + ((xh >> 1) & 3)) & 3) + 8 * (g & 1);
                *reinterpret_cast<uint2*>(dst) = make_uint2(pa[0], pb[0]);
                *reinterpret_cast<uint2*>(dst + IPLANEB) = make_uint2(pa[1], pb[1]);
            }
        }
        __syncthreads();

        // ---- phase 2: block_4_3 from the LDS image.  This wave: channels 16 wave .. + 15, M-tiles j = 0, 1 = output rows 2 j, 2 j + 1
#pragma unroll 1
        for (int j = 0; j < 2; j++) {
            const unsigned char* ibase = img + (4 * j) * IROWB;              // image row 2 (2 j + (m >> 3)) with the lane offsets
            f32x4_m16 hi = f32x4_m16{bv3[0], bv3[1], bv3[2], bv3[3]}, lo = f32x4_m16{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < C::NST2; st++) {
                const unsigned char* src = ibase + p2tap(st) + (st % 3 == 2 ? p2lane1 : p2lane0);
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(src);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(src + IPLANEB);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[st][0], a1, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[st][1], a0, lo, 0, 0, 0);
                hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[st][0], a0, hi, 0, 0, 0);
            }
            uint32_t pa[3], pb[3];
            s3p::act_split<2>(fmaf(hi[0], S3_F16_SCALE, lo[0]), fmaf(hi[1], S3_F16_SCALE, lo[1]), pa);
            s3p::act_split<2>(fmaf(hi[2], S3_F16_SCALE, lo[2]), fmaf(hi[3], S3_F16_SCALE, lo[3]), pb);
            const int oy = ty0 + 2 * j + (m >> 3), ox = tx0 + (m & 7);
            uint16_t* o = out16 + (((size_t)b * H3 + oy) * W3 + ox) * C::C3 + 16 * wave + 4 * g;
            *reinterpret_cast<uint2*>(o) = make_uint2(pa[0], pb[0]);
            *reinterpret_cast<uint2*>(o + o_plane) = make_uint2(pa[1], pb[1]);
        }
    }   // persistent tile loop
}

}  // namespace hnet
