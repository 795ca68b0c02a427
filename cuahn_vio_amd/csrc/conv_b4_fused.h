// conv_b4_fused.h — block_4_0 (7x7 s1, 2->8 @224x320) and block_4_1 (5x5 s2, 8->16 -> 112x160) in ONE kernel
// (reference model_to_trace.py:210-211 via conv() :7-15; SURVEY.md §7 "fuse conv0->conv1 through LDS").
//
// block_4_0's output is the largest activation of the network (573 440 values per pair; 880 MB per 256 pairs in the
// three-plane bf16 format) and block_4_1 re-reads every value 6.25 times: unfused, the two layers cost 0.34 + 0.58 ms
// per 256 pairs, both bound by that traffic.  Here a workgroup owns an 8x32 tile of block_4_1 outputs:
//   phase 0  stage the fp32 input patch (25 x 76 px x 2 ch, zero outside the image) in LDS, weights -> registers
//   phase 1  block_4_0 on the 19 x 67 region the tile needs (pixel-pair GEMM on v_mfma_f32_16x16x4_f32 exactly as
//            conv_first.h); bias + LeakyReLU, zero outside the image (= block_4_1's zero padding), split into three
//            bf16 planes and written to LDS as 16-byte pixel chunks [plane][row][column parity][column/2][8 ch]
//   phase 2  block_4_1 straight from that LDS image: per MFMA step lane group g reads the chunk of tap 4*step+g
//            (consecutive output columns -> consecutive chunks, conflict free), six v_mfma_f32_16x16x32_bf16 per step
//            (split-bf16 x3, igemm_s3.h) against weights held in 84 VGPRs; output in S3 planes.
// The 8-channel intermediate never touches HBM.  Workgroups are persistent (grid = 2 per CU, tiles strided) so the 112
// weight registers per lane are loaded once, not once per tile.
#pragma once
#include <hip/hip_runtime.h>
#include "igemm_s3.h"

namespace hnet {

namespace b4f {
constexpr int TH1 = 8, TW1 = 32;                 // block_4_1 output tile
constexpr int RH = 2 * TH1 + 3, RW = 2 * TW1 + 3; // block_4_0 region: 19 x 67
constexpr int PH0 = RH + 6, PW0 = 76;            // fp32 input patch: 25 x 76 px (67 + 7 taps + pad)
constexpr int PROW0 = PW0 * 2;                   // floats per patch row
constexpr int XH = 34;                           // chunks per (row, parity) of the S3 image (ceil(67/2) = 34)
constexpr int PLANE = RH * 2 * XH * 8;           // bf16 elements per plane of the S3 image
constexpr int N_MT0 = 2 * RH + 3;                // block_4_0 M-tiles: 19 rows x 2 + 3 for columns 64..66
constexpr int LDS_BYTES = PH0 * PROW0 * 4 + 3 * PLANE * 2;
}  // namespace b4f

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// w0frag: [28][64] floats (pack_first_weights, Cout 8);  w1frag: [7 steps][3 planes][64 lanes] x 16 bytes
__global__ __launch_bounds__(256) void block4_fused_kernel(const float* __restrict__ x_in, const float* __restrict__ w0frag,
                                                           const float* __restrict__ bias0, const u32x4* __restrict__ w1frag,
                                                           const float* __restrict__ bias1, uint16_t* __restrict__ out16,
                                                           size_t o_plane, int n_tiles) {
    using namespace b4f;
    constexpr int H0 = 224, W0 = 320, H1 = 112, W1 = 160;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* patch0 = reinterpret_cast<float*>(lds_raw);
    uint16_t* img = reinterpret_cast<uint16_t*>(lds_raw + PH0 * PROW0 * 4);    // S3 image of the block_4_0 region

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;

    // ---- weights -> registers, once per (persistent) workgroup
    float w0[28];
#pragma unroll
    for (int t = 0; t < 28; t++) w0[t] = w0frag[t * 64 + lane];
    u32x4 w1[7][3];
#pragma unroll
    for (int st = 0; st < 7; st++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++) w1[st][pl] = w1frag[(st * 3 + pl) * 64 + lane];
    const float bv = bias0[lane & 7];
    const float bv1 = bias1[m];
    const int dx = m >> 3, co = m & 7;
    // phase-2 tap offsets of this lane group: tap t = 4*step + g
    int tapoff[7];
#pragma unroll
    for (int st = 0; st < 7; st++) {
        const int t = 4 * st + g;
        const int kh = t / 5, kw = t - kh * 5;
        tapoff[st] = t < 25 ? ((kh * 2 + (kw & 1)) * XH + (kw >> 1)) * 8 : 0;
    }
    // phase-1 store position of this lane inside a regular M-tile: column 8g + 2r + dx of a 32-column half
    const int e_lane = (dx * XH + 4 * g) * 8 + co;

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    int bid = tile;
    const int bx = bid % (W1 / TW1); bid /= (W1 / TW1);
    const int by = bid % (H1 / TH1);
    const int b = bid / (H1 / TH1);
    const int ty0 = by * TH1, tx0 = bx * TW1;
    const int Ry0 = 2 * ty0 - 2, Rx0 = 2 * tx0 - 2;          // image coordinates of region pixel (0,0)

    // ---- phase 0: input patch
    __syncthreads();                                         // previous tile's phase 2 is done with the LDS
    const float* inb = x_in + (size_t)b * H0 * W0 * 2;
    for (int i = tid; i < PH0 * PW0; i += 256) {
        const int pr = i / PW0, pc = i - pr * PW0;
        const int iy = Ry0 - 3 + pr, ix = Rx0 - 3 + pc;
        const bool ok = iy >= 0 && iy < H0 && ix >= 0 && ix < W0;
        const float2 v = *reinterpret_cast<const float2*>(inb + (ok ? ((size_t)iy * W0 + ix) * 2 : 0));
        *reinterpret_cast<float2*>(&patch0[pr * PROW0 + pc * 2]) = ok ? v : make_float2(0.f, 0.f);
    }
    __syncthreads();

    // ---- phase 1: block_4_0 over the region, into the S3 image
    {
        for (int mt = wave; mt < N_MT0; mt += 4) {
            const bool regular = mt < 2 * RH;                // wave-uniform
            int row, pair;                                   // this lane's A row (a pixel pair of the region)
            if (regular) { row = mt >> 1; pair = (mt & 1) * 16 + m; }
            else { const int idx = (mt - 2 * RH) * 16 + m; row = min(idx >> 1, RH - 1); pair = 32 + (idx & 1); }
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 7; kh++) {
                const f32x4_t a = *reinterpret_cast<const f32x4_t*>(&patch0[(row + kh) * PROW0 + pair * 4 + 4 * g]);
#pragma unroll
                for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w0[kh * 4 + e], acc, 0, 0, 0);
            }
            // D: col n = (dx, co); row 4g + r = pixel pair within the M-tile.  Values outside the image are block_4_1's
            // zero padding.
            if (regular) {
                const int rrow = mt >> 1, half = mt & 1;
                const bool row_in = (unsigned)(Ry0 + rrow) < (unsigned)H0;
                const int ebase = e_lane + (rrow * 2 * XH + half * 16) * 8;
                const int ix0 = Rx0 + half * 32 + 8 * g + dx;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = acc[r] + bv;
                    v = v > 0.f ? v : v * 0.1f;
                    if (!row_in || (unsigned)(ix0 + 2 * r) >= (unsigned)W0) v = 0.f;
                    uint16_t sa, sb, sc;
                    split3(v, sa, sb, sc);
                    const int e = ebase + r * 8;
                    img[e] = sa; img[PLANE + e] = sb; img[2 * PLANE + e] = sc;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int idx = (mt - 2 * RH) * 16 + 4 * g + r;
                    const int rrow = idx >> 1, rcol = 2 * (32 + (idx & 1)) + dx;
                    if (rrow < RH && rcol < RW) {
                        const int iy = Ry0 + rrow, ix = Rx0 + rcol;
                        float v = acc[r] + bv;
                        v = v > 0.f ? v : v * 0.1f;
                        if (iy < 0 || iy >= H0 || ix < 0 || ix >= W0) v = 0.f;
                        uint16_t sa, sb, sc;
                        split3(v, sa, sb, sc);
                        const int e = ((rrow * 2 + (rcol & 1)) * XH + (rcol >> 1)) * 8 + co;
                        img[e] = sa; img[PLANE + e] = sb; img[2 * PLANE + e] = sc;
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 2: block_4_1 from the S3 image; tap t = 4*step + g, chunk = its 8 channels
    uint16_t* st_lds = reinterpret_cast<uint16_t*>(lds_raw) + wave * (3 * 16 * 16);   // overlays the dead input patch
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
        const int mt = wave * 4 + j;
        const int oy = mt >> 1, half = mt & 1;
        const int ox = half * 16 + m;
        const int base = ((2 * oy) * 2 * XH + ox) * 8;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 7; st++) {
            bf16x8 a[3];
#pragma unroll
            for (int pl = 0; pl < 3; pl++) a[pl] = *reinterpret_cast<const bf16x8*>(&img[pl * PLANE + base + tapoff[st]]);
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, w1[st][0]);
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, w1[st][1]);
            const bf16x8 b2 = __builtin_bit_cast(bf16x8, w1[st][2]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b0, acc, 0, 0, 0);
        }
        // D: col n = lane&15 = cout; row 4g + r = output pixel ox' = half*16 + 4g + r
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float v = acc[r] + bv1;
            v = v > 0.f ? v : v * 0.1f;
            uint16_t sa, sb, sc;
            split3(v, sa, sb, sc);
            const int px = 4 * g + r;
            st_lds[(0 * 16 + px) * 16 + m] = sa;
            st_lds[(1 * 16 + px) * 16 + m] = sb;
            st_lds[(2 * 16 + px) * 16 + m] = sc;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const size_t orow = ((size_t)b * H1 + ty0 + oy) * W1 + tx0 + half * 16;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int piece = q * 64 + lane;                 // 96 pieces of 16 B: [plane][16 px][2 halves of 8 ch]
            if (piece < 96) {
                const int pl = piece >> 5, rem = piece & 31, px = rem >> 1, hh = rem & 1;
                const u32x4 v = *reinterpret_cast<const u32x4*>(&st_lds[(pl * 16 + px) * 16 + hh * 8]);
                *reinterpret_cast<u32x4*>(out16 + pl * o_plane + (orow + px) * 16 + hh * 8) = v;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
  }   // persistent tile loop
}

}  // namespace hnet
