// conv_b4_fused.h — block_4_0 (7x7 s1, 2->8 @224x320) and block_4_1 (5x5 s2, 8->16 -> 112x160) in ONE kernel
// (reference model_to_trace.py:210-211 via conv() :7-15; SURVEY.md §7 "fuse conv0->conv1 through LDS").
//
// block_4_0's output is the largest activation of the network (573 440 values per pair; 880 MB per 256 pairs in the
// three-plane bf16 format) and block_4_1 re-reads every value 6.25 times: unfused, the two layers cost 0.34 + 0.58 ms
// per 256 pairs, both bound by that traffic.  Here a workgroup owns an 8x32 tile of block_4_1 outputs:
//   phase 0  stage the input patch (25 x 76 px x 2 ch, zero outside the image) in LDS, split into three bf16 planes
//   phase 1  block_4_0 on the 19 x 67 region the tile needs: pixel-pair GEMM of conv_first.h (M = pairs of adjacent
//            pixels, N = (dx, cout) = 16, K = (kh, kw' 0..7, ci)) on v_mfma_f32_16x16x32_bf16 with split-bf16 x3
//            operands (six MFMAs per step, two kernel rows per step); bias + LeakyReLU, zero outside the image
//            (= block_4_1's zero padding), split again and written to LDS as 16-byte pixel chunks
//            [plane][row][column parity][column/2][8 ch]
//   phase 2  block_4_1 straight from that LDS image: per MFMA step lane group g reads the chunk of tap 4*step+g
//            (consecutive output columns -> consecutive chunks), six MFMAs per step against weights held in VGPRs;
//            output in S3 planes.
// The 8-channel intermediate never touches HBM.  Workgroups are persistent (one 512-thread workgroup per CU, tiles
// strided) so the 132 weight registers per lane are loaded once, not once per tile.
#pragma once
#include <hip/hip_runtime.h>
#include "igemm_s3.h"

namespace hnet {

namespace b4f {
constexpr int THREADS = 512, WAVES = THREADS / 64;
constexpr int TH1 = 8, TW1 = 32;                 // block_4_1 output tile
constexpr int RH = 2 * TH1 + 3, RW = 2 * TW1 + 3; // block_4_0 region: 19 x 67
constexpr int PH0 = RH + 6, PW0 = 76;            // input patch: 25 x 76 px (67 + 7 taps + pad)
constexpr int PROW0 = PW0 * 2;                   // bf16 elements per patch row (2 channels)
constexpr int PPLANE = PH0 * PROW0;              // elements per patch plane
constexpr int XH = 34;                           // chunks per (row, parity) of the S3 image (ceil(67/2) = 34)
constexpr int PLANE = RH * 2 * XH * 8;           // bf16 elements per plane of the S3 image
constexpr int N_MT0 = 2 * RH + 3;                // block_4_0 M-tiles: 19 rows x 2 + 3 for columns 64..66
constexpr int LDS_BYTES = 3 * PPLANE * 2 + 3 * PLANE * 2;
}  // namespace b4f

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// w0frag: [4 steps][3 planes][64 lanes] x 16 bytes (B fragments of the pixel-pair GEMM, pack in hnet_capi.hip)
// w1frag: [7 steps][3 planes][64 lanes] x 16 bytes
__global__ __launch_bounds__(512) void block4_fused_kernel(const float* __restrict__ x_in, const u32x4* __restrict__ w0frag,
                                                           const float* __restrict__ bias0, const u32x4* __restrict__ w1frag,
                                                           const float* __restrict__ bias1, uint16_t* __restrict__ out16,
                                                           size_t o_plane, int n_tiles, int dbg /* bit 3: walk the tiles from the end of the batch; bits 0-2 (only ever set by a -DHNET_B4_ABLATE profiling build): 1 = drop phase-1 stores, 2 = drop phase-2 MFMAs, 4 = drop phase-1 MFMAs */) {
    using namespace b4f;
    constexpr int H0 = 224, W0 = 320, H1 = 112, W1 = 160;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint16_t* patch = reinterpret_cast<uint16_t*>(lds_raw);                    // [3][PH0][PROW0] bf16
    uint16_t* img = patch + 3 * PPLANE;                                         // S3 image of the block_4_0 region

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;

    // ---- weights -> registers, once per (persistent) workgroup
    u32x4 w0[4][3];
#pragma unroll
    for (int st = 0; st < 4; st++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++) w0[st][pl] = w0frag[(st * 3 + pl) * 64 + lane];
    u32x4 w1[7][3];
#pragma unroll
    for (int st = 0; st < 7; st++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++) w1[st][pl] = w1frag[(st * 3 + pl) * 64 + lane];
    // The MFMAs are issued with the operands swapped (weights as A, pixels as B), i.e. they produce the TRANSPOSED tile:
    // D row 4g + r = output channel, D column m = pixel.  A lane then holds four consecutive channels of ONE pixel and
    // stores them with one 8-byte LDS write per plane; with pixels along the rows every value needed its own 2-byte
    // store and the four lane groups hit the same banks (19 % of this kernel's LDS cycles were conflicts).
    const int dx = g >> 1, co0 = 4 * (g & 1);            // phase 1: D row 4g + r = (dx, co0 + r)
    float bv[4], bv1[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { bv[r] = bias0[co0 + r]; bv1[r] = bias1[4 * g + r]; }
    // phase-1 A offsets of this lane group: step st covers kernel rows 2st, 2st+1; group g -> row 2st + (g>>1), taps 4(g&1)..
    int aoff[4];
#pragma unroll
    for (int st = 0; st < 4; st++) aoff[st] = min(2 * st + (g >> 1), 6) * PROW0 + 8 * (g & 1);   // row 7 has zero weights
    // phase-2 tap offsets: tap t = 4*step + g
    int tapoff[7];
#pragma unroll
    for (int st = 0; st < 7; st++) {
        const int t = 4 * st + g;
        const int kh = t / 5, kw = t - kh * 5;
        tapoff[st] = t < 25 ? ((kh * 2 + (kw & 1)) * XH + (kw >> 1)) * 8 : 0;
    }
    // phase-1 store position of this lane inside a regular M-tile: pixel column 2m + dx of a 32-column half, channels co0..co0+3
    const int e_lane = (dx * XH + m) * 8 + co0;

    // patch pixels of the NEXT tile are prefetched into registers while the current tile computes (the workgroup is
    // alone on its CU, so an un-overlapped global load would be fully exposed every tile)
    constexpr int PPT = (PH0 * PW0 + THREADS - 1) / THREADS;     // patch pixels per thread (4)
    float2 pre[PPT];
    uint32_t pre_ok = 0;            // validity bits; applied when the registers are consumed, so the loads stay in flight
    const bool reverse = (dbg & 8) != 0;     // walk the tiles from the end of the batch (Infinity-Cache friendly ordering experiments)
    auto patch_load = [&](int t) {
        pre_ok = 0;
        int bid = reverse ? n_tiles - 1 - t : t;
        const int bx = bid % (W1 / TW1); bid /= (W1 / TW1);
        const int by = bid % (H1 / TH1);
        const int b = bid / (H1 / TH1);
        const int Ry0 = 2 * by * TH1 - 2, Rx0 = 2 * bx * TW1 - 2;
        const float* inb = x_in + (size_t)b * H0 * W0 * 2;
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = min(tid + q * THREADS, PH0 * PW0 - 1);
            const int pr = i / PW0, pc = i - pr * PW0;
            const int iy = Ry0 - 3 + pr, ix = Rx0 - 3 + pc;
            const bool ok = iy >= 0 && iy < H0 && ix >= 0 && ix < W0;
            pre[q] = *reinterpret_cast<const float2*>(inb + (ok ? ((size_t)iy * W0 + ix) * 2 : 0));   // unconditional load
            pre_ok |= ok ? (1u << q) : 0u;
        }
    };
    if ((int)blockIdx.x < n_tiles) patch_load(blockIdx.x);

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int bid = reverse ? n_tiles - 1 - tile : tile;
        const int bx = bid % (W1 / TW1); bid /= (W1 / TW1);
        const int by = bid % (H1 / TH1);
        const int b = bid / (H1 / TH1);
        const int ty0 = by * TH1, tx0 = bx * TW1;
        const int Ry0 = 2 * ty0 - 2, Rx0 = 2 * tx0 - 2;      // image coordinates of region pixel (0,0)

        // ---- phase 0: prefetched patch -> three bf16 planes in LDS
        __syncthreads();                                     // previous tile's phase 2 is done with the LDS
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = tid + q * THREADS;
            if (i < PH0 * PW0) {
                const int pr = i / PW0, pc = i - pr * PW0;
                const bool ok = (pre_ok >> q) & 1u;
                uint16_t a0, a1, a2, b0, b1, b2;
                split3(ok ? pre[q].x : 0.f, a0, a1, a2);
                split3(ok ? pre[q].y : 0.f, b0, b1, b2);
                const int e = pr * PROW0 + pc * 2;
                *reinterpret_cast<uint32_t*>(&patch[e]) = (uint32_t)a0 | ((uint32_t)b0 << 16);
                *reinterpret_cast<uint32_t*>(&patch[PPLANE + e]) = (uint32_t)a1 | ((uint32_t)b1 << 16);
                *reinterpret_cast<uint32_t*>(&patch[2 * PPLANE + e]) = (uint32_t)a2 | ((uint32_t)b2 << 16);
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < n_tiles) patch_load(tile + gridDim.x);   // in flight during phases 1 and 2

        // ---- phase 1: block_4_0 over the region, into the S3 image
        for (int mt = wave; mt < N_MT0; mt += WAVES) {
            const bool regular = mt < 2 * RH;                // wave-uniform
            int row, pair;                                   // this lane's A row (a pixel pair of the region)
            if (regular) { row = mt >> 1; pair = (mt & 1) * 16 + m; }
            else { const int idx = (mt - 2 * RH) * 16 + m; row = min(idx >> 1, RH - 1); pair = 32 + (idx & 1); }
            const int abase = row * PROW0 + pair * 4;
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            if (!(dbg & 4))
#pragma unroll
            for (int st = 0; st < 4; st++) {
                bf16x8 a[3];
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {             // 8 bf16 = taps 4(g&1)..+3 x 2 ch of one kernel row; 8-byte aligned
                    const uint16_t* src = &patch[pl * PPLANE + abase + aoff[st]];
                    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(src);
                    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(src + 4);
                    a[pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, w0[st][0]);
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, w0[st][1]);
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, w0[st][2]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, a[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, a[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, a[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, a[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, a[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, a[0], acc, 0, 0, 0);
            }
            // D (transposed): row 4g + r = (dx, co0 + r), column m = pixel pair of the M-tile.  Outside the image = block_4_1's zero padding.
            auto put = [&](int e, bool ok) {
                uint16_t sa[4], sb[4], sc[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = acc[r] + bv[r];
                    v = v > 0.f ? v : v * 0.1f;
                    split3(ok ? v : 0.f, sa[r], sb[r], sc[r]);
                }
                *reinterpret_cast<uint2*>(&img[e]) = make_uint2((uint32_t)sa[0] | ((uint32_t)sa[1] << 16), (uint32_t)sa[2] | ((uint32_t)sa[3] << 16));
                *reinterpret_cast<uint2*>(&img[PLANE + e]) = make_uint2((uint32_t)sb[0] | ((uint32_t)sb[1] << 16), (uint32_t)sb[2] | ((uint32_t)sb[3] << 16));
                *reinterpret_cast<uint2*>(&img[2 * PLANE + e]) = make_uint2((uint32_t)sc[0] | ((uint32_t)sc[1] << 16), (uint32_t)sc[2] | ((uint32_t)sc[3] << 16));
            };
            if (dbg & 1) { if (acc[0] == 12345.f) img[0] = 1; }
            else if (regular) {
                const int rrow = mt >> 1, half = mt & 1;
                const int ix = Rx0 + half * 32 + 2 * m + dx;
                put(e_lane + (rrow * 2 * XH + half * 16) * 8, (unsigned)(Ry0 + rrow) < (unsigned)H0 && (unsigned)ix < (unsigned)W0);
            } else {
                const int idx = (mt - 2 * RH) * 16 + m;
                const int rrow = idx >> 1, rcol = 2 * (32 + (idx & 1)) + dx;
                if (rrow < RH && rcol < RW) {
                    const int iy = Ry0 + rrow, ix = Rx0 + rcol;
                    put(((rrow * 2 + (rcol & 1)) * XH + (rcol >> 1)) * 8 + co0, iy >= 0 && iy < H0 && ix >= 0 && ix < W0);
                }
            }
        }
        __syncthreads();

        // ---- phase 2: block_4_1 from the S3 image; tap t = 4*step + g, chunk = its 8 channels
        uint16_t* st_lds = patch + wave * (3 * 16 * 16);     // overlays the dead input patch
#pragma unroll 1
        for (int j = 0; j < 16 / WAVES; j++) {
            const int mt = wave * (16 / WAVES) + j;
            const int oy = mt >> 1, half = mt & 1;
            const int ox = half * 16 + m;
            const int base = ((2 * oy) * 2 * XH + ox) * 8;
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            if (!(dbg & 2))
#pragma unroll
            for (int st = 0; st < 7; st++) {
                bf16x8 a[3];
#pragma unroll
                for (int pl = 0; pl < 3; pl++) a[pl] = *reinterpret_cast<const bf16x8*>(&img[pl * PLANE + base + tapoff[st]]);
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, w1[st][0]);
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, w1[st][1]);
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, w1[st][2]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, a[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, a[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, a[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, a[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, a[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, a[0], acc, 0, 0, 0);
            }
            // D (transposed): row 4g + r = cout, column m = output pixel ox' = half*16 + m: 8 bytes (4 channels) per lane and plane
            {
                uint16_t sa[4], sb[4], sc[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = acc[r] + bv1[r];
                    v = v > 0.f ? v : v * 0.1f;
                    split3(v, sa[r], sb[r], sc[r]);
                }
                *reinterpret_cast<uint2*>(&st_lds[(0 * 16 + m) * 16 + 4 * g]) = make_uint2((uint32_t)sa[0] | ((uint32_t)sa[1] << 16), (uint32_t)sa[2] | ((uint32_t)sa[3] << 16));
                *reinterpret_cast<uint2*>(&st_lds[(1 * 16 + m) * 16 + 4 * g]) = make_uint2((uint32_t)sb[0] | ((uint32_t)sb[1] << 16), (uint32_t)sb[2] | ((uint32_t)sb[3] << 16));
                *reinterpret_cast<uint2*>(&st_lds[(2 * 16 + m) * 16 + 4 * g]) = make_uint2((uint32_t)sc[0] | ((uint32_t)sc[1] << 16), (uint32_t)sc[2] | ((uint32_t)sc[3] << 16));
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const size_t orow = ((size_t)b * H1 + ty0 + oy) * W1 + tx0 + half * 16;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int piece = q * 64 + lane;             // 96 pieces of 16 B: [plane][16 px][2 halves of 8 ch]
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
