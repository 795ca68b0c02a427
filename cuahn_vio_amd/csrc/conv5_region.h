// conv5_region.h — the two 5x5 / stride-2 layers with many channels, block_2_2 (64 -> 128 @28x40 -> 14x20) and block_1_2
// (128 -> 128 @14x20 -> 7x10), as a REGION-RESIDENT convolution (round 2; reference model_to_trace.py:98,103 via conv() :7-15).
//
// In the implicit-GEMM kernel these layers re-stage every input value 6.25 times (once per tap that uses it): rocprofv3 shows 277 /
// 298 MB fetched past the L2 per launch for 113 / 58 MB of operands (the resident tiles' working set exceeds the 4 MB L2), and the
// A tile's ds_write_b128 traffic - LDS stores cost 2.5 x loads per byte (tools/lds_probe.hip) - is as large as the whole tile is read.
// Here a 512-thread workgroup owns TH = 7 output rows x all columns of one pair (140 / 70 pixels = 9 / 5 M-tiles of 16,
// row-major) and ALL 128 output channels, and walks K as (16-channel chunk, tap pair):
//   * per chunk the (2 TH + 3) x (2 WO + 3) input region x 16 channels is staged ONCE (three bf16 planes, 32 bytes per pixel,
//     [plane][row][column parity][column / 2][channel half] as in conv_patch_s2.h);
//   * wave (wm, wn) owns M-tiles wm, wm + 2, ... and output-channel tiles 2 wn, 2 wn + 1.  Its weight fragments of MFMA step s (taps 2 s, 2 s + 1 of
//     the chunk; 13 steps, the 26th tap has zero weights) come straight from global memory (host-packed per (step, wave, tile, plane,
//     lane): six coalesced 1 KiB loads per step, L2 resident) into registers, two steps ahead - the weights never touch LDS, so there
//     is no barrier inside a chunk (a first version staged 128 x 32 weight tiles through LDS with one barrier per step for eight
//     lock-stepped waves: 0.208 ms for block_2_2 against 0.180 ms of the implicit GEMM, MFMA busy 0.41 with LDS busy 0.22);
//   * every activation fragment read from LDS feeds 12 MFMAs (0.25 LDS reads per MFMA against 0.5 in the 64 x 64 GEMM tile) and the
//     only LDS stores are the region's.
// Lane (m = pixel of the M-tile, g): tap 2 s + (g >> 1), channel half g & 1 -> one ds_read_b128 at 32 bytes between lanes and 16 between
// the groups g, g + 1 (conflict free); an M-tile wraps from one output row to the next, and two region rows are 128 bytes mod 256 apart so
// that the wrapped lanes continue the bank pattern.  Output: S3 planes, 8 bytes (4 channels) per lane and plane.
//
// STATUS (end of round 2): correct (all parity tests pass with HNET_CONV5_REGION=1) and at PARITY with the implicit GEMM, not ahead:
// in-process A/B at batch 256 (profiles/r02_ab_conv5_region.log) block_1_2 0.103 vs 0.105 ms, block_2_2 0.184 vs 0.181 ms; MFMA busy
// 0.47 (GEMM 0.46), LDS busy 0.16 (0.47), VALU busy 0.27-0.32 (0.20): the LDS is relieved as intended, the time went to vector issue
// (region decomposition per chunk, 64-bit weight addresses, the epilogue) and to 7 % more MFMAs (9 M-tiles for 8.75, 52 steps for 50);
// the weight stream misses the L2 more than expected (485 MB fetched per launch for block_2_2).  Opt-in, default off; the next steps
// would be a lane-invariant staging table and SGPR-based weight addressing.
#pragma once
#include <hip/hip_runtime.h>
#include "igemm_s3.h"

namespace hnet {

template <int CIN_, int HI_, int WI_, int NP_ = 3> struct Conv5Cfg {
    static constexpr int CIN = CIN_, HI = HI_, WI = WI_, NP = NP_, COUT = 128;
    static constexpr int HO = HI / 2, WO = WI / 2, TH = 7, TILES_Y = HO / TH;
    static constexpr int NPIX = TH * WO, MT = (NPIX + 15) / 16;          // 140 / 70 pixels, 9 / 5 M-tiles
    static constexpr int RH = 2 * TH + 3, RW = 2 * WO + 3, XH = (RW + 1) / 2;
    // bytes per region row and plane, padded so that two region rows = 32 WO bytes mod 256: the lanes of an M-tile that wrap to the next
    // output row then continue the 32-byte-stride bank pattern of the lanes before the wrap (ROWB = 16 WO mod 128)
    static constexpr int ROWB0 = 2 * XH * 32;
    static constexpr int ROWB = ROWB0 + (((16 * WO - ROWB0) % 128) + 128) % 128;
    static constexpr int RPLANE = RH * ROWB;                              // bytes per region plane
    static constexpr int NCHUNK = CIN / 16, NSTEP = 13;
    static constexpr int LDS_BYTES = NP * RPLANE;
    static_assert(HO % TH == 0 && ROWB % 16 == 0 && (ROWB - 16 * WO) % 128 == 0, "tile / padding rule");
};

// in: S3 planes [NP][B][HI][WI][CIN] (plane stride i_plane elements); wfrag: [NCHUNK * 13 steps][4 wn][2 j][3 planes][64 lanes] x 16 B - MFMA A-operand
// fragments of output-channel tile 2 wn + j (lane (n, g): row n, K = 8 g .. 8 g + 7 = tap 2 s + (g >> 1), channels 16 c + 8 (g & 1) ..), hnet_capi.hip;
// out16: S3 planes [B][HO][WO][128]
template <int CIN, int HI, int WI, int NP>
__global__ __launch_bounds__(512) void conv5_region_kernel(const uint16_t* __restrict__ in, size_t i_plane, const u32x4* __restrict__ wfrag,
                                                              const float* __restrict__ bias, uint16_t* __restrict__ out16, size_t o_plane) {
    typedef Conv5Cfg<CIN, HI, WI, NP> C;
    constexpr int WO = C::WO, HO = C::HO, TH = C::TH, MT = C::MT, NPIX = C::NPIX, RH = C::RH, RW = C::RW, XH = C::XH, ROWB = C::ROWB;
    constexpr int RPLANE = C::RPLANE, NCHUNK = C::NCHUNK, NSTEP = C::NSTEP, TOTAL = NCHUNK * NSTEP;
    extern __shared__ __attribute__((aligned(16))) unsigned char region[];   // [NP][RH][ROWB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;                              // M-tiles wm, wm + 2, ...; output-channel tiles 2 wn, 2 wn + 1
    const int m = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / C::TILES_Y, ty = blockIdx.x % C::TILES_Y;
    const int oy0 = ty * TH;
    const int Ry0 = 2 * oy0 - 2, Rx0 = -2;                                // image coordinates of region pixel (0, 0): pad 2

    // ---- region staging: items (plane, row, column, half) of 16 bytes; decomposed per use (a table would cost 2 x 18 registers)
    constexpr int R_ITEMS = NP * RH * RW * 2, R_PER = (R_ITEMS + 511) / 512, RB = R_PER;     // RB loads in flight per lane (register budget: 72 accumulators + 48 weight registers)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (size_t)b * HI * WI * CIN), 0, 0x7FFFFFF0, 0x00020000);
    auto stage_region = [&](int chunk) {
#pragma unroll 1
        for (int q0 = 0; q0 < R_PER; q0 += RB) {
            u32x4 buf[RB];
            uint32_t loff[RB];
#pragma unroll
            for (int q = 0; q < RB; q++) {
                const int item = tid + 512 * (q0 + q);
                const int it = item < R_ITEMS ? item : 0;
                const int pl = it / (RH * RW * 2), r1 = it - pl * (RH * RW * 2), pr = r1 / (RW * 2), r2 = r1 - pr * (RW * 2), pc = r2 >> 1, hf = r2 & 1;
                const int iy = Ry0 + pr, ix = Rx0 + pc;
                const bool ok = item < R_ITEMS && iy >= 0 && iy < HI && ix >= 0 && ix < WI;
                const uint32_t goff = ok ? (uint32_t)(((size_t)pl * i_plane + ((size_t)iy * WI + ix) * CIN + hf * 8) * 2) : S3_OOB;
                loff[q] = item < R_ITEMS ? (uint32_t)(pl * RPLANE + pr * ROWB + (((pc & 1) * XH + (pc >> 1)) * 2 + hf) * 16) : 0xFFFFFFFFu;
                buf[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, goff, chunk * 32, 0));   // soffset: 16 channels = 32 bytes further
            }
#pragma unroll
            for (int q = 0; q < RB; q++)
                if (loff[q] != 0xFFFFFFFFu) *reinterpret_cast<u32x4*>(region + loff[q]) = buf[q];
        }
    };

    // ---- weight fragments: straight from global memory (L2 resident, 1.2 / 2.5 MB per layer) into registers, two steps ahead; no LDS, and
    // therefore no workgroup barrier inside a channel chunk
    bf16x8 w0[2][3], w1[2][3];
    auto w_load = [&](bf16x8 (&w)[2][3], int sg) {
        const u32x4* src = wfrag + ((size_t)sg * 4 + wn) * (2 * 3 * 64) + lane;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int pl = 0; pl < s3_wplanes<NP>; pl++) w[j][pl] = __builtin_bit_cast(bf16x8, src[(j * 3 + pl) * 64]);
    };

    // ---- accumulators: M-tiles wm + 2 i x output-channel tiles 2 wn + j; bias = initial value
    constexpr int MTW = (MT + 1) / 2;
    f32x4_m16 acc[MTW][2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        f32x4_m16 bv;
#pragma unroll
        for (int r = 0; r < 4; r++) bv[r] = bias[(2 * wn + j) * 16 + 4 * g + r] * s3_acc_scale<NP>;
#pragma unroll
        for (int i = 0; i < MTW; i++) acc[i][j] = bv;
    }
    // lane-invariant LDS byte offsets of this lane's pixel in each of its M-tiles (pixels beyond the tile are clamped, not stored)
    uint32_t a_off[MTW];
#pragma unroll
    for (int i = 0; i < MTW; i++) {
        const int pp = min((wm + 2 * i) * 16 + m, NPIX - 1), oy = pp / WO, ox = pp - oy * WO;
        a_off[i] = (uint32_t)(2 * oy * ROWB + ox * 32 + (g & 1) * 16);
    }
    const int tapsel = g >> 1;                                            // lane groups 2, 3 take the odd tap of a step

    // one MFMA step: taps 2 s, 2 s + 1 of the chunk in LDS (the 26th tap has zero weights: address clamped to tap 24)
    auto step = [&](int s, const bf16x8 (&w)[2][3]) {
        const int t = min(2 * s + tapsel, 24), kh = t / 5, kw = t - kh * 5;
        const uint32_t tap_off = (uint32_t)(kh * ROWB + ((kw & 1) * XH + (kw >> 1)) * 32);
        // fragments one M-tile ahead, and no further (sched_barrier): left alone the compiler hoists most of the 27 reads of a step and spills
        bf16x8 af[2][3];
#pragma unroll
        for (int pl = 0; pl < NP; pl++) af[0][pl] = *reinterpret_cast<const bf16x8*>(region + pl * RPLANE + a_off[0] + tap_off);
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            if (wm + 2 * i < MT) {                                        // wave-uniform (the last M-tile exists for wm = 0 only)
                if (i + 1 < MTW) {
#pragma unroll
                    for (int pl = 0; pl < NP; pl++) af[(i + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(region + pl * RPLANE + a_off[i + 1] + tap_off);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[i][0] = s3_mfma16<NP>(acc[i][0], w[0], af[i & 1]);
                acc[i][1] = s3_mfma16<NP>(acc[i][1], w[1], af[i & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    static_assert(NSTEP % 2 == 1, "13 steps per chunk: the two register sets swap roles every chunk");
    w_load(w0, 0);
    w_load(w1, 1);
#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; chunk++) {
        __syncthreads();                                                  // every wave has finished the previous chunk's region
        stage_region(chunk);
        __syncthreads();
        const int sg0 = chunk * NSTEP;
        // steps in pairs; the register set that was just used is refilled with the fragments of two steps later
        if ((chunk & 1) == 0) {
#pragma unroll 1
            for (int s = 0; s < NSTEP - 1; s += 2) {
                step(s, w0);     if (sg0 + s + 2 < TOTAL) w_load(w0, sg0 + s + 2);
                step(s + 1, w1); if (sg0 + s + 3 < TOTAL) w_load(w1, sg0 + s + 3);
            }
            step(NSTEP - 1, w0); if (sg0 + NSTEP + 1 < TOTAL) w_load(w0, sg0 + NSTEP + 1);
        } else {
#pragma unroll 1
            for (int s = 0; s < NSTEP - 1; s += 2) {
                step(s, w1);     if (sg0 + s + 2 < TOTAL) w_load(w1, sg0 + s + 2);
                step(s + 1, w0); if (sg0 + s + 3 < TOTAL) w_load(w0, sg0 + s + 3);
            }
            step(NSTEP - 1, w1); if (sg0 + NSTEP + 1 < TOTAL) w_load(w1, sg0 + NSTEP + 1);
        }
    }

    // ---- epilogue: D (transposed) row 4 g + r = channel of the n-tile, column m = pixel: 8 bytes per lane and plane
#pragma unroll
    for (int i = 0; i < MTW; i++) {
        const int pp = (wm + 2 * i) * 16 + m;
        if (wm + 2 * i < MT && pp < NPIX) {
            const int oy = pp / WO, ox = pp - oy * WO;
            uint16_t* o = out16 + (((size_t)b * HO + oy0 + oy) * WO + ox) * 128 + 2 * wn * 16 + 4 * g;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                uint32_t pa[3], pb[3];
                s3p::act_split<NP>(acc[i][j][0], acc[i][j][1], pa);
                s3p::act_split<NP>(acc[i][j][2], acc[i][j][3], pb);
#pragma unroll
                for (int pl = 0; pl < NP; pl++) *reinterpret_cast<uint2*>(o + j * 16 + pl * o_plane) = make_uint2(pa[pl], pb[pl]);
            }
        }
    }
}

}  // namespace hnet
