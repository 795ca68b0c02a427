// conv_patch_s2.h — stride-2 KSxKS convolution, 16 -> 32 channels, straight from an LDS-resident input patch
// (block_3_1: 5x5 and block_4_2: 3x3, both 112x160x16 -> 56x80x32; reference model_to_trace.py:108,212 via conv() :7-15).
//
// These two layers have a large image and few channels: in the implicit-GEMM kernel every input value is staged
// 6.25 (5x5) / 2.25 (3x3) times and the kernel is bound by that gather traffic (0.33 / 0.19 ms per 256 pairs).
// Here a workgroup owns an 8 x 16 tile of output pixels, copies the (2*8+KS-2) x (2*16+KS-2) input region ONCE into
// LDS (three bf16 planes, 16-byte chunks of 8 channels laid out [plane][row][column parity][channel half][column/2]
// so that consecutive output columns read consecutive chunks) and feeds v_mfma_f32_16x16x32_bf16 directly from it:
// MFMA step st covers taps 2st, 2st+1; lane group g reads the chunk (tap 2st + (g>>1), channel half g&1).
// Split-bf16 x3 arithmetic as igemm_s3.h (six MFMAs per step).  Each wave owns one 16-channel half of the outputs
// and keeps its weight fragments in VGPRs for the lifetime of the workgroup.
//
// Operand reads (round 2, two steps).  Round 1 kept the region as [plane][row][column parity][channel half][column / 2] chunks and read
// a fragment with one ds_read_b128: 2-way bank conflicts on every read (conflict share 0.49-0.53 of the LDS cycles, profiles/r01_v7) -
// a ds_read_b128 is served in passes of 16 lanes that mix two lane groups (rows {0-3, 12-15} of group g with {4-11} of g + 1,
// tools/lds_probe.hip), and the two groups' 16-chunk windows were XH chunks apart.  Step 1 (B128 = false, kept as the A/B switch
// HNET_PATCH_B128=0): two ds_read_b64 per fragment, odd groups high half first, weights packed in the matching channel order -
// conflict free by construction, but the probe shows a ds_read_b64 costs as many LDS cycles (5.3) as a conflict-free ds_read_b128
// (5.4), i.e. twice per byte.  Step 2 (B128 = true, default): [plane][row][column parity][column / 2][channel half] - 32 bytes
// between the lanes of a group, 16 between the groups g, g + 1 of a tap - read with one ds_read_b128, conflict free in that pass
// structure: block_3_1 0.199 -> 0.192 ms, block_4_2 0.150 -> 0.140 ms in process (profiles/r02_ab_patch_b128.log).
// NP = number of bf16 planes (3 = split-bf16, 1 = plain bf16 operands).
#pragma once
#include <hip/hip_runtime.h>
#include "igemm_s3.h"
#include "conv_b4_fused.h"      // s3p::split_pair / lrelu

namespace hnet {

template <int KS, int NP = 3> struct PatchS2Cfg {
    static constexpr int CIN = 16, COUT = 32, PAD = (KS - 1) / 2;
    static constexpr int TH = 8, TW = 16;                       // output tile
    static constexpr int RH = 2 * TH + KS - 2, RW = 2 * TW + KS - 2;
    static constexpr int XH = (RW + 1) / 2;                     // chunks per (row, parity, half)
    static constexpr int PLANE = RH * 2 * 2 * XH * 8;           // bf16 elements per plane
    static constexpr int NSTEP = (KS * KS + 1) / 2;             // two taps per 32-deep MFMA step
    static constexpr int LDS_BYTES = NP * PLANE * 2;
};

typedef float f32x4_p __attribute__((ext_vector_type(4)));
typedef short bf16x4_p __attribute__((ext_vector_type(4)));

// in: S3 planes [3][B][H][W][16];  wfrag: [2 halves of cout][NSTEP][3][64 lanes] x 16 B;  out16: [3][B][H/2][W/2][32]
template <int KS, int NP, bool B128 = false, int RB5 = 1>
__global__ __launch_bounds__(256, 2) void conv_patch_s2_kernel(const uint16_t* __restrict__ in, size_t i_plane,
                                                            const u32x4* __restrict__ wfrag, const float* __restrict__ bias,
                                                            uint16_t* __restrict__ out16, size_t o_plane, int H, int W,
                                                            int n_tiles, int reverse) {
    typedef PatchS2Cfg<KS, NP> C;
    constexpr int TH = C::TH, TW = C::TW, RH = C::RH, RW = C::RW, XH = C::XH, PLANE = C::PLANE, NSTEP = C::NSTEP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint16_t* img = reinterpret_cast<uint16_t*>(lds_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int nt = wave & 1;                                    // this wave's half of the output channels
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;              // = floor((H + 2*PAD - KS) / 2) + 1 for KS = 3, 5
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;

    bf16x8 wv[NSTEP][3];
#pragma unroll
    for (int st = 0; st < NSTEP; st++)
#pragma unroll
        for (int pl = 0; pl < s3_wplanes<NP>; pl++) wv[st][pl] = __builtin_bit_cast(bf16x8, wfrag[((nt * NSTEP + st) * 3 + pl) * 64 + lane]);
    // operands swapped (weights as A): the MFMA yields the transposed tile, D row 4g + r = cout, column m = output pixel,
    // so that a lane stores four consecutive channels of one pixel with one 8-byte LDS write per plane (conv_b4_fused.h)
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; r++) bv[r] = bias[nt * 16 + 4 * g + r] * s3_acc_scale<NP>;
    // element offset of tap t inside a plane (channel half 0): compile-time constants, selected per lane group when used
    // (a 13-entry per-lane table cost 13 VGPRs: with the 156 weight registers of the 5x5 kernel that was one wave per SIMD less)
    auto tap_elem = [](int t) constexpr {
        const int tt = t < KS * KS ? t : KS * KS - 1;            // the padding tap of the last step has zero weights
        const int kh = tt / KS, kw = tt - kh * KS;
        return B128 ? ((kh * 2 + (kw & 1)) * XH + (kw >> 1)) * 16 : (((kh * 2 + (kw & 1)) * 2) * XH + (kw >> 1)) * 8;
    };
    // B128 (A/B switch HNET_PATCH_B128): region rows as [column parity][column / 2][channel half] chunks (32 bytes per pixel) and ONE
    // ds_read_b128 per fragment plane - 32 bytes between the lanes of a group, 16 between the groups g, g + 1 of a tap: 5.4 LDS cycles per
    // wave-instruction in tools/lds_probe.hip, against 2 x 5.3 for the two ds_read_b64 of the half-major layout
    const int lane_off = B128 ? (g & 1) * 8 : (g & 1) * (XH * 8 + 4);   // channel half of the group; (b64 form) odd groups: high 8 bytes first
    const int second = 4 - 8 * (g & 1);                         // element offset from the first to the second 8-byte read
    // staging items of one region row: (plane, column, channel half), NP*RW*2 of them, ITEMS per lane
    constexpr int ROW_ITEMS = NP * RW * 2, ITEMS = (ROW_ITEMS + 63) / 64;
    constexpr int RB = KS == 5 ? RB5 : 5;                       // region rows loaded per batch (register budget: RB5 = 1 in split-bf16)

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        // reverse: walk the tiles from the end of the batch.  The producer wrote this 440 MB input front to back and the
        // 256 MB Infinity Cache is LRU: what it still holds is the END of the tensor, which the forward order would evict
        // before reaching it
        int bid = s3p::xcd_tile(reverse ? n_tiles - 1 - tile : tile, n_tiles, gridDim.x);
        const int bx = bid % tiles_x; bid /= tiles_x;
        const int by = bid % tiles_y;
        const int b = bid / tiles_y;
        const int ty0 = by * TH, tx0 = bx * TW;
        const int Ry0 = 2 * ty0 - C::PAD, Rx0 = 2 * tx0 - C::PAD;

        // ---- stage the input region: 16-byte copies, zero outside the image.  A wave takes whole region rows
        // (row index wave-uniform); its lanes walk the 3 planes x RW columns x 2 channel halves of the row with
        // per-lane offsets that were decomposed once, outside the tile loop.
        __syncthreads();
        // per-lane decomposition of the staging items, formed per tile on an opaque lane id so that these 16 registers are
        // not live across the MFMA phase (the 5x5 kernel holds 156 weight registers)
        int it_pc[ITEMS], it_loff[ITEMS];
        size_t it_goff[ITEMS];
        uint32_t it_valid = 0;
        {
            int lv = lane;
            asm volatile("" : "+v"(lv));
#pragma unroll
            for (int q = 0; q < ITEMS; q++) {
                const int item = lv + 64 * q;
                const int it = item < ROW_ITEMS ? item : 0;
                const int pl = it / (RW * 2), rem = it - pl * (RW * 2), pc = rem >> 1, hf = rem & 1;
                it_pc[q] = pc;
                it_goff[q] = pl * i_plane + hf * 8;
                it_loff[q] = pl * PLANE + (B128 ? ((pc & 1) * XH + (pc >> 1)) * 2 + hf : ((pc & 1) * 2 + hf) * XH + (pc >> 1)) * 8;
                if (item < ROW_ITEMS) it_valid |= 1u << q;
            }
        }
        const uint16_t* inb = in + (size_t)b * H * W * 16;
        // loads are issued RB rows at a time into registers and consumed (zero select + ds_write) afterwards, so that
        // RB*ITEMS 16-byte loads per lane are in flight instead of one (a select next to its load serialises them)
        constexpr int ROWS_PER_WAVE = (RH + 3) / 4;
#pragma unroll
        for (int r0 = 0; r0 < ROWS_PER_WAVE; r0 += RB) {
            u32x4 buf[RB][ITEMS];
#pragma unroll
            for (int rr = 0; rr < RB; rr++) {
                const int pr = wave + 4 * (r0 + rr);
                const int iy = Ry0 + pr;
                const bool row_ok = pr < RH && iy >= 0 && iy < H;
                const size_t grow = (size_t)(row_ok ? iy : 0) * W * 16;
#pragma unroll
                for (int q = 0; q < ITEMS; q++) {
                    const int ix = Rx0 + it_pc[q];
                    const bool ok = row_ok && ix >= 0 && ix < W;
                    buf[rr][q] = *reinterpret_cast<const u32x4*>(inb + it_goff[q] + (ok ? grow + (size_t)ix * 16 : 0));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rr = 0; rr < RB; rr++) {
                const int pr = wave + 4 * (r0 + rr);
                const int iy = Ry0 + pr;
                const bool row_ok = iy >= 0 && iy < H;
                const int lrow = pr * 2 * 2 * XH * 8;
#pragma unroll
                for (int q = 0; q < ITEMS; q++) {
                    const int ix = Rx0 + it_pc[q];
                    const bool ok = row_ok && ix >= 0 && ix < W;
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    if (pr < RH && (it_valid & (1u << q))) *reinterpret_cast<u32x4*>(&img[lrow + it_loff[q]]) = ok ? buf[rr][q] : z;
                }
            }
        }
        __syncthreads();

        // ---- MFMAs: this wave's M-tiles are output rows (wave>>1), (wave>>1)+2, ... of the tile
#pragma unroll 1
        for (int j = 0; j < TH / 2; j++) {
            const int oy = (wave >> 1) + 2 * j;
            const int base = (B128 ? ((2 * oy) * 2 * XH + m) * 16 : ((2 * oy) * 2 * 2 * XH + m) * 8) + lane_off;   // row 2*oy, parity 0, column m, this group's half
            int gv = g;                                           // opaque per M-tile: the 13 tap addresses are formed next to their
            asm volatile("" : "+v"(gv));                          // reads instead of being hoisted into 13 registers
            const bool odd_tap = (gv >> 1) != 0;                  // lane groups 2, 3 take the odd tap of a step
            f32x4_p acc = {bv[0], bv[1], bv[2], bv[3]};          // bias = initial accumulator
#pragma unroll
            for (int st = 0; st < NSTEP; st++) {
                bf16x8 a[3];
#pragma unroll
                for (int pl = 0; pl < NP; pl++) {
                    const uint16_t* src = &img[pl * PLANE + base + (odd_tap ? tap_elem(2 * st + 1) : tap_elem(2 * st))];
                    if constexpr (B128) {
                        a[pl] = *reinterpret_cast<const bf16x8*>(src);
                    } else {
                        const bf16x4_p first = *reinterpret_cast<const bf16x4_p*>(src);
                        const bf16x4_p other = *reinterpret_cast<const bf16x4_p*>(src + second);
                        a[pl] = __builtin_shufflevector(first, other, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
                acc = s3_mfma16<NP>(acc, wv[st], a);
            }
            // D (transposed): row 4g + r = cout within the half, column m = output column: a lane holds four consecutive channels of one
            // pixel, 8 bytes per plane, and stores them straight to global memory (the 16 lanes of a group x 4 groups cover 32 of the 64
            // bytes of 16 consecutive pixels; the partner wave of the other channel half writes the rest of each line).  Round 1 staged
            // the tile through LDS to form 16-byte stores: one more LDS round trip and two waits per M-tile for nothing measurable.
            {
                const int Y = ty0 + oy, X = tx0 + m;
                uint32_t pa[3], pb[3];
                s3p::act_split<NP>(acc[0], acc[1], pa);
                s3p::act_split<NP>(acc[2], acc[3], pb);
                if (Y < Ho && X < Wo) {
                    uint16_t* o = out16 + (((size_t)b * Ho + Y) * Wo + X) * 32 + nt * 16 + 4 * g;
#pragma unroll
                    for (int pl = 0; pl < NP; pl++) *reinterpret_cast<uint2*>(o + pl * o_plane) = make_uint2(pa[pl], pb[pl]);
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// block_3_2 / block_4_3: 3x3 stride 2, 32 -> 64 channels, 56x80 -> 28x40, in the same patch-resident form (round 2).
// In the implicit-GEMM kernel these two layers ran at 0.257 MFMA busy (profiles/r02_v2): K is only 288 (nine 32-deep
// K-tiles), so prologue, barriers and the 24 KB epilogue of every 64 x 64 tile weigh as much as its MFMAs, and the A tile is
// staged 2.25 times.  Here: a workgroup owns 4 rows x 20 columns of output pixels = five M-tiles of 16 pixels (row-major over
// the tile, so the 40-column image splits into two tiles without a ragged one), stages the 9 x 41 x 32-channel input region
// once (16-byte chunks of 8 channels, [plane][row][column parity][channel quarter][column / 2]), and wave w computes output
// channels 16 w .. 16 w + 15 of all five M-tiles with its 9 x 3 weight fragments (one tap x 32 channels per MFMA step) in
// registers for the lifetime of the (persistent) workgroup.  Lane group g reads channel quarter g of the step's tap; the
// 16 bytes as two ds_read_b64 with the halves swapped for odd groups (see above), weights packed to match.
template <int NP = 3> struct Patch32Cfg {
    static constexpr int CIN = 32, COUT = 64, KS = 3, PAD = 1;
    static constexpr int TH = 4, TW = 20, MT = TH * TW / 16;     // 80 pixels = 5 M-tiles
    static constexpr int RH = 2 * TH + 1, RW = 2 * TW + 1;
    static constexpr int XH = (RW + 1) / 2;                     // chunks per (row, parity, quarter)
    // bf16 elements per region row and plane, + 64 bytes: an M-tile of 16 pixels wraps from output row oy to row oy + 1; with two region
    // rows = 128 bytes mod 256 the wrapped lanes continue the 32-byte-stride bank pattern of the first ones (see the kernel)
    static constexpr int ROW = 2 * 4 * XH * 8 + 32;
    static constexpr int PLANE = RH * ROW;
    static constexpr int NSTEP = 9;
    static constexpr int LDS_BYTES = NP * PLANE * 2;
};

// in: S3 planes [3][B][56][80][32];  wfrag: [4 n-tiles][9 steps][3][64 lanes] x 16 B;  out16: [3][B][28][40][64]
template <int NP>
__global__ __launch_bounds__(256, 2) void conv_patch32_s2_kernel(const uint16_t* __restrict__ in, size_t i_plane,
                                                                  const u32x4* __restrict__ wfrag, const float* __restrict__ bias,
                                                                  uint16_t* __restrict__ out16, size_t o_plane, int n_tiles, int reverse) {
    typedef Patch32Cfg<NP> C;
    constexpr int TH = C::TH, TW = C::TW, MT = C::MT, RH = C::RH, RW = C::RW, XH = C::XH, ROW = C::ROW, PLANE = C::PLANE, NSTEP = C::NSTEP;
    constexpr int H = 56, W = 80, Ho = 28, Wo = 40, tiles_x = Wo / TW, tiles_y = Ho / TH;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint16_t* img = reinterpret_cast<uint16_t*>(lds_raw);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int nt = wave;

    bf16x8 wv[NSTEP][3];
#pragma unroll
    for (int st = 0; st < NSTEP; st++)
#pragma unroll
        for (int pl = 0; pl < s3_wplanes<NP>; pl++) wv[st][pl] = __builtin_bit_cast(bf16x8, wfrag[((nt * NSTEP + st) * 3 + pl) * 64 + lane]);
    f32x4_p bv;
#pragma unroll
    for (int r = 0; r < 4; r++) bv[r] = bias[nt * 16 + 4 * g + r] * s3_acc_scale<NP>;

    // element offset of tap t inside a plane (quarter 0, column 0)
    auto tap_elem = [](int t) constexpr {
        const int kh = t / 3, kw = t - kh * 3;
        return kh * ROW + ((kw & 1) * 4 * XH + (kw >> 1) * 2) * 8;
    };
    // LDS layout of a region row: [column parity][quarter pair][column / 2][quarter & 1] chunks of 16 bytes: a lane (pixel m, group g) reads
    // chunk ((g >> 1) XH + x0 + m) 2 + (g & 1), i.e. 32 bytes between the lanes of a group and 16 bytes between groups g and g + 1.
    // tools/lds_probe.hip (MI355X, LDS cycles per wave-instruction): ds_read_b128 5.4 in this pattern (and for lane groups a multiple of
    // 256 bytes apart), 8.1 when the groups are 128 / 272 / 336 bytes apart (the quarter-major layout used first); ds_read_b64 5.3 whatever
    // the pattern - so the two-ds_read_b64 form (conflict free by construction) paid 10.6 cycles per fragment plane for the same 1 KiB.
    const int goff = ((g >> 1) * XH * 2 + (g & 1)) * 8;
    // staging items of one region row: (plane, column, channel quarter)
    constexpr int ROW_ITEMS = NP * RW * 4, ITEMS = (ROW_ITEMS + 63) / 64;
    constexpr int ROWS_PER_WAVE = (RH + 3) / 4;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int bid = s3p::xcd_tile(reverse ? n_tiles - 1 - tile : tile, n_tiles, gridDim.x);
        const int bx = bid % tiles_x; bid /= tiles_x;
        const int by = bid % tiles_y;
        const int b = bid / tiles_y;
        const int ty0 = by * TH, tx0 = bx * TW;
        const int Ry0 = 2 * ty0 - 1, Rx0 = 2 * tx0 - 1;

        // ---- stage the input region: 16-byte copies, zero outside the image; a wave takes whole region rows.  Buffer loads: the
        // per-lane part of an address is one 32-bit byte offset, out-of-image chunks get an offset beyond the descriptor's range (= 0),
        // and all of the wave's rows (3 x 7 loads per lane) are in flight before the first LDS write
        __syncthreads();
        {
            int it_pc[ITEMS], it_loff[ITEMS];
            uint32_t it_goff[ITEMS];
            uint32_t it_valid = 0;
            int lv = lane;
            asm volatile("" : "+v"(lv));                         // formed per tile: not live across the MFMA phase
#pragma unroll
            for (int q = 0; q < ITEMS; q++) {
                const int item = lv + 64 * q;
                const int it = item < ROW_ITEMS ? item : 0;
                const int pl = it / (RW * 4), rem = it - pl * (RW * 4), pc = rem >> 2, cq = rem & 3;
                it_pc[q] = pc;
                it_goff[q] = (uint32_t)((pl * i_plane + cq * 8 + (size_t)pc * 32) * 2);
                it_loff[q] = pl * PLANE + ((pc & 1) * 4 * XH + ((cq >> 1) * XH + (pc >> 1)) * 2 + (cq & 1)) * 8;
                if (item < ROW_ITEMS) it_valid |= 1u << q;
            }
            const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (size_t)b * H * W * 32), 0, 0x7FFFFFF0, 0x00020000);
            u32x4 buf[ROWS_PER_WAVE][ITEMS];
#pragma unroll
            for (int rr = 0; rr < ROWS_PER_WAVE; rr++) {
                const int pr = wave + 4 * rr;
                const int iy = Ry0 + pr;
                const bool row_ok = pr < RH && iy >= 0 && iy < H;
                const int soff = row_ok ? (iy * W + Rx0) * 64 : 0;      // bytes; Rx0 = -1 for the left tile: the column test below covers it
#pragma unroll
                for (int q = 0; q < ITEMS; q++) {
                    const int ix = Rx0 + it_pc[q];
                    const bool ok = row_ok && ix >= 0 && ix < W;
                    buf[rr][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? it_goff[q] + (uint32_t)soff : S3_OOB, 0, 0));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rr = 0; rr < ROWS_PER_WAVE; rr++) {
                const int pr = wave + 4 * rr;
#pragma unroll
                for (int q = 0; q < ITEMS; q++)
                    if (pr < RH && (it_valid & (1u << q))) *reinterpret_cast<u32x4*>(&img[pr * ROW + it_loff[q]]) = buf[rr][q];
            }
        }
        __syncthreads();

        // ---- MFMAs: all five M-tiles for this wave's 16 output channels
        uint16_t* const ob = out16 + ((size_t)b * Ho * Wo + (size_t)ty0 * Wo + tx0) * 64 + nt * 16 + 4 * g;
#pragma unroll
        for (int j = 0; j < MT; j++) {
            // this lane's pixel of M-tile j: p = 16 j + m, row p / 20, column p % 20; input row 2 oy, column 2 ox (parity 0, chunk ox)
            const int pp = 16 * j + m, oy = pp / TW, ox = pp - oy * TW;
            const int base = 2 * oy * ROW + ox * 16 + goff;
            f32x4_p acc = bv;                                     // bias = initial accumulator
#pragma unroll
            for (int st = 0; st < NSTEP; st++) {
                bf16x8 a[3];
#pragma unroll
                for (int pl = 0; pl < NP; pl++) {
                    a[pl] = *reinterpret_cast<const bf16x8*>(&img[pl * PLANE + base + tap_elem(st)]);
                }
                acc = s3_mfma16<NP>(acc, wv[st], a);
            }
            // schedule (round 4): the M-tile's fragment reads first, then its MFMAs - left to itself the scheduler put every step's reads directly in front of its
            // MFMAs and waited lgkmcnt(0) per step (conv_b42_fused.h has the measurement)
            __builtin_amdgcn_sched_group_barrier(0x100, NP * NSTEP, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, (NP == 3 ? 6 : NP == 2 ? 3 : 1) * NSTEP, 0);
            // D (transposed): row 4g + r = cout 16 nt + 4g + r, column m = pixel: 8 bytes per lane and plane straight to global memory
            // (forming whole 128-byte lines through LDS first, the four waves' quarters together, measured 0.1028 vs 0.1004 ms: not worth
            // the two extra barriers per tile)
            uint32_t pa[3], pb[3];
            s3p::act_split<NP>(acc[0], acc[1], pa);
            s3p::act_split<NP>(acc[2], acc[3], pb);
            uint16_t* o = ob + (size_t)(oy * Wo + ox) * 64;
#pragma unroll
            for (int pl = 0; pl < NP; pl++) *reinterpret_cast<uint2*>(o + pl * o_plane) = make_uint2(pa[pl], pb[pl]);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 4, fp16-plane mode: the same layer with the NEXT tile's region in flight during the MFMA phase.  conv_patch32_s2_kernel loads a
// tile's 47 KB region and waits for it before every MFMA phase: 15 700 cycles per tile for 2160 cycles of MFMA (0.055 ms at batch 256, two
// workgroups per CU cover only part of each other's waits).  Here the 21 buffer loads of tile t + 1 are issued right after tile t's region
// has been written to LDS and stay in flight (84 registers) under tile t's MFMAs.  To afford them the weights are held in the two-plane /
// two-accumulator form (planes 2 = W0 and 1 = W1 of the packed one-accumulator fragments: 72 registers instead of 108; hi += W0 A0,
// lo += W0 A1 + W1 A0, result 4096 hi + lo - the arithmetic of the fused kernels) and the fragment reads of an M-tile are scheduled in two groups.
template <int NP>
__global__ __launch_bounds__(256, 2) void conv_patch32_s2_pf_kernel(const uint16_t* __restrict__ in, size_t i_plane,
                                                                     const u32x4* __restrict__ wfrag, const float* __restrict__ bias,
                                                                     uint16_t* __restrict__ out16, size_t o_plane, int n_tiles, int reverse) {
    static_assert(NP == 2, "fp16-plane mode");
    typedef Patch32Cfg<NP> C;
    constexpr int TH = C::TH, TW = C::TW, MT = C::MT, RH = C::RH, RW = C::RW, XH = C::XH, ROW = C::ROW, PLANE = C::PLANE, NSTEP = C::NSTEP;
    constexpr int H = 56, W = 80, Ho = 28, Wo = 40, tiles_x = Wo / TW, tiles_y = Ho / TH;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint16_t* img = reinterpret_cast<uint16_t*>(lds_raw);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int nt = wave;

    f16x8 wv[NSTEP][2];                                  // [step][W0, W1]
#pragma unroll
    for (int st = 0; st < NSTEP; st++) {
        wv[st][0] = __builtin_bit_cast(f16x8, wfrag[((nt * NSTEP + st) * 3 + 2) * 64 + lane]);
        wv[st][1] = __builtin_bit_cast(f16x8, wfrag[((nt * NSTEP + st) * 3 + 1) * 64 + lane]);
    }
    f32x4_p bv;
#pragma unroll
    for (int r = 0; r < 4; r++) bv[r] = bias[nt * 16 + 4 * g + r];

    auto tap_elem = [](int t) constexpr {
        const int kh = t / 3, kw = t - kh * 3;
        return kh * ROW + ((kw & 1) * 4 * XH + (kw >> 1) * 2) * 8;
    };
    const int goff = ((g >> 1) * XH * 2 + (g & 1)) * 8;
    constexpr int ROW_ITEMS = NP * RW * 4, ITEMS = (ROW_ITEMS + 63) / 64;
    constexpr int ROWS_PER_WAVE = (RH + 3) / 4;

    auto origin = [&](int tile, int& b, int& ty0, int& tx0) {
        int bid = s3p::xcd_tile(reverse ? n_tiles - 1 - tile : tile, n_tiles, gridDim.x);
        const int bx = bid % tiles_x; bid /= tiles_x;
        const int by = bid % tiles_y;
        b = bid / tiles_y;
        ty0 = by * TH; tx0 = bx * TW;
    };
    // the region of a tile: [plane][row][column][channel quarter] items of 16 bytes; a wave takes whole region rows (wave + 4 rr), a lane the items lane + 64 q of a row
    u32x4 buf[ROWS_PER_WAVE][ITEMS];
    auto issue_loads = [&](int tile) {
        int b, ty0, tx0;
        origin(tile, b, ty0, tx0);
        const int Ry0 = 2 * ty0 - 1, Rx0 = 2 * tx0 - 1;
        int lv = lane;
        asm volatile("" : "+v"(lv));                             // the item arithmetic is formed per tile: nothing of it lives across the MFMA phase
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (size_t)b * H * W * 32), 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
        for (int rr = 0; rr < ROWS_PER_WAVE; rr++) {
            const int pr = wave + 4 * rr;
            const int iy = Ry0 + pr;
            const bool row_ok = pr < RH && iy >= 0 && iy < H;
            const int soff = row_ok ? (iy * W + Rx0) * 64 : 0;      // bytes; Rx0 = -1 for the left tile: the column test below covers it
#pragma unroll
            for (int q = 0; q < ITEMS; q++) {
                const int item = lv + 64 * q;
                const int it = item < ROW_ITEMS ? item : 0;
                const int pl = it / (RW * 4), rem = it - pl * (RW * 4), pc = rem >> 2, cq = rem & 3;
                const uint32_t goff_b = (uint32_t)((pl * i_plane + cq * 8 + (size_t)pc * 32) * 2);
                const int ix = Rx0 + pc;
                const bool ok = row_ok && ix >= 0 && ix < W;      // out of the image: an offset beyond the descriptor's range reads zeros
                buf[rr][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? goff_b + (uint32_t)soff : S3_OOB, 0, 0));
            }
        }
    };
    auto write_lds = [&]() {
        int lv = lane;
        asm volatile("" : "+v"(lv));
#pragma unroll
        for (int rr = 0; rr < ROWS_PER_WAVE; rr++) {
            const int pr = wave + 4 * rr;
#pragma unroll
            for (int q = 0; q < ITEMS; q++) {
                const int item = lv + 64 * q;
                const int it = item < ROW_ITEMS ? item : 0;
                const int pl = it / (RW * 4), rem = it - pl * (RW * 4), pc = rem >> 2, cq = rem & 3;
                const int loff = pl * PLANE + ((pc & 1) * 4 * XH + ((cq >> 1) * XH + (pc >> 1)) * 2 + (cq & 1)) * 8;
                if (pr < RH && item < ROW_ITEMS) *reinterpret_cast<u32x4*>(&img[pr * ROW + loff]) = buf[rr][q];
            }
        }
    };

    if ((int)blockIdx.x < n_tiles) issue_loads(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int b, ty0, tx0;
        origin(tile, b, ty0, tx0);
        __syncthreads();                                         // the previous tile's MFMA phase is done with the region
        write_lds();
        __syncthreads();
        if (tile + (int)gridDim.x < n_tiles) issue_loads(tile + gridDim.x);      // in flight during the MFMA phase
        __builtin_amdgcn_sched_barrier(0);

        uint16_t* const ob = out16 + ((size_t)b * Ho * Wo + (size_t)ty0 * Wo + tx0) * 64 + nt * 16 + 4 * g;
#pragma unroll
        for (int j = 0; j < MT; j++) {
            const int pp = 16 * j + m, oy = pp / TW, ox = pp - oy * TW;
            const int base = 2 * oy * ROW + ox * 16 + goff;
            f32x4_p hi = bv, lo = f32x4_p{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < NSTEP; st++) {
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(&img[base + tap_elem(st)]);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(&img[PLANE + base + tap_elem(st)]);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[st][0], a1, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[st][1], a0, lo, 0, 0, 0);
                hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[st][0], a0, hi, 0, 0, 0);
            }
            // the reads of steps 0 - 4, then MFMAs with the reads of steps 5 - 8 between them (two fragment groups of 40 + 32 registers)
            __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 15, 0);
            uint32_t pa[3], pb[3];
            s3p::act_split<NP>(fmaf(hi[0], S3_F16_SCALE, lo[0]), fmaf(hi[1], S3_F16_SCALE, lo[1]), pa);
            s3p::act_split<NP>(fmaf(hi[2], S3_F16_SCALE, lo[2]), fmaf(hi[3], S3_F16_SCALE, lo[3]), pb);
            uint16_t* o = ob + (size_t)(oy * Wo + ox) * 64;
#pragma unroll
            for (int pl = 0; pl < NP; pl++) *reinterpret_cast<uint2*>(o + pl * o_plane) = make_uint2(pa[pl], pb[pl]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

}  // namespace hnet
