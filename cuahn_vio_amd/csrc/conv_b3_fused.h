// conv_b3_fused.h — block_3_0 (7x7 s1, 2 -> 16 @112x160) and block_3_1 (5x5 s2, 16 -> 32 -> 56x80) in ONE kernel
// (reference model_to_trace.py:108-109, :171-178 via conv() :7-15).  Round 3; fp16-plane arithmetic (HNET_PREC_F16X2) only.
//
// Unfused, block_3_0 writes its 16-channel map (294 MB per 256 pairs as two fp16 planes) and block_3_1 reads it straight back:
// 0.086 + 0.129 ms per step, both at 3.4 TB/s.  Here a workgroup owns an 8 x 16 tile of block_3_1 outputs:
//   phase 0  the (2*8 + 9) x 41-pixel input patch of the tile (fp32 NHWC [B][112][160][2] from the block's prep kernel) is split into
//            fp16 planes and staged in LDS (zero outside the image = block_3_0's zero padding); the NEXT tile's patch is prefetched
//            into registers while this one computes
//   phase 1  block_3_0 on the 19 x 35 region the tile needs: the pixel-pair GEMM of conv_first.h (M = 2 region rows x 16 pairs of
//            adjacent pixels, N = (pixel-in-pair, cout) = 32, one v_mfma_f32_32x32x16_f16 group per kernel row: K = 8 taps x 2 ch),
//            bias + LeakyReLU, zero outside the 112 x 160 image (= block_3_1's zero padding), split into planes and written to LDS
//            as 32-byte pixels [plane][row][column parity][column / 2][16 ch]
//   phase 2  block_3_1 straight from that LDS image: transposed 16x16x32 tiles (weights as the A operand, held in VGPRs): an M-tile
//            is one output row of the tile (16 pixels), a 32-deep K step is two filter taps x 16 channels (13 steps for the 25 taps);
//            wave w owns output channels 16 (w & 1) .. +15 and the four rows (w >> 1) + 2 j, which it advances TOGETHER through the K
//            loop (the weight fragment of a step is used four times); output as fp16 planes, 8 bytes per lane and plane.
// The 16-channel intermediate never touches HBM.  Workgroups are persistent (the 160 weight registers of a lane are loaded once).
//
// Arithmetic: the two-plane / two-accumulator form of the implicit-GEMM kernels (igemm_s3.h): w = W0 + W1 / 4096, a = A0 + A1 / 4096,
// hi += W0 A0, lo += W1 A0 + W0 A1, result = hi + lo / 4096 (three fp16 MFMAs per product, fp32 accumulation; the weights of both
// layers fit the register file in two planes, not in three).
//
// (Round-3 addendum: the two 16-byte chunks of a pixel are swapped where (column / 2 >> 2) & 1 - the phase-1 stores, 16 lanes at 32-byte stride, go from
// 4-way to 2-way bank conflicts, conflict share of the kernel 0.22 -> 0.12; the reads below stay conflict free.)
// LDS reads of phase 2 are conflict free by layout: a ds_read_b128 is served in four passes of 16 lanes - pixels {0-3, 12-15} of lane
// group g with pixels {4-11} of group g + 1 - and with 32-byte pixels group g reads even 16-byte slots, group g + 1 (the other channel
// half, or the next tap: an even number of slots away, plus one) odd ones.
#pragma once
#include <hip/hip_runtime.h>
#include "igemm_s3.h"
#include "kernels.h"

namespace hnet {

#ifdef HNET_B3_TRACE            // tools/trace_b3.hip: cycles per phase (s_memtime), summed in registers over the tiles of a workgroup and written once at the end
__device__ unsigned long long* g_b3_trace;
#define B3_T(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (tile_no > 1) tr_acc[k] += now_ - tr_prev; tr_prev = now_; } while (0)
#else
#define B3_T(k) do { } while (0)
#endif

struct B3Cfg {
    static constexpr int TH = 8, TW = 16, THREADS = 256;
    static constexpr int H0 = 112, W0 = 160, H1 = 56, W1 = 80, C0 = 16, C1 = 32;
    static constexpr int RH = 2 * TH + 3, RW = 2 * TW + 3;       // block_3_0 region of a tile: 19 x 35
    static constexpr int PH = 26, PW = 44;                       // input patch rows (RH + 6 = 25, + 1: the last M-tile's second row) x pixels
    // bytes per patch row and plane (one dword = both channels of a pixel).  A phase-1 M-tile pairs region rows r and r + 8: 8 x 176 = 128 mod 256,
    // so the two 128-byte windows a ds_read_b64 touches lie in disjoint bank halves (rows r, r + 1 overlapped in 12 of 32 banks; a pitch of
    // 384 fixes that too but puts the 16 rows of the leftover M-tiles on two bank offsets: measured conflict share 0.23 -> 0.34)
    static constexpr int PROWB = PW * 4;
    static constexpr int PPLANEB = PH * PROWB;
    static constexpr int XH = 18;                                // pixels per (row, parity): ceil(35 / 2)
    static constexpr int IROWB = 2 * XH * 32;                    // bytes per image row and plane
    static constexpr int IPLANEB = RH * IROWB;
    static constexpr int NP = 2;
    static constexpr int LDS_BYTES = NP * (PPLANEB + IPLANEB);   // patch + image (the kernel adds W0_BYTES of weight fragments behind them)
    static constexpr int SPARE_BYTES = 64;                       // target of the stores of lanes that hold no pixel (phase-1 epilogue)
    static constexpr int W0_BYTES = 7 * 3 * 64 * 16;             // block_3_0's fragments in the three-plane fp16 form (one accumulator per value: s3_mfma16)
    static constexpr int TILES_X = W1 / TW, TILES_Y = H1 / TH;   // 5 x 7 tiles per pair
    static constexpr int N_MT0 = 12;                             // phase-1 M-tiles: 10 pairs of rows + 2 tiles for the pixel pairs of columns 32..34
    static constexpr int NSTEP1 = 13;                            // phase-2 K steps (two taps each; the 26th tap has zero weights)
    static_assert(W1 % TW == 0 && H1 % TH == 0, "tiles cover the 56 x 80 output exactly");
};

// w0frag: [7 kernel rows][3 planes][64 lanes] x 16 B (hnet_create's b30_frag, the fragments of conv7_c2_s1_s3_kernel): lane (n = l & 31 = (dx, co), hh = l >> 5)
//         holds kk = 8 hh .. 8 hh + 7 of W'[kh][kk = 2 kw' + ci][n] = W[co][ci][kh][kw' - dx]   (conv_first.h; weight planes W0, W1, W2 of s3_format.h wsplit2h)
// w1frag: [2 n-tiles][13 steps][2 planes][64 lanes] x 16 B: lane (i = l & 15, g = l >> 4) holds output channel 16 nt + i, tap 2 st + (g >> 1),
//         input channels 8 (g & 1) .. + 7
template <int NP>
__global__ __launch_bounds__(256, 2) void block3_fused_kernel(const float* __restrict__ x_in, const u32x4* __restrict__ w0frag,
                                                              const float* __restrict__ bias0, const u32x4* __restrict__ w1frag,
                                                              const float* __restrict__ bias1, uint16_t* __restrict__ out16, size_t o_plane,
                                                              int n_tiles) {
    static_assert(NP == 2, "two fp16 planes (HNET_PREC_F16X2): the other modes run the two layers unfused");
    typedef B3Cfg C;
    constexpr int H0 = C::H0, W0 = C::W0, H1 = C::H1, W1 = C::W1, TH = C::TH, TW = C::TW, RH = C::RH, PH = C::PH, PW = C::PW;
    constexpr int PROWB = C::PROWB, PPLANEB = C::PPLANEB, XH = C::XH, IROWB = C::IROWB, IPLANEB = C::IPLANEB;
    typedef short bf16x4_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char* const patch = lds_raw;                               // [2 planes][PH][PW] dwords
    unsigned char* const img = lds_raw + 2 * PPLANEB;                   // [2 planes][RH][2 parities][XH][16 ch] fp16

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- weights -> registers, once per (persistent) workgroup
    // block_3_1's 104 registers stay; block_3_0's fragments (14 x 1 KiB) live in LDS and are read per kernel row (the two sets together
    // with both phases' accumulators do not fit 256 registers: 33 dwords of scratch in the first version)
    f16x8 w1[C::NSTEP1][2];
    u32x4* const w0s = reinterpret_cast<u32x4*>(lds_raw + C::LDS_BYTES);     // [7][3][64] x 16 B, lane-linear
    for (int i = tid; i < 7 * 3 * 64; i += 256) w0s[i] = w0frag[i];
    const int nt = wave & 1;                                            // phase 2: this wave's half of the 32 output channels
#pragma unroll
    for (int st = 0; st < C::NSTEP1; st++)
#pragma unroll
        for (int pl = 0; pl < 2; pl++) w1[st][pl] = __builtin_bit_cast(f16x8, w1frag[((nt * C::NSTEP1 + st) * 2 + pl) * 64 + lane]);

    // phase-1 lane roles (32x32x16, weights as A operand: D row = n = (dx, co), D column = pixel pair of the M-tile)
    const int pcol = lane & 31, hh = lane >> 5, prow = pcol >> 4, pair = pcol & 15;
    // D row (r & 3) + 8 (r >> 2) + 4 hh = n: group q = r >> 2 holds channels 8 (q & 1) + 4 hh + i: its four biases = one 16-byte LDS read
    __shared__ __attribute__((aligned(16))) float bias0s[16];
    if (tid < 16) bias0s[tid] = bias0[tid] * S3_F16_SCALE;             // the accumulator carries 4096 x the sum (one-accumulator form)
    // phase-2 lane roles (16x16x32 transposed: D row 4 g + r = output channel, D column m = pixel of the output row)
    const int m = lane & 15, g = lane >> 4;
    __shared__ __attribute__((aligned(16))) float bias1s[32];           // (read per output row: four registers less across phase 1)
    if (tid < 32) bias1s[tid] = bias1[tid];
    // lane part of the phase-2 read addresses: pixel column 2 m (+ kw), channel half g & 1; the tap of step st is t = 2 st + (g >> 1) = (kh, kw):
    // its offset is one of two compile-time constants per step (p2tap), selected by g >> 1  (t = 25 has zero weights: any valid address)
    // the two 16-byte chunks of a pixel are SWAPPED where (x / 2 >> 2) & 1 (bank model, tools/lds_bank_model.py: the 16 lanes of a phase-1 store group
    // then hit 2 instead of 4 addresses per bank, the phase-2 reads stay conflict free): the lane offset depends on the tap's kw >> 1 = 0, 1, 2
    // phase 1: this lane's patch address for the three M-tiles of its wave (mt = wave + 4 j): region row and pixel pair as in the epilogue
    uint32_t p1a[C::N_MT0 / 4];                                          // byte offsets into lds_raw (32-bit: registers are short in phase 1)
#pragma unroll
    for (int j = 0; j < C::N_MT0 / 4; j++) {
        const int mt = wave + 4 * j;
        int row, pr2;
        if (mt < 10) { row = mt < 8 ? mt + 8 * prow : 2 * mt + prow; pr2 = pair; }
        else { const int idx = (mt - 10) * 32 + pcol; row = idx >> 1; pr2 = 16 + (idx & 1); }
        const int rrow = row < RH ? row : RH - 1;                       // (rows beyond the region: reads stay inside the patch, nothing is stored)
        p1a[j] = (uint32_t)(rrow * PROWB + pr2 * 8 + 16 * hh);
    }
    int hi4 = 8;                                                        // opaque byte offset: two ds_read_b64 instead of one ds_read2_b64 (conv_first.h)
    asm volatile("" : "+v"(hi4));

    // ---- patch prefetch (registers): unconditional float2 loads from a clamped address, zero selected when consumed
    constexpr int PPT = (PH * PW + 255) / 256;
    float2 px[PPT];
    uint32_t okbits = 0;
    auto tile_origin = [&](int t, int& b, int& ty, int& tx) {
        int bid = s3p::xcd_tile(t, n_tiles, gridDim.x);
        tx = bid % C::TILES_X; bid /= C::TILES_X;
        ty = bid % C::TILES_Y;
        b = bid / C::TILES_Y;
    };
    auto patch_load = [&](int t) {
        int b, ty, tx;
        tile_origin(t, b, ty, tx);
        const int Py0 = 2 * ty * TH - 5, Px0 = 2 * tx * TW - 5;         // input pixel of patch (0, 0): region origin (2 ty0 - 2, 2 tx0 - 2) minus the 7x7 halo
        const float* inb = x_in + (size_t)b * H0 * W0 * 2;
        okbits = 0;
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = min(tid + q * 256, PH * PW - 1);
            const int pr = i / PW, pc = i - pr * PW;
            const int iy = Py0 + pr, ix = Px0 + pc;
            const bool ok = iy >= 0 && iy < H0 && ix >= 0 && ix < W0;
            px[q] = *reinterpret_cast<const float2*>(inb + (ok ? ((size_t)iy * W0 + ix) * 2 : 0));
            okbits |= ok ? (1u << q) : 0u;
        }
    };
    if ((int)blockIdx.x < n_tiles) patch_load(blockIdx.x);

    [[maybe_unused]] int tile_no = -1;
    [[maybe_unused]] unsigned long long tr_acc[4] = {0, 0, 0, 0}, tr_prev = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_no++;
        B3_T(0);
        int b, ty, tx;
        tile_origin(tile, b, ty, tx);
        const int ty0 = ty * TH, tx0 = tx * TW;
        const int Ry0 = 2 * ty0 - 2, Rx0 = 2 * tx0 - 2;                 // image coordinates of region pixel (0, 0)

        // ---- phase 0: the prefetched patch -> fp16 planes in LDS
        __syncthreads();                                                // the previous tile is done with the patch and the image
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = tid + q * 256;
            if (i < PH * PW) {
                const bool ok = (okbits >> q) & 1u;
                uint32_t pk[3];
                s3p::split_pair<2>(ok ? px[q].x : 0.f, ok ? px[q].y : 0.f, pk);
                const int pr = i / PW, pc = i - pr * PW;
                *reinterpret_cast<uint32_t*>(patch + pr * PROWB + pc * 4) = pk[0];
                *reinterpret_cast<uint32_t*>(patch + PPLANEB + pr * PROWB + pc * 4) = pk[1];
            }
        }
        __syncthreads();
        B3_T(1);

        // ---- phase 1: block_3_0 over the region -> LDS image.  M-tile mt < 8: region rows mt, mt + 8, pixel pairs 0..15; mt = 8: rows 16, 17;
        //      mt = 9: row 18 (+ an unused one); mt = 10, 11: the pairs 16, 17 (columns 32..35) of all rows, 32 (row, pair) combinations each.
        // Software pipeline over the 3 x 7 kernel-row steps of a wave (round 4; compiled as a loop a step was "six reads, lgkmcnt(0), three MFMAs" and the
        // epilogue of an M-tile - 120 vector instructions - ran with nothing beside it: 2900 cycles per M-tile for 672 cycles of MFMA, tools/trace_b3.hip):
        // a ring of three steps (activation fragment pair + the three weight fragments, read from LDS), step s + 2 is read while step s multiplies; one
        // accumulator per value (the three-plane form of s3_mfma16: the accumulator carries 4096 x the sum), two accumulator sets, so that the epilogue of
        // M-tile j sits in the shadows of the MFMAs of M-tile j + 1.
        {
            constexpr int NJ = C::N_MT0 / 4, NS1 = 7 * NJ, RING = 3, RINGW = 2;
            bf16x4_t fal[RING][2], fah[RING][2];                         // 8 fp16 = 4 taps x 2 ch per plane, as two 8-byte reads
            f16x8 fw[RINGW][3];                                          // weight fragments: one step ahead (registers), activations two
            f32x16 acc[2];
            auto rdA = [&](int sg) {
                const int j = sg / 7, kh = sg - 7 * j, sl = sg % RING;
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    const unsigned char* src = patch + p1a[j] + pl * PPLANEB + kh * PROWB;
                    fal[sl][pl] = *reinterpret_cast<const bf16x4_t*>(src);
                    fah[sl][pl] = *reinterpret_cast<const bf16x4_t*>(src + hi4);
                }
            };
            auto rdW = [&](int sg) {
                const int kh = sg % 7;
#pragma unroll
                for (int pl = 0; pl < 3; pl++) fw[sg % RINGW][pl] = __builtin_bit_cast(f16x8, w0s[(kh * 3 + pl) * 64 + lane]);
            };
            auto epi1 = [&](int j) {
                const int set = j & 1;
                const int mt = wave + 4 * j;                            // wave-uniform
                int row, pr2;                                           // this lane's region row and pixel pair
                if (mt < 10) { row = mt < 8 ? mt + 8 * prow : 2 * mt + prow; pr2 = pair; }
                else { const int idx = (mt - 10) * 32 + pcol; row = idx >> 1; pr2 = 16 + (idx & 1); }
                const bool row_ok = row < RH;
                // D row 8 q + 4 hh + i = (dx = q >> 1, co = 8 (q & 1) + 4 hh + i): a lane holds four consecutive channels of pixel 2 pr2 + dx
                const int iy = Ry0 + row;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int dx = q >> 1, col = 2 * pr2 + dx;
                    const bool ok = (unsigned)iy < (unsigned)H0 && (unsigned)(Rx0 + col) < (unsigned)W0;
                    uint32_t pa[3], pb[3];
                    s3p::act_split<2>(acc[set][4 * q], acc[set][4 * q + 1], pa, ok);
                    s3p::act_split<2>(acc[set][4 * q + 2], acc[set][4 * q + 3], pb, ok);
                    // (no branch - it would end the scheduling region the epilogue is interleaved in: lanes without a pixel store to a spare slot behind the weights)
                    const bool st_ok = row_ok && pr2 < XH;
                    unsigned char* dst = st_ok ? img + ((row * 2 + dx) * XH + pr2) * 32 + 16 * ((q & 1) ^ ((pr2 >> 2) & 1)) + 8 * hh
                                               : lds_raw + C::LDS_BYTES + C::W0_BYTES + 8 * hh - IPLANEB * 0;
                    *reinterpret_cast<uint2*>(dst) = make_uint2(pa[0], pb[0]);
                    *reinterpret_cast<uint2*>(st_ok ? dst + IPLANEB : dst + 16) = make_uint2(pa[1], pb[1]);
                }
            };
            rdA(0); rdW(0); rdA(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sg = 0; sg < NS1; sg++) {
                const int j = sg / 7, kh = sg - 7 * j, set = j & 1, sl = sg % RING;
                if (kh == 0) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const f32x4_m16 bq = *reinterpret_cast<const f32x4_m16*>(&bias0s[8 * (q & 1) + 4 * hh]);
#pragma unroll
                        for (int i = 0; i < 4; i++) acc[set][4 * q + i] = bq[i];
                    }
                }
                if (sg + 2 < NS1) rdA(sg + 2);
                if (sg + 1 < NS1) rdW(sg + 1);
                f16x8 a[2];
#pragma unroll
                for (int pl = 0; pl < 2; pl++) a[pl] = __builtin_bit_cast(f16x8, __builtin_shufflevector(fal[sl][pl], fah[sl][pl], 0, 1, 2, 3, 4, 5, 6, 7));
                acc[set] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[sg % RINGW][2], a[1], acc[set], 0, 0, 0);       // W2 A1 + W1 A0 + W0 A0 = 4096 a w, smallest first
                acc[set] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[sg % RINGW][1], a[0], acc[set], 0, 0, 0);
                acc[set] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[sg % RINGW][0], a[0], acc[set], 0, 0, 0);
                if (kh == 0 && j > 0) epi1(j - 1);                      // scheduled into the shadows of this M-tile's MFMAs (groups below)
                if (sg + 2 < NS1) __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
                else if (sg + 1 < NS1) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (j > 0) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                }
                if (j > 0 && kh > 0) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                if (kh == 6) __builtin_amdgcn_sched_barrier(0);
            }
            epi1(NJ - 1);
        }
        __syncthreads();
        if (tile + (int)gridDim.x < n_tiles) patch_load(tile + gridDim.x);   // in flight during phase 2 (its ten registers are not live in phase 1, which is short of them)
        B3_T(2);

        // ---- phase 2: block_3_1 from the LDS image.  This wave: channels 16 nt .. + 15, output rows (wave >> 1) + 2 j (j = 0..3), one row (M-tile) at a time: the
        //      weights of all 13 steps sit in registers.  Software pipeline (round 4; compiled as a loop every step was "two reads, lgkmcnt(0), three MFMAs": ~180
        //      cycles for 48 cycles of MFMA): a ring of four fragment pairs, the pair of step s + 4 is read into the slot step s has just consumed - across the four
        //      M-tiles - and every address is a lane-invariant register (p2a[step]) plus an instruction immediate (row, plane).
        {
            constexpr int NS = C::NSTEP1, RING = 4, NMT = 4;
            // the thirteen lane addresses are rebuilt per tile (13 x 4 vector instructions): kept across the tile loop they cost phase 1 thirteen registers it does not have
            int zero_op = 0;
            asm volatile("" : "+v"(zero_op));
            auto p2tap = [](int t) constexpr { const int tt = t < 24 ? t : 24; const int kh = tt / 5, kw = tt - 5 * kh; return kh * C::IROWB + ((kw & 1) * C::XH + (kw >> 1)) * 32; };
            const unsigned char* p2a[C::NSTEP1];                                // step st: lane groups 0, 1 read tap 2 st, groups 2, 3 tap 2 st + 1 (tap 25 has zero weights)
            {
                const bool ghi = (g >> 1) != 0;
        #pragma unroll
                for (int st = 0; st < C::NSTEP1; st++) {
                    const int tA = 2 * st < 24 ? 2 * st : 24, tB = 2 * st + 1 < 24 ? 2 * st + 1 : 24;
                    const int kw = ghi ? tB % 5 : tA % 5;
                    const uint32_t lanepart = (uint32_t)(m * 32 + 16 * ((g & 1) ^ (((m + (kw >> 1)) >> 2) & 1)));
                    p2a[st] = img + (2 * (wave >> 1)) * IROWB + zero_op + (ghi ? p2tap(tB) : p2tap(tA)) + lanepart;
                }
            }
            f16x8 fb[RING][2];
            f32x4_m16 hi2[2], lo2[2];
            auto rd2 = [&](int sg) {                                         // sg = 13 j + st
                const int j = sg / NS, st = sg - NS * j;
                const unsigned char* src = p2a[st] + (4 * j) * IROWB;         // image row 2 (oy = (wave >> 1) + 2 j): the wave's part is in p2a
                fb[sg % RING][0] = *reinterpret_cast<const f16x8*>(src);
                fb[sg % RING][1] = *reinterpret_cast<const f16x8*>(src + IPLANEB);
            };
            auto epi2 = [&](int j) {
                const int set = j & 1;
                const int oy = (wave >> 1) + 2 * j;
                // D (transposed): row 4 g + r = output channel 16 nt + 4 g + r, column m = pixel: 8 bytes (4 channels) per lane and plane
                uint32_t pa[3], pb[3];
                s3p::act_split<2>(fmaf(hi2[set][0], S3_F16_SCALE, lo2[set][0]), fmaf(hi2[set][1], S3_F16_SCALE, lo2[set][1]), pa);
                s3p::act_split<2>(fmaf(hi2[set][2], S3_F16_SCALE, lo2[set][2]), fmaf(hi2[set][3], S3_F16_SCALE, lo2[set][3]), pb);
                uint16_t* o = out16 + ((((size_t)b * H1 + ty0 + oy) * W1 + tx0 + m) * C::C1 + 16 * nt + 4 * g);
                *reinterpret_cast<uint2*>(o) = make_uint2(pa[0], pb[0]);
                *reinterpret_cast<uint2*>(o + o_plane) = make_uint2(pa[1], pb[1]);
            };
#pragma unroll
            for (int sg = 0; sg < RING; sg++) rd2(sg);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NMT; j++) {
                const int set = j & 1;
                hi2[set] = *reinterpret_cast<const f32x4_m16*>(&bias1s[16 * nt + 4 * g]);
                lo2[set] = f32x4_m16{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < NS; st++) {
                    const int sg = NS * j + st;
                    lo2[set] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[st][0], fb[sg % RING][1], lo2[set], 0, 0, 0);
                    lo2[set] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[st][1], fb[sg % RING][0], lo2[set], 0, 0, 0);
                    hi2[set] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[st][0], fb[sg % RING][0], hi2[set], 0, 0, 0);
                    if (sg + RING < NS * NMT) rd2(sg + RING);
                    if (j > 0 && st == 0) epi2(j - 1);                      // the previous row's epilogue sits in the shadows of this row's first MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    if (sg + RING < NS * NMT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (j > 0 && st < 8) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            epi2(NMT - 1);
        }
        B3_T(3);
    }   // persistent tile loop
#ifdef HNET_B3_TRACE
    if (blockIdx.x < 8 && lane == 0) for (int k = 0; k < 4; k++) g_b3_trace[(blockIdx.x * 4 + wave) * 5 + k] = tr_acc[k];
    if (blockIdx.x < 8 && lane == 0) g_b3_trace[(blockIdx.x * 4 + wave) * 5 + 4] = (unsigned long long)(tile_no - 1);
#endif
}

}  // namespace hnet
