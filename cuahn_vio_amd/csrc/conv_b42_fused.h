// conv_b42_fused.h — block_4_2 (3x3 s2, 16 -> 32 @112x160 -> 56x80) and block_4_3 (3x3 s2, 32 -> 64 -> 28x40) in ONE kernel
// (reference model_to_trace.py:212-213 via conv() :7-15).  Round 3; fp16-plane arithmetic (HNET_PREC_F16X2) only.
//
// Unfused, the two layers are the patch kernels of conv_patch_s2.h: 0.101 + 0.056 ms per 256 pairs at 0.16 - 0.28 matrix-pipe busy - they
// wait for memory: block_4_2 reads 294 MB and writes 147 MB that block_4_3 reads straight back.  Here a workgroup owns a 4 x 8 tile of
// block_4_3 outputs (35 tiles per pair):
//   phase 0  the 19 x 35-pixel patch of block_4_1's output the tile needs (two fp16 planes, 16 channels = 32 bytes per pixel and plane) is
//            copied into LDS as [plane][row][column parity][column / 2][16 ch] by LDS-DMA from the bordered array the block-4 kernel writes
//            (zero outside the image = block_4_2's zero padding); the copy of tile t + 1 runs under phase 2 of tile t
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

#ifdef HNET_B42_TRACE           // tools/trace_b42.hip: cycles per phase (s_memtime), summed in registers over the tiles of a workgroup and written once at the end
__device__ unsigned long long* g_b42_trace;    // (a store per stamp made the traced waves wait for their LDS-DMA at every stamp: vmcnt(0) in front of the store)
#define B42_T(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (tile_no > 1) tr_acc[k] += now_ - tr_prev; tr_prev = now_; } while (0)
#else
#define B42_T(k) do { } while (0)
#endif

struct B42Cfg {
    static constexpr int TH = 4, TW = 8, THREADS = 256;
    static constexpr int H1 = 112, W1 = 160, H2 = 56, W2 = 80, H3 = 28, W3 = 40, C1 = 16, C2 = 32, C3 = 64;
    static constexpr int RH = 2 * TH + 1, RW = 2 * TW + 1;       // block_4_2 region of a tile: 9 x 17
    static constexpr int PH = 2 * RH + 1, PW = 2 * RW + 1;       // block_4_1 patch: 19 x 35
    static constexpr int XHP = 18;                               // patch pixels per (row, parity)
    static constexpr int PROWB = 2 * XHP * 32 + 16, PPLANEB = PH * PROWB;   // + 16: the rows of the column-16 M-tile (one lane per row) spread over the banks
    static constexpr int XHR = 9;                                // region pixels per (row, parity)
    static constexpr int IROWB = 2 * XHR * 64, IPLANEB = RH * IROWB;
    static constexpr int LDS_DATA = 2 * (PPLANEB + IPLANEB);     // patch + image
    static constexpr int LDS_BYTES = LDS_DATA + 3 * 256 * 16;   // + the DMA source offsets of every lane (12 dwords; registers are short: see dma_issue)
    static constexpr int TILES_X = W3 / TW, TILES_Y = H3 / TH;   // 5 x 7 tiles per pair
    static constexpr int N_MT1 = RH + 1;                         // phase-1 M-tiles: one per region row (columns 0..15) + one for column 16
    static constexpr int NST1 = 5, NST2 = 9;
    static_assert(W3 % TW == 0 && H3 % TH == 0, "tiles cover the 28 x 40 output exactly");
};

// in16:   block_4_1 output, fp16 planes with a zero border [2][B][B42_HP][B42_WP][16] (kernels.h; i_plane elements per plane)
// w2frag: block_4_2, [2 n-tiles][5 steps][2 planes][64 lanes] x 16 B: lane (i = l & 15, g = l >> 4): channel 16 nt + i, tap 2 st + (g >> 1), ci 8 (g & 1) + j
// w3frag: block_4_3, [4 n-tiles][9 steps][2 planes][64 lanes] x 16 B: lane (i, g): channel 16 nt + i, tap st, ci 8 g + j
template <int NP>
__global__ __launch_bounds__(256, 2) void block42_fused_kernel(const uint16_t* __restrict__ in16, size_t i_plane, const u32x4* __restrict__ w2frag,
                                                               const float* __restrict__ bias2, const u32x4* __restrict__ w3frag,
                                                               const float* __restrict__ bias3, uint16_t* __restrict__ out16, size_t o_plane,
                                                               int n_tiles) {
    static_assert(NP == 2, "two fp16 planes (HNET_PREC_F16X2): the other modes run the two layers unfused");
    typedef B42Cfg C;
    constexpr int H2 = C::H2, W2 = C::W2, H3 = C::H3, W3 = C::W3, TH = C::TH, TW = C::TW, RH = C::RH;
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

    // ---- patch copy by LDS-DMA (round 4; until then through registers: 11 global loads, 11 ds_write_b128 and ~300 address instructions per thread and tile, at
    // 0.31 matrix-pipe busy).  in16 is the bordered array of kernels.h B42_* (written by the fused block-4 kernel): the zero border is in memory, the patch of a tile
    // starts at pixel (16 ty, 32 tx) of the bordered image.  The patch [2 planes][PH rows][PROWB bytes] is one linear run of NCHK 16-byte chunks in LDS;
    // wave-instruction k (64 chunks) is issued by wave k % 4, lane l fetching chunk 64 k + l, whose source offset from the tile's first pixel is lane invariant.
    // chunk cb of a row = (parity, x / 2, channel half) in the parity-split order; cb = 72 is the row's bank pad (copies chunk 0), pixel 35 of the odd half is the
    // next pixel in memory (never read)
    constexpr int CPRB = PROWB / 16, CPP = PPLANEB / 16, NCHK = 2 * CPP, NINS = (NCHK + 63) / 64, IPW = (NINS + 3) / 4;
    static_assert(PROWB % 16 == 0 && CPRB == 4 * XHP + 1, "patch rows are whole chunks");
    static_assert(IPW <= 12, "offset stash");
    uint32_t dma_ok = 0;
    {
        // the eleven offsets of a lane are needed once per tile: they live in LDS ([3][256 threads] x 16 bytes), not in registers (kept there, the compiler spilled
        // seven of them and waited vmcnt(0) - for the previous copies - in front of every scratch reload: 3900 cycles per tile for the eleven DMA instructions)
        uint32_t dsrc[12];
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int q = 64 * (wave + 4 * i) + lane;
            const int qq = (i < IPW && q < NCHK) ? q : 0;
            const int pl = qq / CPP, r = qq - pl * CPP, prow = r / CPRB, cb0 = r - prow * CPRB, cb = cb0 < 4 * XHP ? cb0 : 0;
            const int par = cb / (2 * XHP), xh = (cb - par * 2 * XHP) >> 1, px = 2 * xh + par;
            dsrc[i] = (uint32_t)(((size_t)pl * i_plane + ((size_t)prow * B42_WP + px) * C::C1 + 8 * (cb & 1)) * 2);
            if (i < IPW) dma_ok |= q < NCHK ? (1u << i) : 0u;
        }
        u32x4* stash = reinterpret_cast<u32x4*>(lds_raw + C::LDS_DATA);
#pragma unroll
        for (int a = 0; a < 3; a++) stash[a * 256 + tid] = u32x4{dsrc[4 * a], dsrc[4 * a + 1], dsrc[4 * a + 2], dsrc[4 * a + 3]};
    }
    // buffer form (32-bit lane offset + scalar tile offset): with a flat 64-bit address per lane the compiler kept the eleven offsets as register pairs, spilled
    // three of them and waited vmcnt(0) - for the previous copies - in front of every reload
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)in16, 0, 0x7FFFFFF0, 0x00020000);
    auto dma_issue = [&](int t) {
        int b, ty, tx;
        tile_origin(t, b, ty, tx);
        const int tile_off = (int)((((size_t)b * B42_HP + 4 * ty * TH) * B42_WP + 4 * tx * TW) * (C::C1 * 2));      // bytes; < 2^31 (hnet_create bounds max_batch)
        const u32x4* stash = reinterpret_cast<const u32x4*>(lds_raw + C::LDS_DATA);
        const u32x4 d0 = stash[tid], d1 = stash[256 + tid], d2 = stash[512 + tid];
        const uint32_t dsrc[12] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3], d2[0], d2[1], d2[2], d2[3]};
#pragma unroll
        for (int i = 0; i < IPW; i++) {
            const int k = wave + 4 * i;                      // wave-uniform
            if (k < NINS && ((dma_ok >> i) & 1u)) {
                const uint32_t vo = dsrc[i];                 // (locals: hipcc 7.2 drops the host-side instantiation when these are array elements / expressions, igemm_pipe.h)
                const int so = tile_off;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rP, (void __attribute__((address_space(3)))*)(lds_raw + k * 1024), 16, vo, so, 0, 0);
            }
        }
    };
    if ((int)blockIdx.x < n_tiles) dma_issue(blockIdx.x);

    [[maybe_unused]] int tile_no = -1;
    [[maybe_unused]] unsigned long long tr_acc[5] = {0, 0, 0, 0, 0}, tr_prev = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_no++;
        B42_T(0);
        int b, ty, tx;
        tile_origin(tile, b, ty, tx);
        const int ty0 = ty * TH, tx0 = tx * TW;
        const int Ry0 = 2 * ty0 - 1, Rx0 = 2 * tx0 - 1;                 // block_4_2 coordinates of region pixel (0, 0)

        // ---- phase 0: this wave's share of the patch copy has landed (it ran under phase 2 of the previous tile) ...
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // ... and everybody else's; the previous tile's phase 2 is done with the image
        asm volatile("" ::: "memory");
        B42_T(1);

        // ---- phase 1: block_4_2 over the region -> LDS image.  This wave: channels 16 nt1 .. + 15, M-tiles (wave >> 1) + 2 j.
        // Software pipeline over the five M-tiles of a wave (round 4; tools/trace_b42.hip: compiled as a loop, an M-tile took 800 - 1000 cycles for 240 cycles of
        // MFMA - ten reads, each waited for with lgkmcnt(0) in front of its MFMAs, then the epilogue with nothing beside it): the ten fragment reads of M-tile
        // j + 1 are issued before the MFMAs of M-tile j, and the epilogue of M-tile j - 1 (vector arithmetic + two LDS stores) is interleaved with those MFMAs.
        {
            constexpr int NJ = C::N_MT1 / 2, RING = 8;
            // fragment ring of eight 32-deep steps (64 registers; two whole sets = 80 did not fit beside the 112 weight registers): step st of M-tile j sits in slot
            // (5 j + st) % 8.  Steps 0 - 2 of M-tile j + 1 are read before the MFMAs of M-tile j, its steps 3, 4 (the slots of M-tile j's steps 0, 1) in the middle of them
            f16x8 fa[RING][2];
            f32x4_m16 hi[2], lo[2];
            auto rd1 = [&](int j, int st0, int st1) {
                const int mt = (wave >> 1) + 2 * j;                     // wave-uniform; mt < RH: region row mt, columns 0..15; mt = RH: column 16, row m
                const bool reg = mt < RH;
                const int row = reg ? mt : min(m, RH - 1);
                const unsigned char* abase = patch + (2 * row) * PROWB + (reg ? p1lane : (uint32_t)(16 * 32 + 16 * (g & 1)));
#pragma unroll
                for (int st = st0; st < st1; st++) {
                    const uint32_t off = ghi ? (uint32_t)p1tap(2 * st + 1) : (uint32_t)p1tap(2 * st);
                    fa[(C::NST1 * j + st) % RING][0] = *reinterpret_cast<const f16x8*>(abase + off);
                    fa[(C::NST1 * j + st) % RING][1] = *reinterpret_cast<const f16x8*>(abase + off + PPLANEB);
                }
            };
            auto mm1 = [&](int j, int st0, int st1) {
                const int set = j & 1;
                if (st0 == 0) { hi[set] = f32x4_m16{bv2[0], bv2[1], bv2[2], bv2[3]}; lo[set] = f32x4_m16{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int st = st0; st < st1; st++) {
                    const int sl = (C::NST1 * j + st) % RING;
                    lo[set] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[st][0], fa[sl][1], lo[set], 0, 0, 0);
                    lo[set] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[st][1], fa[sl][0], lo[set], 0, 0, 0);
                    hi[set] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[st][0], fa[sl][0], hi[set], 0, 0, 0);
                }
            };
            auto epi1 = [&](int j, int set) {
                const int mt = (wave >> 1) + 2 * j;
                const bool reg = mt < RH;
                const int row = reg ? mt : min(m, RH - 1), col = reg ? m : 16;
                // D (transposed): row 4 g + r = channel 16 nt1 + 4 g + r, column m = this lane's region pixel; zero outside the 56 x 80 image
                const bool ok = (unsigned)(Ry0 + row) < (unsigned)H2 && (unsigned)(Rx0 + col) < (unsigned)W2;
                uint32_t pa[3], pb[3];
                s3p::act_split<2>(fmaf(hi[set][0], S3_F16_SCALE, lo[set][0]), fmaf(hi[set][1], S3_F16_SCALE, lo[set][1]), pa, ok);
                s3p::act_split<2>(fmaf(hi[set][2], S3_F16_SCALE, lo[set][2]), fmaf(hi[set][3], S3_F16_SCALE, lo[set][3]), pb, ok);
                {   // (no branch - it would end the scheduling region: the lanes m >= RH of the column-16 M-tile repeat row RH - 1, same address, same value)
                    const int xh = col >> 1;
                    unsigned char* dst = img + row * IROWB + ((col & 1) * XHR + xh) * 64 + 16 * ((2 * nt1 + (g >> 1) + ((xh >> 1) & 3)) & 3) + 8 * (g & 1);
                    *reinterpret_cast<uint2*>(dst) = make_uint2(pa[0], pb[0]);
                    *reinterpret_cast<uint2*>(dst + IPLANEB) = make_uint2(pa[1], pb[1]);
                }
            };
            rd1(0, 0, C::NST1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                // the first three steps of the next M-tile; this M-tile's first two steps with the first half of the previous epilogue in their shadows
                if (j + 1 < NJ) rd1(j + 1, 0, 3);
                mm1(j, 0, 2);
                if (j + 1 < NJ) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                __builtin_amdgcn_sched_barrier(0);
                // their slots are free: steps 3, 4 of the next M-tile; this M-tile's last three steps over the previous epilogue
                if (j + 1 < NJ) rd1(j + 1, 3, C::NST1);
                mm1(j, 2, C::NST1);
                if (j > 0) epi1(j - 1, (j - 1) & 1);
                if (j + 1 < NJ) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int q = 0; q < 9; q++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (j > 0) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
                if (j > 0) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            epi1(NJ - 1, (NJ - 1) & 1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this wave's image stores
        B42_T(2);
        __builtin_amdgcn_s_barrier();                                   // (raw barrier: a __syncthreads() would also drain the global stores of phase 2)
        asm volatile("" ::: "memory");
        if (tile + (int)gridDim.x < n_tiles) dma_issue(tile + gridDim.x);    // the patch is dead: the next tile's copy runs under phase 2
        B42_T(3);

        // ---- phase 2: block_4_3 from the LDS image.  This wave: channels 16 wave .. + 15, M-tiles j = 0, 1 = output rows 2 j, 2 j + 1.
        // One fragment set (18 x 4 registers): reads (0), MFMAs (0), reads (1) into the same registers, epilogue (0) under their latency, MFMAs (1), epilogue (1)
        {
            f16x8 fb[C::NST2][2];
            f32x4_m16 hi2[2], lo2[2];
            auto rd2 = [&](int j) {
                const unsigned char* ibase = img + (4 * j) * IROWB;          // image row 2 (2 j + (m >> 3)) with the lane offsets
#pragma unroll
                for (int st = 0; st < C::NST2; st++) {
                    const unsigned char* src = ibase + p2tap(st) + (st % 3 == 2 ? p2lane1 : p2lane0);
                    fb[st][0] = *reinterpret_cast<const f16x8*>(src);
                    fb[st][1] = *reinterpret_cast<const f16x8*>(src + IPLANEB);
                }
            };
            auto mm2 = [&](int j) {
                hi2[j] = f32x4_m16{bv3[0], bv3[1], bv3[2], bv3[3]};
                lo2[j] = f32x4_m16{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < C::NST2; st++) {
                    lo2[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[st][0], fb[st][1], lo2[j], 0, 0, 0);
                    lo2[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[st][1], fb[st][0], lo2[j], 0, 0, 0);
                    hi2[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[st][0], fb[st][0], hi2[j], 0, 0, 0);
                }
            };
            auto epi2 = [&](int j) {
                uint32_t pa[3], pb[3];
                s3p::act_split<2>(fmaf(hi2[j][0], S3_F16_SCALE, lo2[j][0]), fmaf(hi2[j][1], S3_F16_SCALE, lo2[j][1]), pa);
                s3p::act_split<2>(fmaf(hi2[j][2], S3_F16_SCALE, lo2[j][2]), fmaf(hi2[j][3], S3_F16_SCALE, lo2[j][3]), pb);
                const int oy = ty0 + 2 * j + (m >> 3), ox = tx0 + (m & 7);
                uint16_t* o = out16 + (((size_t)b * H3 + oy) * W3 + ox) * C::C3 + 16 * wave + 4 * g;
                *reinterpret_cast<uint2*>(o) = make_uint2(pa[0], pb[0]);
                *reinterpret_cast<uint2*>(o + o_plane) = make_uint2(pa[1], pb[1]);
            };
            rd2(0);
            __builtin_amdgcn_sched_barrier(0);
            mm2(0);
            __builtin_amdgcn_sched_barrier(0);
            rd2(1);
            __builtin_amdgcn_sched_barrier(0);
            epi2(0);
            __builtin_amdgcn_sched_barrier(0);
            mm2(1);
            __builtin_amdgcn_sched_barrier(0);
            epi2(1);
        }
        B42_T(4);
    }   // persistent tile loop
#ifdef HNET_B42_TRACE
    if (blockIdx.x < 8 && lane == 0) for (int k = 0; k < 5; k++) g_b42_trace[(blockIdx.x * 4 + wave) * 6 + k] = tr_acc[k];
    if (blockIdx.x < 8 && lane == 0) g_b42_trace[(blockIdx.x * 4 + wave) * 6 + 5] = (unsigned long long)(tile_no - 1);
#endif
}

}  // namespace hnet
