// igemm_region.h — implicit GEMM of the stride-2 layers whose tiles are bound by OPERAND DELIVERY (round 4): the input region of the tile's frame
// pairs lives in LDS (one 64-channel chunk at a time, double buffered), the MFMA fragments of every filter tap are read straight from it, and the
// weights go from L2 STRAIGHT INTO REGISTERS as pre-packed MFMA fragments.  No staging tiles, no barrier inside the K loop.
//
// Why.  The tiled kernels (igemm_s3_lean_kernel, igemm_pipe.h) stage an im2col tile per K-tile, i.e. they fetch every activation 2.25 (3 x 3, stride 2) or
// 6.25 (5 x 5) times through the texture path, which delivers ~38 B / clk / CU (76 GB/s per CU, tools/trace_pipe.hip).  An 80-row tile needs 55 - 77 B / clk /
// CU to keep the matrix pipe busy, so block_1_2 (70 rows per pair) and the 4 x 5 layers sat on that limit in BOTH kernel families
// (profiles/r04_experiments_not_shipped.log, item 2).  With the region resident each activation crosses the texture path once per workgroup:
//   block_1_2, per workgroup (one pair, 128 channels):   2.6 MB -> 1.78 MB   (A: 50 x 20 KB of im2col tiles -> 143 KB of region; weights 1.64 MB either way)
//   block_2_4 / 3_5 / 4_6, per workgroup (4 pairs x 64): 1.3 MB -> 0.88 MB
// A first build kept the weight ring of igemm_pipe.h in LDS: two stages were all that fitted next to the region, and an 80 x 128 tile's K-tile (960 matrix
// cycles) is shorter than a DMA round trip, so the ring ran dry (0.090 ms against 0.072 for the lean kernel).  Every wave owns ONE 16-channel N-tile here, so
// a weight fragment is needed by exactly one wave: it loads its own (hnet_create packs them in fragment order: a wave-instruction is 1 KB contiguous),
// 2.5 K-tiles ahead in registers (a ring of five 32-deep steps), and the waves of a workgroup never wait for each other except at a chunk switch.
//
// K order: 64-channel chunk major, taps minor (the region of a chunk serves all KS x KS taps).  Another summation order than the tap-major kernels: same
// products, results agree to fp32 rounding (goldens + element-wise test).  KSPLIT (the 4 x 5 layers, N-tile 64): waves 0 - 3 take the first taps of every
// chunk, waves 4 - 7 the rest, for the same four N-tiles; the two partial accumulators are added through LDS in a fixed order.
//
// LDS: two region buffers [2 planes][RP + 16 rows][64 halves]; a row = one region pixel, 128 bytes = eight 16-byte pieces; rows RP .. RP + 15 are zeros
// (padding taps and rows beyond M read zero row RP + (T & 15): the row of their own bank class, so that a fragment with padding lanes stays conflict-free).  The LDS-DMA fetches any 64 pieces per instruction, so the layout is free; it is chosen for the fragment
// reads: the pixels are split into the four (y & 1, x & 1) images (a fragment read touches ONE of them: stride 2), pixel (pair, y, x) is row
//   q = base[image] + pair PS[image] + (y >> 1) Wo + (x >> 1),      T = pair Ho Wo + (y >> 1) Wo + (x >> 1)       (W / 2 = Wo, (H + 1) / 2 = Ho)
// so that the 16 GEMM rows m .. m + 15 of a fragment (consecutive output pixels, across output rows and pairs) read rows with CONSECUTIVE T, and piece c of
// the row sits at position c ^ 2 ((T >> 1) & 3).  ds_read_b128 serves 16 lanes per LDS cycle - rows {0-3, 12-15} of one piece with rows {4-11} of its
// neighbour piece -: rows T and T + 8 share the XOR and the parity of q (PS even) and are in different halves of that split, every other pair of rows differs
// in one of the two: conflict-free for any first row (a first build with rows in raster order of a parity-split line was 2 - 3-way conflicted wherever a
// fragment crossed an output row: tools/trace_region.hip, the reads alone took longer than the MFMAs).
#pragma once
#include "igemm_pipe.h"

namespace hnet {

// P pairs per tile (P x Ho x Wo <= 80 rows); RP = region rows allocated (>= the four images' 2 P (even(Ho Wo) + even((H / 2) Wo)), multiple of 8); KSPLIT: 64-channel N-tiles, K halves over the wave halves
// HI_ x WI_: the layer's input size (round 5: compile-time - the divisions by Wo, Ho Wo and the image sizes of the one-time address set-up were a thousand vector
// instructions per workgroup, more than the K loop's own; s3_dispatch.h region_ok checks the launch against them)
template <int CIN_, int KS_, int P_, int RP_, bool KSPLIT_, int HI_, int WI_>
struct RegionCfg {
    static constexpr int CIN = CIN_, KS = KS_, P = P_, RP = RP_, PAD = (KS_ - 1) / 2;
    static constexpr int HI = HI_, WI = WI_, HO = (HI_ + 1) / 2, WO = WI_ / 2;
    static_assert(WI_ % 2 == 0, "the parity images assume an even input width");
    static constexpr bool KSPLIT = KSPLIT_;
    static constexpr int TM = 5, BM = 80, BN = KSPLIT_ ? 64 : 128, NWAVE = 8, NT = 512;
    static constexpr int NCHUNK = CIN / 64, NTAP = KS * KS;
    static constexpr int TAPS_X = KSPLIT_ ? (NTAP + 1) / 2 : NTAP;           // taps per chunk of a wave; KSPLIT: the second half's list is padded with a zero-weight tap
    static constexpr int NTAP_PAD = KSPLIT_ ? 2 * TAPS_X : NTAP;             // K-tiles per chunk in the packed weights (taps >= NTAP: zeros)
    static constexpr int DEPTH = 5;                                          // 32-deep STEPS of weight fragments in flight per wave (2.5 K-tiles, 40 VGPRs); divides the 2 TAPS_X steps of a chunk (slot = step % DEPTH in every chunk)
    static_assert((2 * TAPS_X) % DEPTH == 0, "weight-fragment ring");
    static constexpr int REG_PLANE = (RP + 16) * 64;                         // halves per plane
    static constexpr int REG_BUF = 2 * REG_PLANE;                            // halves per region buffer (two planes)
    static constexpr int LDS_BYTES = 2 * REG_BUF * 2;
    static constexpr int RG = RP / 8;                                        // region DMA groups (8 rows each)
    static_assert(RP % 8 == 0 && LDS_BYTES <= 160 * 1024 && CIN % 64 == 0, "region");
    // packed weights (hnet_create): [N / 16][NCHUNK][NTAP_PAD][2 steps][2 planes][64 lanes][8 halves]
    static constexpr size_t wfrag_halves(int n) { return (size_t)(n / 16) * NCHUNK * NTAP_PAD * 2 * 2 * 64 * 8; }
};

template <class C, bool OUT32>
__global__ __launch_bounds__(512, 2) void igemm_s3_region_kernel(S3Params p) {
    constexpr int TM = C::TM, BN = C::BN, NWAVE = C::NWAVE, CIN = C::CIN, KS = C::KS, PAD = C::PAD;
    constexpr int REG_PLANE = C::REG_PLANE, REG_BUF = C::REG_BUF, NTAP = C::NTAP, NCHUNK = C::NCHUNK;
    constexpr int DEPTH = C::DEPTH, NTAP_PAD = C::NTAP_PAD;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_r[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = C::KSPLIT ? wave >> 2 : 0, wn = C::KSPLIT ? wave & 3 : wave;
    constexpr int R = C::HO * C::WO, HW = C::HI * C::WI;
    const int rows_tile = C::P * R;                              // valid GEMM rows of a tile (<= 80)
    // workgroup -> tile (XCD-aware: an XCD's contiguous range of tiles shares weights AND neighbouring pairs)
    int mt, n0;
    {
        const int nx = gridDim.x, ny = gridDim.y, total = nx * ny;
        const int lin = blockIdx.x + blockIdx.y * nx;
        const int xcd = lin & 7, idx = lin >> 3;
        const int base = total >> 3, rem = total & 7;
        const int Lt = xcd * base + min(xcd, rem) + idx;
        mt = Lt / ny;
        n0 = (Lt % ny) * BN;
    }
    const int m0 = mt * rows_tile, pair0 = mt * C::P;
    const int m_end = min(p.M, m0 + rows_tile);
    const int px_total = (p.M / R) * HW;                         // pixels of the whole batch (beyond: out of range = zeros)

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, 0x7FFFFFF0, 0x00020000);
    const int a_pl = (int)(p.a_plane * 2);

    // the zero rows of both buffers (never a DMA target)
    {   // 16 rows x 128 B x 2 planes x 2 buffers = 512 pieces of 16 bytes
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(&smem_r[(tid >> 8) * REG_BUF + ((tid >> 7) & 1) * REG_PLANE + C::RP * 64 + (tid & 127) * 8]) = z;
    }

    // ---- region DMA: groups g = wave, wave + 8, ... (8 region rows x 128 B each)
    const int grow = lane >> 3, gphys = lane & 7;
    constexpr int RGW = (C::RG + NWAVE - 1) / NWAVE;
    // the four images: sizes per pair (even), first rows
    constexpr int Wo = C::WO, hh_e = (C::HI + 1) >> 1, hh_o = C::HI >> 1;
    constexpr int ps_e = (hh_e * Wo + 1) & ~1, ps_o = (hh_o * Wo + 1) & ~1;
    constexpr int base2 = 2 * C::P * ps_e;                           // first row of the odd-y images
    uint32_t rvoff[RGW];                                         // byte offset of this lane's 16 bytes of chunk 0 (S3_OOB: no pixel there / beyond the batch)
#pragma unroll
    for (int j = 0; j < RGW; j++) {
        const int q = (wave + NWAVE * j) * 8 + grow;             // LDS row; this lane fills its piece gphys
        const int yo = q >= base2, qq = yo ? q - base2 : q, ps = yo ? ps_o : ps_e, hh = yo ? hh_o : hh_e;
        const int xo = qq >= C::P * ps, q3 = xo ? qq - C::P * ps : qq;
        const int pl_ = yo ? q3 / ps_o : q3 / ps_e;                  // (divisions by constants: two multiplications and a select instead of a run-time division)
        const int t = q3 - pl_ * ps, Y = t / Wo, xh = t - Y * Wo;
        const int T = pl_ * R + t;
        const int gq = (pair0 + pl_) * HW + (2 * Y + yo) * C::WI + 2 * xh + xo;
        const bool ok = q < base2 + 2 * C::P * ps_o && t < hh * Wo && gq < px_total;
        rvoff[j] = ok ? (uint32_t)((gq * CIN + (gphys ^ (2 * ((T >> 1) & 3))) * 8) * 2) : S3_OOB;
    }
    auto dma_region = [&](int c) {                               // chunk c -> buffer c & 1
#pragma unroll
        for (int j = 0; j < RGW; j++) {
            const int g = wave + NWAVE * j;
            if (g < C::RG) {
                const uint32_t vo = rvoff[j];
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    const int so = c * 128 + pl * a_pl;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(smem_r + (c & 1) * REG_BUF + pl * REG_PLANE + g * 512), 16, vo, so, 0, 0);
                }
            }
        }
    };

    // ---- weight fragments: this wave's N-tile, K-tile (chunk c, tap t): 4 KB contiguous = [2 steps][2 planes][64 lanes] x 16 B
    const uint32_t wlane = (uint32_t)(lane * 16);
    const int wtile0 = ((n0 >> 4) + wn) * (NCHUNK * NTAP_PAD);   // K-tile index of (this N-tile, chunk 0, tap 0)
    const int t_lo_w = khalf * C::TAPS_X;                        // this wave's first tap of every chunk
    bf16x8 fw[DEPTH][2];
    auto load_w = [&](int slot, int c, int sidx) {               // slot compile-time; c, sidx (step of this wave's part of the chunk: tap sidx / 2, half sidx & 1) wave-uniform
#if defined(HNET_REGION_ABLATE) && HNET_REGION_ABLATE == 1
        if (c | (sidx >= DEPTH)) return;                         // ablation (tools/trace_region.hip, wrong results): no weight loads inside the K loop
#endif
        const int so = (wtile0 + c * NTAP_PAD + t_lo_w) * 4096 + sidx * 2048;
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const int so2 = so + pl * 1024;
            fw[slot][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, wlane, so2, 0));
        }
    };

    // ---- A fragment addressing.  Round 5: everything about a (tap, row) address that does not depend on the lane is a scalar of the tap - input pixel
    // (y, x) = (2 oy - PAD + kh, 2 ox - PAD + kw), so its parity image is (kh - PAD) & 1, (kw - PAD) & 1 and its position inside the image is the row's own
    // (oy, ox) shifted by dy = (kh - PAD) >> 1, dx = (kw - PAD) >> 1 - and the lane keeps three sums per M-tile row (byte address of output pixel (oy, ox) in
    // the even-y and in the odd-y images, its T) plus one word of validity bits (which kh / kw fall inside the image).  Eleven vector instructions per
    // (tap, row) instead of ~ 20 (the kernel was vector-issue bound: 4.6 - 8.0 VALU per MFMA, profiles/r04_v4_pmc_mfma_lds.csv).  Same addresses.
    const int r16 = lane & 15, g16 = lane >> 4;
    int qe_b[TM], qo_b[TM], tt0[TM];                             // x 128 bytes: row of (pair, oy, ox) in the (even y, even x) / (odd y, even x) image; T of (pair, oy, ox)
    uint32_t okb[TM];                                            // bit kh: y inside the image; bit 8 + kw: x inside; 0 for rows beyond the tile
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int ml = i * 16 + r16;                             // row of the tile
        const bool ok = m0 + ml < m_end;
        const int pl_ = ml / R, rem = ml - pl_ * R, oy = rem / Wo, ox = rem - oy * Wo;
        const int t0 = oy * Wo + ox;
        qe_b[i] = (pl_ * ps_e + t0) * 128;
        qo_b[i] = (base2 + pl_ * ps_o + t0) * 128;
        tt0[i] = pl_ * R + t0;                                   // (rows beyond the tile read zero rows - of their own bank class: T stays linear in the row)
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < KS; k++) {
            bits |= ((unsigned)(2 * oy - PAD + k) < (unsigned)C::HI) ? (1u << k) : 0u;
            bits |= ((unsigned)(2 * ox - PAD + k) < (unsigned)C::WI) ? (1u << (8 + k)) : 0u;
        }
        okb[i] = ok ? bits : 0u;
    }
    bf16x8 fa[2][TM][2];
    int a_byte[TM];                                              // byte offset (inside a plane of a buffer) of this lane's step-0 chunk of the current tap
    constexpr int xoff_e = C::P * ps_e * 128, xoff_o = C::P * ps_o * 128;      // odd-x image behind the even-x image of the same y parity
    auto tap_addr = [&](int t) {                                 // t wave-uniform; t >= NTAP (the padding tap of KSPLIT): the zero row
#if defined(HNET_REGION_ABLATE) && HNET_REGION_ABLATE == 2
        if (t != t_lo_w) return;                                 // ablation (wrong results): one address computation per chunk
#endif
        // scalars of the tap
        const int kh = t / KS, kw = t - kh * KS;
        const int ey = kh - PAD, ex = kw - PAD;
        const int yo = ey & 1, xo = ex & 1, dt = (ey >> 1) * Wo + (ex >> 1);
        const int s_b = (xo ? (yo ? xoff_o : xoff_e) : 0) + dt * 128;
        const uint32_t need = t < NTAP ? ((1u << kh) | (1u << (8 + kw))) : 0xFFFFFFFFu;      // (the padding tap: never satisfied)
#pragma unroll
        for (int i = 0; i < TM; i++) {
            // (opaque to the optimiser: the addresses of a tap are the same in every chunk, and hoisting all KS x KS x TM of them out of the chunk loop spills)
            int t0_ = tt0[i], qs_ = yo ? qo_b[i] : qe_b[i];
            asm volatile("" : "+v"(t0_), "+v"(qs_));
            const int T = t0_ + dt;
            const bool ok = (okb[i] & need) == need;
            const int qb = ok ? qs_ + s_b : (C::RP + (T & 15)) * 128;
            a_byte[i] = qb | ((g16 ^ (T & 6)) << 4);
        }
    };
    auto read_a = [&](int set, int buf, int st) {                // st = 32-deep step of the K-tile; a_byte holds the tap's addresses
#if defined(HNET_REGION_ABLATE) && HNET_REGION_ABLATE == 4
        if (buf | st) return;                                    // ablation (wrong results): (almost) no fragment reads
#endif
        const unsigned char* rg = reinterpret_cast<const unsigned char*>(smem_r + buf * REG_BUF);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
                fa[set][i][pl] = *reinterpret_cast<const bf16x8*>(rg + pl * (REG_PLANE * 2) + (a_byte[i] ^ (st << 6)));
    };

    f32x4_m16 acc[TM], accl[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) { acc[i] = f32x4_m16{0.f, 0.f, 0.f, 0.f}; accl[i] = f32x4_m16{0.f, 0.f, 0.f, 0.f}; }
    auto mfma = [&](int set, int slot) {
#if defined(HNET_REGION_ABLATE) && HNET_REGION_ABLATE == 3
        {                                                        // ablation (wrong results): no MFMAs, the fragments stay alive
#pragma unroll
            for (int i = 0; i < TM; i++) asm volatile("" ::"v"(fa[set][i][0]), "v"(fa[set][i][1]));
            asm volatile("" ::"v"(fw[slot][0]), "v"(fw[slot][1]));
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < TM; i++) {
            bf16x8 w3[3] = {fw[slot][0], fw[slot][1], fw[slot][0]}, a3[3] = {fa[set][i][0], fa[set][i][1], fa[set][i][0]};
            s3_mfma16_2acc(acc[i], accl[i], w3, a3);
        }
    };

    // this wave's taps of every chunk: t_lo .. t_lo + TAPS_X - 1
    constexpr int NJ = C::TAPS_X, NS = 2 * NJ;
    const int t_lo = t_lo_w;

    // ---- prologue: both region buffers on their way, the first weight fragments in flight
    dma_region(0);
    if (NCHUNK > 1) dma_region(1);
#pragma unroll
    for (int d = 0; d < DEPTH; d++) load_w(d, 0, d);
    __builtin_amdgcn_s_waitcnt(0x0070);                          // (vmcnt(0): this wave's region DMAs have landed; the fragment loads too - once)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    for (int c = 0; c < NCHUNK; c++) {
        const int buf = c & 1;
        tap_addr(t_lo);
        read_a(0, buf, 0);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            read_a(1, buf, 1);
            mfma(0, (2 * j) % DEPTH);
            __builtin_amdgcn_sched_barrier(0);
            // refill the slot: the step DEPTH ahead in this wave's sequence (a later one of this chunk, or one of the next chunk's first DEPTH)
            if (2 * j + DEPTH < NS) load_w((2 * j) % DEPTH, c, 2 * j + DEPTH);
            else if (c + 1 < NCHUNK) load_w((2 * j) % DEPTH, c + 1, 2 * j + DEPTH - NS);
            if (j + 1 < NJ) {                                    // addresses + step-0 fragments of the next tap of this chunk
                tap_addr(t_lo + j + 1);
                read_a(0, buf, 0);
            }
            mfma(1, (2 * j + 1) % DEPTH);
            __builtin_amdgcn_sched_barrier(0);
            if (2 * j + 1 + DEPTH < NS) load_w((2 * j + 1) % DEPTH, c, 2 * j + 1 + DEPTH);
            else if (c + 1 < NCHUNK) load_w((2 * j + 1) % DEPTH, c + 1, 2 * j + 1 + DEPTH - NS);
        }
        if (c + 1 < NCHUNK) {
            // chunk switch: this wave's reads of buffer `buf` are complete and its part of the region DMA issued at the previous switch has landed; behind the
            // barrier that holds for every wave: chunk c + 2 may overwrite `buf`, chunk c + 1's buffer is complete.  vmcnt(2 DEPTH): everything but the
            // weight fragments in flight (issued after that DMA) - the ring stays full across the switch
            static_assert(DEPTH == 5, "s_waitcnt immediate: vmcnt(10) lgkmcnt(0)");
            __builtin_amdgcn_s_waitcnt(0x007A);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 < NCHUNK) dma_region(c + 2);
        }
    }

#pragma unroll
    for (int i = 0; i < TM; i++) acc[i] += accl[i] * S3_F16_INV;

    if constexpr (C::KSPLIT) {                                   // the two K halves of an N-tile: waves 4 - 7 hand theirs over through LDS (the regions are dead)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        float* red = reinterpret_cast<float*>(smem_r);           // [4 N-tiles][TM][256 floats]
        if (khalf) {
#pragma unroll
            for (int i = 0; i < TM; i++) *reinterpret_cast<f32x4_m16*>(&red[(wn * TM + i) * 256 + lane * 4]) = acc[i];
        }
        __syncthreads();
        if (khalf) return;
#pragma unroll
        for (int i = 0; i < TM; i++) acc[i] += *reinterpret_cast<const f32x4_m16*>(&red[(wn * TM + i) * 256 + lane * 4]);
    }

    // ---- epilogue: lane (em, eg) holds channels 4 eg .. 4 eg + 3 of row em of every M-tile, for the wave's 16-channel N-tile
    const int em = lane & 15, eg = lane >> 4;
    const int n = n0 + wn * 16 + 4 * eg;
    const f32x4_m16 bv = *reinterpret_cast<const f32x4_m16*>(p.bias + n);
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m = m0 + i * 16 + em;
        f32x4_m16 v = acc[i];
        {   // (packed: bias add and the 0.1 multiple on value pairs, one v_max_f32 per value; x > 0 ? x : 0.1 x == max(x, 0.1 x) for every non-NaN x)
            const s3p::f32x2_p x0 = s3p::f32x2_p{v[0], v[1]} + s3p::f32x2_p{bv[0], bv[1]}, x1 = s3p::f32x2_p{v[2], v[3]} + s3p::f32x2_p{bv[2], bv[3]};
            const s3p::f32x2_p t0 = x0 * 0.1f, t1 = x1 * 0.1f;
            v[0] = s3p::vmax1(x0[0], t0[0]); v[1] = s3p::vmax1(x0[1], t0[1]); v[2] = s3p::vmax1(x1[0], t1[0]); v[3] = s3p::vmax1(x1[1], t1[1]);
        }
        if (m < m_end) {
            if constexpr (OUT32) {
                *reinterpret_cast<f32x4_m16*>(p.out32 + (size_t)m * p.N + n) = v;
            } else {
                uint32_t pa[3], pb[3];
                s3p::split_pair<2>(v[0], v[1], pa);
                s3p::split_pair<2>(v[2], v[3], pb);
#pragma unroll
                for (int pl = 0; pl < 2; pl++)
                    *reinterpret_cast<uint2*>(p.out16 + pl * p.o_plane + (size_t)m * p.N + n) = make_uint2(pa[pl], pb[pl]);
            }
        }
    }
}

}  // namespace hnet
