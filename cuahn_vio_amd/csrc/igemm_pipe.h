// igemm_pipe.h — software-pipelined implicit GEMM of the fp16-plane mode (round 4): operands go global -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds: no staging registers, no ds_write), two LDS stages of 64-deep K tiles, ONE barrier per K-tile placed
// BETWEEN the tile's two 32-deep MFMA groups, and every MFMA group runs on fragments that were read from LDS while the previous group
// was in the matrix pipe (two fragment register sets).  The waves of a workgroup therefore never sit in a common "load" or "store" phase:
// between two barriers a wave issues 15 - 30 MFMAs whose operands are already in registers, the ds_reads of the next group and (behind the
// barrier) the DMAs of the K-tile after the next, which get a whole K-tile of flight time.
//
//   prologue   DMA(0 -> stage 0); wait; barrier; DMA(1 -> stage 1); F0 <- stage 0, step 0
//   K-tile it  F1 <- stage cur, step 1 | MFMA(F0) | lgkmcnt(0), vmcnt(0): DMA(it + 1) has landed | barrier |
//              DMA(it + 2 -> stage cur)  (every wave finished reading stage cur before the barrier) | F0 <- stage cur ^ 1, step 0 | MFMA(F1)
//
// Same K order, same MFMA sequence per accumulator (hi += W0 A0; lo += W0 A1, W1 A0; result hi + lo / 4096) as igemm_s3_lean_kernel: results
// are bit-identical to it wherever that kernel runs without split-K.
//
// Tiles are made of whole frame pairs: the GEMM rows of one pair of these layers are 70 (7 x 10 outputs), 280 (14 x 20) or
// 20 (4 x 5); a workgroup owns BMV = 70 / 140 / 80 valid rows padded to BM = 80 / 160 / 80 (five 16-row MFMA tiles per wave), so that at
// batch 256 every layer is exactly 256 or 512 workgroups on the 256 CUs - the 128-row tiles of the eight-wave kernel gave 140 / 280 / 560.
// Rows BMV .. BM - 1 of a tile are zero rows (out-of-range DMA offsets) and are never stored.
#pragma once
#include "igemm_s3.h"

namespace hnet {

// TM, TN: 16 x 16 MFMA tiles per wave along M / N;  WVM x WVN waves;  BMV: valid GEMM rows per workgroup tile (<= BM = 16 TM WVM)
template <int TM_, int TN_, int WVM_, int WVN_, int BMV_>
struct PipeCfg {
    static constexpr int TM = TM_, TN = TN_, WVM = WVM_, WVN = WVN_;
    static constexpr int NWAVE = WVM * WVN, NT = 64 * NWAVE;
    static constexpr int BM = 16 * TM * WVM, BN = 16 * TN * WVN, BMV = BMV_, BK = 64;
    static constexpr int TILE_A = BM * BK, TILE_B = BN * BK;                 // halves per plane
    static constexpr int STAGE = 2 * (TILE_A + TILE_B);                      // halves per stage (two planes of each operand)
    static constexpr int LDS_BYTES = 2 * STAGE * 2;
    static constexpr int GA = BM / 8, GB = BN / 8, NG = GA + GB;             // DMA groups: 8 rows x 128 B = one wave-instruction per plane
    static constexpr int GPW = (NG + NWAVE - 1) / NWAVE;                     // groups per wave
    static_assert(BMV <= BM && LDS_BYTES <= 160 * 1024, "tile");
};

typedef void __attribute__((address_space(3))) * lds_ptr_t;

// phase timestamps for tools/trace_pipe.hip (compiled out of the library): [block < 8][wave < 8][S3T_SLOTS]
#ifdef HNET_S3_TRACE
#define PIPE_T()                                                                                                          \
    do {                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        if (p.trace && blockIdx.y == 0 && blockIdx.x < 8 && tcount < S3T_SLOTS) {                                                    \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                   \
            if (lane == 0) p.trace[(size_t)(blockIdx.x * 8 + wave) * S3T_SLOTS + tcount] = t_;                            \
            tcount++;                                                                                                     \
        }                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    } while (0)
#else
#define PIPE_T() do {} while (0)
#endif

template <class L, class C, bool OUT32, int NT_ = C::NT, int WPS_ = (C::NWAVE + 3) / 4>
__global__ __launch_bounds__(NT_, WPS_) void igemm_s3_pipe_kernel(S3Params p) {
    constexpr int TM = C::TM, TN = C::TN, BN = C::BN, BK = C::BK, NWAVE = C::NWAVE;
    constexpr int TILE_A = C::TILE_A, TILE_B = C::TILE_B, STAGE = C::STAGE, GA = C::GA, NG = C::NG, GPW = C::GPW;
    static_assert(L::template lean_ok<64>() && !L::HAS_MASK, "layers whose 64-deep K tiles lie inside one filter tap");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_p[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WVN, wn = wave % C::WVN;

    // workgroup -> tile: XCD x owns a contiguous range of the (M-tile major) tile order, the N-tiles of one M-tile adjacent (igemm_s3.h)
    int m0, n0;
    {
        const int nx = gridDim.x, ny = gridDim.y, total = nx * ny;
        const int lin = blockIdx.x + blockIdx.y * nx;
        const int xcd = lin & 7, idx = lin >> 3;
        const int base = total >> 3, rem = total & 7;
        const int Lt = xcd * base + min(xcd, rem) + idx;
        m0 = (Lt / ny) * C::BMV;
        n0 = (Lt % ny) * BN;
    }
    p.M = min(p.M, m0 + C::BMV);             // rows BMV .. BM - 1 of the tile do not exist: zero rows, never stored

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, 0x7FFFFFF0, 0x00020000);
    const int a_pl = (int)(p.a_plane * 2), w_pl = (int)(p.w_plane * 2);     // plane strides in bytes

    // ---- DMA groups of this wave: group g = wave + NWAVE j covers tile rows 8 g' .. 8 g' + 7 of A (g < GA) or of the weights
    // lane l -> row 8 g' + l / 8, PHYSICAL chunk l % 8 of the 128-byte LDS row; it fetches the LOGICAL chunk phys ^ ((row >> 1) & 7)
    // (the DMA writes LDS linearly: the swizzle of s3_swz_m16<8> is applied on the source side)
    LeanRow<L> arow[GPW];
    uint32_t wvoff[GPW];
#pragma unroll
    for (int j = 0; j < GPW; j++) {
        const int g = wave + NWAVE * j;
        const int gr = (g < GA ? g : g - GA) * 8 + (lane >> 3);
        const int lchunk = (lane & 7) ^ ((gr >> 1) & 7);
        arow[j] = LeanRow<L>::make(p, m0 + (g < GA ? gr : 0), n0, lchunk);
        if (g >= GA) arow[j].valid = false;
        const int n = n0 + gr;
        wvoff[j] = (g >= GA && g < NG && n < p.N) ? (uint32_t)((n * p.Kp + lchunk * 8) * 2) : S3_OOB;
    }
    auto dma = [&](int it, int stage) {        // `it`, `stage` wave-uniform
#if defined(HNET_PIPE_ABLATE) && HNET_PIPE_ABLATE == 91
        if (it > 1) return;                    // ablation (tools/trace_pipe.hip, wrong results): no DMA inside the K loop
#endif
        const typename LeanRow<L>::Tap t = LeanRow<L>::template tap<BK>(p, it);
        uint16_t* sb = smem_p + stage * STAGE;
#pragma unroll
        for (int j = 0; j < GPW; j++) {
            const int g = wave + NWAVE * j;
            if (g < GA) {
                const uint32_t vo = arow[j].voffset(p, t);
#pragma unroll
                for (int pl = 0; pl < 2; pl++)
                {
                    const int so = pl * a_pl;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(sb + pl * TILE_A + g * 512), 16, vo, so, 0, 0);
                }
            } else if (g < NG) {
#pragma unroll
                for (int pl = 0; pl < 2; pl++)
                {   // (locals: passing the array element / the sum directly makes hipcc 7.2 drop the HOST-side instantiation of the kernel without a diagnostic)
                    const uint32_t wv = wvoff[j];
                    const int so = it * (BK * 2) + pl * w_pl;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(sb + 2 * TILE_A + pl * TILE_B + (g - GA) * 512), 16, wv, so, 0, 0);
                }
            }
        }
    };

    // ---- fragment addresses: lane (r16, g16) reads row r16 of every 16-row tile, logical chunk 4 step + g16
    const int r16 = lane & 15, g16 = lane >> 4;
    const int key = (r16 >> 1) & 7;                    // (row >> 1) & 7 of every row this lane reads (tile origins are multiples of 16)
    int a_off[2], b_off[2];                            // halves, per 32-deep step
#pragma unroll
    for (int st = 0; st < 2; st++) {
        a_off[st] = (wm * TM * 16 + r16) * BK + ((4 * st + g16) ^ key) * 8;
        b_off[st] = 2 * TILE_A + (wn * TN * 16 + r16) * BK + ((4 * st + g16) ^ key) * 8;
    }
    bf16x8 fa[2][TM][2], fb[2][TN][2];                 // two fragment sets
    auto read_frags = [&](int set, int stage, int st) {
#if defined(HNET_PIPE_ABLATE) && HNET_PIPE_ABLATE == 93
        if (stage | st) return;                // ablation (wrong results): no fragment reads inside the K loop
#endif
        const uint16_t* sb = smem_p + stage * STAGE;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int pl = 0; pl < 2; pl++) fa[set][i][pl] = *reinterpret_cast<const bf16x8*>(&sb[a_off[st] + pl * TILE_A + i * 16 * BK]);
#pragma unroll
        for (int jn = 0; jn < TN; jn++)
#pragma unroll
            for (int pl = 0; pl < 2; pl++) fb[set][jn][pl] = *reinterpret_cast<const bf16x8*>(&sb[b_off[st] + pl * TILE_B + jn * 16 * BK]);
    };

    f32x4_m16 acc[TM][TN], accl[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int jn = 0; jn < TN; jn++) { acc[i][jn] = f32x4_m16{0.f, 0.f, 0.f, 0.f}; accl[i][jn] = f32x4_m16{0.f, 0.f, 0.f, 0.f}; }
    auto mfma = [&](int set) {
#if defined(HNET_PIPE_ABLATE) && HNET_PIPE_ABLATE == 92
        {                                      // ablation (wrong results): no MFMAs, the fragments stay alive
#pragma unroll
            for (int i = 0; i < TM; i++) asm volatile("" ::"v"(fa[set][i][0]), "v"(fa[set][i][1]));
#pragma unroll
            for (int jn = 0; jn < TN; jn++) asm volatile("" ::"v"(fb[set][jn][0]), "v"(fb[set][jn][1]));
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int jn = 0; jn < TN; jn++) {
                bf16x8 w3[3] = {fb[set][jn][0], fb[set][jn][1], fb[set][jn][0]}, a3[3] = {fa[set][i][0], fa[set][i][1], fa[set][i][0]};
                s3_mfma16_2acc(acc[i][jn], accl[i][jn], w3, a3);
            }
    };

    const int n_iter = p.Kp / BK;
    // Ping-pong: a DMA instruction takes the texture addresser 16 cycles (64 B / clk / CU) and its issue BLOCKS the wave while
    // the queue is full - eight waves issuing their ~10 DMAs together stood ~1200 cycles per K-tile with the matrix pipe idle (tools/trace_pipe.hip).
    // The two waves of a SIMD therefore issue half a K-tile apart: waves 0 .. NWAVE/2 - 1 ("X") right behind the barrier, into the stage that has
    // just become free (tile it + 2); the others ("Y") at the top of the next K-tile - while X issues, Y has the matrix pipe and vice versa.
    const bool y_half = NWAVE >= 8 && wave >= NWAVE / 2;
    dma(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (n_iter > 1 && !y_half) dma(1, 1);
    read_frags(0, 0, 0);

    // one ds_read per MFMA while there are reads (an MFMA holds the vector issue for 8 of its 16 cycles), then the rest of the MFMAs
#define HNET_PIPE_INTERLEAVE()                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < 2 * (TM + TN); k_++) {              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      \
    }                                                                           \
    __builtin_amdgcn_sched_group_barrier(0x008, 3 * TM * TN, 0);                \
    __builtin_amdgcn_sched_barrier(0)
#ifdef HNET_S3_TRACE
    int tcount = 0;
#endif
    for (int it = 0; it < n_iter - 1; it++) {
        const int cur = it & 1;
        PIPE_T();                                      // 0: K-tile start
        if (y_half) dma(it + 1, cur ^ 1);              // (it + 1 < n_iter inside this loop; stage cur ^ 1 has been free since the barrier of tile it - 1)
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1, cur, 1);
        mfma(0);
        HNET_PIPE_INTERLEAVE();
        PIPE_T();                                      // 1: first MFMA group + reads issued
        __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0): DMA(it + 1) of this wave has landed; lgkmcnt(0): its reads of stage cur are complete
        PIPE_T();                                      // 2: waits over
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PIPE_T();                                      // 3: past the barrier
        if (it + 2 < n_iter && !y_half) dma(it + 2, cur);
        __builtin_amdgcn_sched_barrier(0);
        PIPE_T();                                      // 4: DMAs issued
        read_frags(0, cur ^ 1, 0);
        mfma(1);
        HNET_PIPE_INTERLEAVE();
    }
    {   // last K-tile: nothing left to fetch
        read_frags(1, (n_iter - 1) & 1, 1);
        mfma(0);
        HNET_PIPE_INTERLEAVE();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        mfma(1);
    }
#undef HNET_PIPE_INTERLEAVE
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();                      // (the epilogue stages through LDS)
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int jn = 0; jn < TN; jn++) acc[i][jn] += accl[i][jn] * S3_F16_INV;

    // ---- epilogue: lane (em = lane & 15, eg = lane >> 4) holds channels 4 eg .. 4 eg + 3 of GEMM row em of every 16 x 16 tile
    const int em = lane & 15, eg = lane >> 4;
    const int mw = m0 + wm * TM * 16, nw = n0 + wn * TN * 16;
    if constexpr (OUT32) {
#pragma unroll
        for (int jn = 0; jn < TN; jn++) {
            const int n = nw + jn * 16 + 4 * eg;
            const f32x4_m16 bv = *reinterpret_cast<const f32x4_m16*>(p.bias + n);
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int m = mw + i * 16 + em;
                if (m < p.M) {
                    f32x4_m16 v = acc[i][jn];
#pragma unroll
                    for (int e = 0; e < 4; e++) { const float x = v[e] + bv[e]; v[e] = x > 0.0f ? x : x * 0.1f; }
                    *reinterpret_cast<f32x4_m16*>(p.out32 + (size_t)m * p.N + n) = v;
                }
            }
        }
    } else {
        // fp16 planes: per 16-row tile the wave's 16 x (16 TN) values go through wave-private LDS ([plane][16 rows][16 TN halves], 8-byte pieces in,
        // 16-byte pieces out) so that a lane stores 16 contiguous bytes and a row's 32 TN bytes are contiguous in memory
        static_assert(TN % 2 == 0 || TN == 1, "row pieces of 16 bytes");
        constexpr int RW = 16 * TN;                    // halves per staged row
        uint16_t* st = smem_p + wave * (2 * 16 * RW);
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int jn = 0; jn < TN; jn++) {
                const int n = nw + jn * 16 + 4 * eg;
                const f32x4_m16 bv = *reinterpret_cast<const f32x4_m16*>(p.bias + n);
                // bias, LeakyReLU and the plane split on value PAIRS (round 5: packed adds / multiplies, one v_max_f32 per value, the second plane by two
                // mixed-precision FMAs - s3p::split_pair<2>: 4.5 instead of ~ 9 vector instructions per value; the same bits as split2h per value)
                uint32_t pa[3], pb[3];
                {
                    const s3p::f32x2_p x0 = s3p::f32x2_p{acc[i][jn][0], acc[i][jn][1]} + s3p::f32x2_p{bv[0], bv[1]};
                    const s3p::f32x2_p x1 = s3p::f32x2_p{acc[i][jn][2], acc[i][jn][3]} + s3p::f32x2_p{bv[2], bv[3]};
                    const s3p::f32x2_p t0 = x0 * 0.1f, t1 = x1 * 0.1f;
                    s3p::split_pair<2>(s3p::vmax1(x0[0], t0[0]), s3p::vmax1(x0[1], t0[1]), pa);
                    s3p::split_pair<2>(s3p::vmax1(x1[0], t1[0]), s3p::vmax1(x1[1], t1[1]), pb);
                }
                // 16-byte chunk c = 2 jn + (eg >> 1) of row em, rotated by (em >> 1) so that the 16 lanes of a store group spread over the banks
                const int c = (2 * jn + (eg >> 1) + (em >> 1)) % (2 * TN);
#pragma unroll
                for (int pl = 0; pl < 2; pl++)
                    *reinterpret_cast<uint2*>(&st[(pl * 16 + em) * RW + c * 8 + (eg & 1) * 4]) = make_uint2(pa[pl], pb[pl]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            constexpr int PIECES = 2 * 16 * 2 * TN;    // 16-byte pieces of the two planes
#pragma unroll
            for (int q = 0; q < (PIECES + 63) / 64; q++) {
                const int piece = q * 64 + lane;
                const int pl = piece / (32 * TN), rem = piece % (32 * TN), row = rem / (2 * TN), ch = rem % (2 * TN);
                const int m = mw + i * 16 + row;
                if (piece < PIECES) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(&st[(pl * 16 + row) * RW + ((ch + (row >> 1)) % (2 * TN)) * 8]);
                    if (m < p.M) *reinterpret_cast<u32x4*>(p.out16 + pl * p.o_plane + (size_t)m * p.N + nw + ch * 8) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The heads' first FC (Dropout -> Linear(5120, 256) -> LeakyReLU for both heads and every MC sample; model_to_trace.py:222-225, 229-232) in the same
// pipelined form.  GEMM rows are (pair, sample): the 128 rows of a tile are 128 / n_local PAIRS, each repeated for its samples, and differ only in their
// dropout masks.  The eight-wave kernel of round 3 staged all 128 masked rows (32 KB per K-tile through the texture addresser and through ds_write).
// Here the A tile in LDS holds only the DISTINCT pairs of the M-tile (four rows at N = 32: 1 KB per K-tile by one DMA instruction), every lane reads
// its fragment from its pair's row (the 16 lanes of a read group hit the same address: a broadcast), and the keep bits are applied to the FRAGMENT:
// the tile's mask bytes ([128 rows][8 bytes] per K-tile, laid out K-tile major by heads_prep_kernel so that they are one 1 KB DMA) give, per lane,
// M-tile and 32-deep step, one byte = one entry of a 4 KB LDS table of 16-byte AND masks.  Traffic per K-tile: 34 KB instead of 64 KB, no ds_write.
//   mask layout (HEADS_MASK_KTILE): mask[((head * 80 + it) * M + m) * 8 + c], c = chunk of 8 elements inside K-tile it, bit e of the byte = element 8 c + e
// Same products in the same order per accumulator as igemm_s3_lean8_kernel<HeadLoaderS3>: bit-identical results.
// ---------------------------------------------------------------------------------------------
struct HeadsPipeCfg {
    // waves as 4 (M) x 2 (N), 32 x 64 each: a masked A fragment (8 vector ANDs) then feeds 12 MFMAs - with 64 x 32 per wave (6 MFMAs per fragment) the
    // ANDs and the MFMAs of the two waves of a SIMD filled 94 % of its vector issue (an MFMA holds it for 8 of its 16 cycles, a vector instruction for 4)
    static constexpr int BM = 128, BN = 128, BK = 64, NWAVE = 8, NT = 512, TM = 2, TN = 4, WVN = 2;
    static constexpr int TILE_A = BM * BK, TILE_B = BN * BK;                  // halves per plane (A: up to 128 distinct pairs)
    static constexpr int MASK_H = BM * 8 / 2;                                 // the mask tile, in halves
    static constexpr int STAGE = 2 * (TILE_A + TILE_B) + MASK_H;
    static constexpr int LUT_H = 256 * 8;
    static constexpr int LDS_BYTES = (2 * STAGE + LUT_H) * 2;
};

template <int NP>
__global__ __launch_bounds__(512, 2) void igemm_heads_pipe_kernel(S3Params p) {
    typedef HeadsPipeCfg C;
    constexpr int TM = C::TM, TN = C::TN, BM = C::BM, BN = C::BN, BK = C::BK, NWAVE = C::NWAVE;
    constexpr int TILE_A = C::TILE_A, TILE_B = C::TILE_B, STAGE = C::STAGE;
    static_assert(NP == 2, "fp16-plane mode");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_h[];
    uint16_t* const lut = smem_h + 2 * STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WVN, wn = wave % C::WVN;

    // workgroup -> tile, N-tile major: an XCD's contiguous range of tiles shares one 128-channel slab of the weights (2.6 MB: L2 resident)
    int m0, n0;
    {
        const int nx = gridDim.x, ny = gridDim.y, total = nx * ny;
        const int lin = blockIdx.x + blockIdx.y * nx;
        const int xcd = lin & 7, idx = lin >> 3;
        const int base = total >> 3, rem = total & 7;
        const int Lt = xcd * base + min(xcd, rem) + idx;
        m0 = (Lt % nx) * BM;
        n0 = (Lt / nx) * BN;
    }
    const int head = n0 >> 8;
    const int pair0 = m0 / p.n_local;
    const int m_last = min(m0 + BM, p.M) - 1;
    const int npair = m_last / p.n_local - pair0 + 1;               // distinct pairs of this M-tile (<= 128)
    const int n_iter = p.Kp / BK;

    if (tid < 256) {     // entry x of the table: 8 keep bits -> 8 x 16-bit lane masks
        u32x4 e;
#pragma unroll
        for (int j = 0; j < 4; j++) e[j] = ((tid >> (2 * j)) & 1u) * 0xFFFFu | ((tid >> (2 * j + 1)) & 1u) * 0xFFFF0000u;
        *reinterpret_cast<u32x4*>(&lut[tid * 8]) = e;
    }

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc((void*)p.mask, 0, p.M * 1280, 0x00020000);   // (reads beyond the last row's bytes return zeros)
    const int a_pl = (int)(p.a_plane * 2), w_pl = (int)(p.w_plane * 2);

    // ---- DMA: weights, two 8-row groups per wave (16 groups); the distinct pairs of A, group g by wave g % 8; the mask tile by wave 7
    const int grow = lane >> 3, gphys = lane & 7;
    uint32_t wvoff[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int r = (wave + NWAVE * j) * 8 + grow;
        wvoff[j] = (uint32_t)(((n0 + r) * p.Kp + (gphys ^ ((r >> 1) & 7)) * 8) * 2);
    }
    const int n_ag = (npair + 7) >> 3;                               // A groups (1 at N = 32)
    const int mask_it = p.M * 8;                                     // bytes per (head, K-tile) slab of the mask
    const uint32_t mvoff = (uint32_t)((head * n_iter) * (size_t)mask_it + (size_t)m0 * 8 + lane * 16);   // (< 2 GB: hnet_create bounds max_batch x N)
    auto dma = [&](int it, int stage) {
        uint16_t* sb = smem_h + stage * STAGE;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int pl = 0; pl < 2; pl++) {
                const uint32_t wv = wvoff[j];
                const int so = it * (BK * 2) + pl * w_pl;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(sb + 2 * TILE_A + pl * TILE_B + (wave + NWAVE * j) * 512), 16, wv, so, 0, 0);
            }
        for (int g = wave; g < n_ag; g += NWAVE) {                   // (wave-uniform trip count)
            const int r = g * 8 + grow;
            const uint32_t av = r < npair ? (uint32_t)(((pair0 + r) * 5120 + (gphys ^ ((r >> 1) & 7)) * 8) * 2) : S3_OOB;
#pragma unroll
            for (int pl = 0; pl < 2; pl++) {
                const int so = it * (BK * 2) + pl * a_pl;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(sb + pl * TILE_A + g * 512), 16, av, so, 0, 0);
            }
        }
        if (wave == NWAVE - 1) {
            const int so = it * mask_it;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rM, (lds_ptr_t)(sb + 2 * (TILE_A + TILE_B)), 16, mvoff, so, 0, 0);
        }
    };

    // ---- fragment addresses
    const int r16 = lane & 15, g16 = lane >> 4;
    int a_off[2][TM], b_off[2], m_off[TM];                           // halves (a, b), bytes (m)
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int ml = wm * TM * 16 + i * 16 + r16;                  // row of the tile
        const int pj = min(m0 + ml, m_last) / p.n_local - pair0;     // its pair's row in the A tile
#pragma unroll
        for (int st = 0; st < 2; st++) a_off[st][i] = pj * BK + ((4 * st + g16) ^ ((pj >> 1) & 7)) * 8;
        m_off[i] = ml * 8 + g16;
    }
    {
        const int key = (r16 >> 1) & 7;
#pragma unroll
        for (int st = 0; st < 2; st++) b_off[st] = 2 * TILE_A + (wn * TN * 16 + r16) * BK + ((4 * st + g16) ^ key) * 8;
    }
    bf16x8 fa[2][TM][2], fb[2][TN][2];
    uint32_t mb[2][TM];                                              // keep byte of (M-tile i, this lane's chunk) per set
    u32x4 mk[2][TM];                                                 // its 16-byte AND mask
    // the reads of the next fragment set in three parts (scheduled under the MFMA groups of the current one): keep bytes + A rows | weights 0, 1 | weights 2, 3
    auto read_part = [&](int set, int stage, int st, int part) {
        const uint16_t* sb = smem_h + stage * STAGE;
        if (part == 0) {
            const uint8_t* mt = reinterpret_cast<const uint8_t*>(sb + 2 * (TILE_A + TILE_B));
#pragma unroll
            for (int i = 0; i < TM; i++) mb[set][i] = mt[m_off[i] + 4 * st];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int pl = 0; pl < 2; pl++) fa[set][i][pl] = *reinterpret_cast<const bf16x8*>(&sb[a_off[st][i] + pl * TILE_A]);
        } else {
#pragma unroll
            for (int jn = 2 * (part - 1); jn < 2 * part; jn++)
#pragma unroll
                for (int pl = 0; pl < 2; pl++) fb[set][jn][pl] = *reinterpret_cast<const bf16x8*>(&sb[b_off[st] + pl * TILE_B + jn * 16 * BK]);
        }
    };
    auto read_frags = [&](int set, int stage, int st) { read_part(set, stage, st, 0); read_part(set, stage, st, 1); read_part(set, stage, st, 2); };
    auto read_lut = [&](int set) {                                   // keep byte -> 16-byte AND mask (4 KB LDS table)
#pragma unroll
        for (int i = 0; i < TM; i++) mk[set][i] = *reinterpret_cast<const u32x4*>(&lut[mb[set][i] * 8]);
    };
    auto and_row = [&](int set, int i) {                             // both planes of the fragment of M-tile i
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            u32x4 v = __builtin_bit_cast(u32x4, fa[set][i][pl]);
            v[0] &= mk[set][i][0]; v[1] &= mk[set][i][1]; v[2] &= mk[set][i][2]; v[3] &= mk[set][i][3];
            fa[set][i][pl] = __builtin_bit_cast(bf16x8, v);
        }
    };

    f32x4_m16 acc[TM][TN], accl[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int jn = 0; jn < TN; jn++) { acc[i][jn] = f32x4_m16{0.f, 0.f, 0.f, 0.f}; accl[i][jn] = f32x4_m16{0.f, 0.f, 0.f, 0.f}; }
    auto mfma_part = [&](int set, int i, int j0) {                 // two of the four 16-channel tiles of M-tile i
#pragma unroll
        for (int jn = j0; jn < j0 + 2; jn++) {
            bf16x8 w3[3] = {fb[set][jn][0], fb[set][jn][1], fb[set][jn][0]}, a3[3] = {fa[set][i][0], fa[set][i][1], fa[set][i][0]};
            s3_mfma16_2acc(acc[i][jn], accl[i][jn], w3, a3);
        }
    };
    // One half of a K-tile: the 24 MFMAs of fragment set `cur` (whose table masks are in registers) in four groups of six with, underneath them, the AND
    // of the second row, the reads of the next set (2 keep bytes + 12 fragments) and, once its keep bytes are back, its table lookups.
#define HNET_HEADS_HALF(cur, nxt, stage, st, WITH_NEXT)                                         \
    do {                                                                                        \
        and_row(cur, 0);                                                                        \
        if (WITH_NEXT) read_part(nxt, stage, st, 0);                                            \
        mfma_part(cur, 0, 0);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        if (WITH_NEXT) read_part(nxt, stage, st, 1);                                            \
        mfma_part(cur, 0, 2);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        and_row(cur, 1);                                                                        \
        if (WITH_NEXT) read_part(nxt, stage, st, 2);                                            \
        mfma_part(cur, 1, 0);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        if (WITH_NEXT) read_lut(nxt);                     /* the keep bytes came back long ago */ \
        mfma_part(cur, 1, 2);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)

    const bool y_half = wave >= NWAVE / 2;                           // (ping-pong of the DMA issue: igemm_s3_pipe_kernel)
    dma(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();                                    // (also: the table is written)
    __builtin_amdgcn_sched_barrier(0);
    if (n_iter > 1 && !y_half) dma(1, 1);
    read_frags(0, 0, 0);
    read_lut(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_sched_barrier(0);

    for (int it = 0; it < n_iter - 1; it++) {
        const int cur = it & 1;
        if (y_half) dma(it + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        HNET_HEADS_HALF(0, 1, cur, 1, true);
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (it + 2 < n_iter && !y_half) dma(it + 2, cur);
        __builtin_amdgcn_sched_barrier(0);
        HNET_HEADS_HALF(1, 0, cur ^ 1, 0, true);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        HNET_HEADS_HALF(0, 1, (n_iter - 1) & 1, 1, true);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_sched_barrier(0);
        HNET_HEADS_HALF(1, 0, 0, 0, false);
    }
#undef HNET_HEADS_HALF

#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int jn = 0; jn < TN; jn++) acc[i][jn] += accl[i][jn] * S3_F16_INV;

    // ---- epilogue: bias + LeakyReLU, fp32 [M][N]; lane (em, eg) holds channels 4 eg .. 4 eg + 3 of row em of every 16 x 16 tile
    const int em = lane & 15, eg = lane >> 4;
    const int mw = m0 + wm * TM * 16, nw = n0 + wn * TN * 16;
#pragma unroll
    for (int jn = 0; jn < TN; jn++) {
        const int n = nw + jn * 16 + 4 * eg;
        const f32x4_m16 bv = *reinterpret_cast<const f32x4_m16*>(p.bias + n);
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int m = mw + i * 16 + em;
            if (m < p.M) {
                f32x4_m16 v = acc[i][jn];
#pragma unroll
                for (int e = 0; e < 4; e++) { const float x = v[e] + bv[e]; v[e] = x > 0.0f ? x : x * 0.1f; }
                *reinterpret_cast<f32x4_m16*>(p.out32 + (size_t)m * p.N + n) = v;
            }
        }
    }
}

}  // namespace hnet
