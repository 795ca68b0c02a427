// igemm_s3_pc.h — wave-specialised split-bf16 implicit GEMM: one PRODUCER wave feeds an LDS ring with
// global_load_lds (LDS-DMA), four CONSUMER waves do nothing but ds_read_b128 + v_mfma_f32_32x32x16_bf16.
//
// Why (tools/trace_s3.hip, profiles/r01): in igemm_s3.h every wave alternates between issuing its 18-22 global loads,
// its 48 MFMAs and its LDS stores.  A CDNA wave issues in order: while its vector-memory instruction waits for a slot
// in the texture addresser (64 B/clk per CU, shared by all resident waves, which all reach their load phase together)
// the MFMAs behind it cannot issue.  Measured per K-tile of the heads GEMM: 1340 cycles issuing loads, 2020 in the
// MFMAs, 2080 waiting for / storing the staged registers, 700 in barriers: the matrix pipe is busy a third of the time.
// Here the address arithmetic, the load issue and the back-pressure all live in the producer wave; the consumers never
// touch vector memory, their only waits are the LDS fragment reads and one barrier per K-tile.
// tools/pc_gemm_proto.hip (timing prototype, heads shape): 0.230 ms against 0.305 ms, consumer-only bound 0.156 ms.
//
// Tile 128 x 128, K-tile 32, 3 ring stages of 48 KiB (one workgroup of 320 threads per CU); consumer wave (wm, wn) owns
// a 64 x 64 sub-tile (2 x 2 MFMA tiles: 12 fragment reads per 24 MFMAs).  The producer waits for K-tile it (vmcnt),
// then the single s_barrier of the iteration tells the consumers that tile it has landed and tells the producer that
// the fragments of tile it-1 are in registers (the consumers pass that barrier half way through tile it-1), whose stage
// it refills with tile it+NSTAGE-1.
//
// MC-dropout masks (heads): the DMA cannot mask, so the producer also copies the keep bytes of the tile (one dword per
// row and K-tile) into the ring and the consumers AND their A fragments with the expanded bits.
#pragma once
#include <hip/hip_runtime.h>
#include "igemm_s3.h"

namespace hnet {

template <int BM, int BN, int NSTAGE, bool HAS_MASK> struct PcCfg {
    static constexpr int BK = 32;
    static constexpr int TILE_A = BM * BK, TILE_B = BN * BK;                 // bf16 elements per plane
    static constexpr int STAGE = 3 * (TILE_A + TILE_B);                      // elements per ring stage
    static constexpr int MASK_BYTES = HAS_MASK ? NSTAGE * BM * 4 : 0;        // one dword per row and stage
    static constexpr int LDS_BYTES = NSTAGE * STAGE * 2 + MASK_BYTES;
    static constexpr int THREADS = 320;
};

template <class L, int BM, int BN, bool OUT32, int NSTAGE>
__global__ __launch_bounds__(320) void igemm_s3_pc_kernel(S3Params p) {
    typedef PcCfg<BM, BN, NSTAGE, L::HAS_MASK> C;
    constexpr int BK = C::BK, TILE_A = C::TILE_A, TILE_B = C::TILE_B, STAGE = C::STAGE;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A_INST = BM / 16, B_INST = BN / 16;                        // 16 rows of 64 B per DMA instruction
    constexpr int M_INST = L::HAS_MASK ? BM / 64 : 0;                        // 64 rows of 4 B
    constexpr int PER_TILE = 3 * (A_INST + B_INST) + M_INST;
    static_assert(PER_TILE * (NSTAGE - 2) <= 63, "vmcnt is a 6-bit counter");
    static_assert(BM % 64 == 0 && BN % 64 == 0, "tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint16_t* smem = reinterpret_cast<uint16_t*>(lds_raw);
    uint32_t* msk = reinterpret_cast<uint32_t*>(lds_raw + (size_t)NSTAGE * STAGE * 2);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: uniform branches
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int n_iter = (p.Kp + BK - 1) / BK;

    if (wave == 4) {
        // ------------------------------------------------------------------ producer
        const int lrow = lane >> 2, lphys = lane & 3;
        typename L::Row rows[A_INST];
        int akp[A_INST];
#pragma unroll
        for (int i = 0; i < A_INST; i++) {
            const int r = i * 16 + lrow;
            rows[i] = L::make_row(p, m0 + r, n0);
            akp[i] = (lphys ^ ((r >> 2) & 3)) * 8;           // the XOR swizzle is applied on the source side
        }
        const uint16_t* wsrc[B_INST];
        bool wvalid[B_INST];
#pragma unroll
        for (int i = 0; i < B_INST; i++) {
            const int r = i * 16 + lrow;
            wvalid[i] = n0 + r < p.N;
            wsrc[i] = p.Wp + (size_t)(wvalid[i] ? n0 + r : 0) * p.Kp + (lphys ^ ((r >> 2) & 3)) * 8;
        }
        const uint8_t* msrc[M_INST > 0 ? M_INST : 1];
        if constexpr (L::HAS_MASK) {
#pragma unroll
            for (int i = 0; i < M_INST; i++) msrc[i] = p.mask + L::make_row(p, m0 + i * 64 + lane, n0).mrow;   // rows >= M: row 0 (their A is zero)
        }
        auto issue = [&](int it, int stage) {
            uint16_t* sbase = smem + stage * STAGE;
#pragma unroll
            for (int i = 0; i < A_INST; i++) {
                bool ok;
                const size_t off = L::offset(p, rows[i], it * BK + akp[i], ok);
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    const uint16_t* src = ok ? p.A + pl * p.a_plane + off : p.zeros;
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                                     (void __attribute__((address_space(3)))*)(sbase + pl * TILE_A + i * 16 * BK), 16, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < B_INST; i++) {
                const bool ok = wvalid[i] && (it * BK < p.Kp);
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    const uint16_t* src = ok ? wsrc[i] + it * BK + pl * p.w_plane : p.zeros;
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                                     (void __attribute__((address_space(3)))*)(sbase + 3 * TILE_A + pl * TILE_B + i * 16 * BK), 16, 0, 0);
                }
            }
            if constexpr (L::HAS_MASK) {
#pragma unroll
                for (int i = 0; i < M_INST; i++)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(msrc[i] + it * (BK / 8)),
                                                     (void __attribute__((address_space(3)))*)(msk + stage * BM + i * 64), 4, 0, 0);
            }
        };
        for (int t = 0; t < NSTAGE - 1 && t < n_iter; t++) issue(t, t);
        for (int it = 0; it < n_iter; it++) {
            // K-tile `it` has landed (tiles it+1 .. it+NSTAGE-2 may still be in flight)
            if (it + NSTAGE - 2 < n_iter) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE * (NSTAGE - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (it + NSTAGE - 1 < n_iter) issue(it + NSTAGE - 1, (it + NSTAGE - 1) % NSTAGE);
        }
        __builtin_amdgcn_s_barrier();                        // pairs with the consumers' barrier before the epilogue
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, fh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // Software-pipelined over half K-tiles: the barrier that publishes tile it+1 sits between the two k16 steps of tile
    // it, so the fragment reads (and mask expansion) of the next half tile run in the shadow of this half's 24 MFMAs
    // instead of leaving the matrix pipe idle after every barrier (prototype: 0.242 -> 0.199 ms).
    auto rd = [&](int stage, int step, bf16x8 (&af)[TM][3], bf16x8 (&bf)[TN][3]) {
        const uint16_t* As = smem + stage * STAGE;
        const uint16_t* Bs = As + 3 * TILE_A;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int r = wm * WM + i * 32 + frow;
            u32x4 av[3];
#pragma unroll
            for (int pl = 0; pl < 3; pl++) av[pl] = *reinterpret_cast<const u32x4*>(&As[pl * TILE_A + r * BK + s3_swz<4>(r, 2 * step + fh)]);
            if constexpr (L::HAS_MASK) {
                // keep byte of chunk 2*step+fh: bit e = element e of the chunk; dword j holds elements 2j (low), 2j+1 (high)
                const int x = (int)((msk[stage * BM + r] >> (8 * (2 * step + fh))) & 0xFFu);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe(x, 2 * j, 1), hi = (uint32_t)__builtin_amdgcn_sbfe(x, 2 * j + 1, 1);
                    const uint32_t d = __builtin_amdgcn_perm(hi, lo, 0x07060100u);
#pragma unroll
                    for (int pl = 0; pl < 3; pl++) av[pl][j] &= d;
                }
            }
#pragma unroll
            for (int pl = 0; pl < 3; pl++) af[i][pl] = __builtin_bit_cast(bf16x8, av[pl]);
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int r = wn * WN + j * 32 + frow;
#pragma unroll
            for (int pl = 0; pl < 3; pl++) bf[j][pl] = *reinterpret_cast<const bf16x8*>(&Bs[pl * TILE_B + r * BK + s3_swz<4>(r, 2 * step + fh)]);
        }
    };
    auto mm = [&](bf16x8 (&af)[TM][3], bf16x8 (&bf)[TN][3]) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                // smallest partial products first (as igemm_s3.h)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
            }
    };
    bf16x8 a0[TM][3], b0[TN][3], a1[TM][3], b1[TN][3];
    __builtin_amdgcn_s_barrier();                            // tile 0 has landed
    asm volatile("" ::: "memory");
    rd(0, 0, a0, b0);
    for (int it = 0; it < n_iter; it++) {
        rd(it % NSTAGE, 1, a1, b1);
        mm(a0, b0);
        if (it + 1 < n_iter) {
            // after this barrier the producer refills the stage of tile `it` itself: its step-1 fragments (issued 24 MFMAs
            // ago) must have arrived in registers first
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // tile it+1 has landed
            asm volatile("" ::: "memory");
            rd((it + 1) % NSTAGE, 0, a0, b0);
        }
        mm(a1, b1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // every fragment read is done: the ring becomes epilogue staging

    const int col = lane & 31, rbase = 4 * fh;
    if constexpr (OUT32) {
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + wn * WN + j * 32 + col;
            const float bv = n < p.N ? p.bias[n] : 0.0f;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                    if (m < p.M && n < p.N) {
                        const float v = acc[i][j][r] + bv;
                        p.out32[(size_t)m * p.N + n] = v > 0.0f ? v : v * 0.1f;
                    }
                }
        }
    } else {
        igemm_store_s3<TM, TN>(acc, smem + wave * (3 * 32 * 32), p.bias, p.out16, p.o_plane, p.M, p.N, m0 + wm * WM, n0 + wn * WN, lane);
    }
}

}  // namespace hnet
