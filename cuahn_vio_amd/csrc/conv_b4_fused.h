// conv_b4_fused.h — block_4_0 (7x7 s1, 2->8 @224x320) and block_4_1 (5x5 s2, 8->16 -> 112x160) in ONE kernel
// (reference model_to_trace.py:210-211 via conv() :7-15; SURVEY.md §7 "fuse conv0->conv1 through LDS").
//
// block_4_0's output is the largest activation of the network (573 440 values per pair; 880 MB per 256 pairs in the
// three-plane bf16 format) and block_4_1 re-reads every value 6.25 times: unfused, the two layers cost 0.34 + 0.58 ms
// per 256 pairs, both bound by that traffic.  Here a workgroup owns a TH1 x 32 tile of block_4_1 outputs:
//   phase 0  stage the input patch ((2 TH1 + 9) x 76 px x 2 ch, zero outside the image) in LDS, split into bf16 planes
//   phase 1  block_4_0 on the (2 TH1 + 3) x 67 region the tile needs: pixel-pair GEMM of conv_first.h (M = pairs of
//            adjacent pixels, N = (dx, cout) = 16, K = (kh, kw' 0..7, ci)) on v_mfma_f32_16x16x32_bf16 with split-bf16
//            operands (six MFMAs per step, two kernel rows per step); bias + LeakyReLU, zero outside the image
//            (= block_4_1's zero padding), split again and written to LDS as 16-byte pixel chunks
//            [plane][row][column parity][column/2][8 ch]
//   phase 2  block_4_1 straight from that LDS image: per MFMA step lane group g reads the chunk of tap 4*step+g
//            (consecutive output columns -> consecutive chunks), six MFMAs per step against weights held in VGPRs;
//            output in S3 planes.
// The 8-channel intermediate never touches HBM.  Workgroups are persistent (tiles strided over the grid) so the 132
// weight registers per lane are loaded once, not once per tile.
//
// Round-2 changes (profiles/r01_v7: SQ_VALU_MFMA_BUSY 0.38, LDS bank-conflict share 0.27):
//  * phase-2 operand reads were one ds_read_b128 per plane with the lane groups g and g+1 (two different taps) 2-way
//    bank-conflicted against each other whatever the tap offset (a ds_read_b128 serves lanes {0-3,12-15,20-27} together:
//    two 16-chunk windows that would have to start at the same bank).  Now each lane reads its 16-byte chunk as two
//    ds_read_b64, even groups low half first, odd groups high half first: the 32 lanes of a b64 access then cover all 64
//    banks exactly once for ANY chunk-aligned tap offsets.  Odd groups so get channels (4..7, 0..3); their weight
//    fragments are packed in that channel order (hnet_capi.hip), the MFMA sums over K and does not care.
//  * geometry is a template: <TH1 = 8, 512 threads> is the round-1 shape (85 KB of LDS, one workgroup per CU: staging,
//    phase 1 and phase 2 of the ONE resident workgroup never overlap); <TH1 = 7, 256 threads> needs 76.5 KB, so TWO
//    independent workgroups share a CU (one wave of each per SIMD) and one's staging / epilogue / stores run under the
//    other's MFMAs.
//  * NP = number of bf16 planes: 3 = split-bf16 (fp32-grade), 1 = plain bf16 operands (HNET_PREC_BF16, reported mode).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "igemm_s3.h"
#include "kernels.h"
#include "warp_dev.h"

#ifndef HNET_B4_ABLATE
#define HNET_B4_ABLATE 0          // tools/trace_b4.hip: 1 no phase 2, 2 no phase 1, 3 no global stores, 4 no phase-1 epilogue arithmetic, 5 no leftover M-tiles (wrong results)
                                  // (6 / 7 of round 6 - timing-only stand-ins for in-kernel sampling - were replaced by the real thing: the WARPIN instantiation below)
#endif

namespace hnet {

#ifdef HNET_B4_TRACE            // tools/trace_b4.hip: cycles per phase (s_memtime), summed in registers over the tiles of a workgroup and written once at the end
__device__ unsigned long long* g_b4_trace;     // (a store per stamp made the traced waves wait for their LDS-DMA at every stamp: vmcnt(0) in front of the store)
#define B4_T(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (tile_no > 1) tr_acc[k] += now_ - tr_prev; tr_prev = now_; } while (0)
#else
#define B4_T(k) do { } while (0)
#endif

template <int TH1_, int THREADS_, int NP_, bool DMA_ = false>
struct B4Cfg {
    static constexpr int TH1 = TH1_, TW1 = 32, THREADS = THREADS_, WAVES = THREADS_ / 64, NP = NP_;
    static constexpr int RH = 2 * TH1 + 3, RW = 2 * TW1 + 3;   // block_4_0 region: (2 TH1 + 3) x 67
    static constexpr bool DMA = DMA_;
    static constexpr int PH0 = RH + 6, PW0 = DMA_ ? 80 : 76;  // input patch: (RH + 6) x 76 px (67 + 7 taps + pad); 80 px = 20 chunks of 16 bytes per row for the LDS-DMA
    static constexpr int PROW0 = PW0 * 2;                    // bf16 elements per patch row (2 channels)
    static constexpr int PPLANE = PH0 * PROW0;               // elements per patch plane
    // chunks per (row, parity) of the S3 image: ceil(67 / 2) = 34; fp16-plane mode 40: an image row (2 XH chunks) is then 5 x 256 bytes, so that taps of one
    // kernel column sit on the same banks and two lane groups can share a conflict-free ds_read_b128 (phase 2, b41_tap() in kernels.h)
#ifdef HNET_B4_P2B128           // tools/trace_b4.hip: A/B of the two phase-2 read forms (timing only: the library packs the weights for the default)
    static constexpr bool P2B128 = NP_ == 2 && HNET_B4_P2B128;
#else
    static constexpr bool P2B128 = NP_ == 2;
#endif
    static constexpr int XH = P2B128 ? 40 : 34;
    static constexpr int PLANE = RH * 2 * XH * 8;            // bf16 elements per plane of the S3 image
    static constexpr int N_MT0 = 2 * RH + (2 * RH + 15) / 16; // block_4_0 M-tiles: RH rows x 2 + the pairs of columns 64..66
    static constexpr int N_MT1 = 2 * TH1;                    // block_4_1 M-tiles: TH1 rows x 2 halves of 16 px
    static constexpr int STAGE = WAVES * NP * 16 * 16;       // phase-2 output staging (overlays the dead patch)
    static constexpr int LDS_BYTES = (NP * PPLANE + NP * PLANE) * 2;
    static_assert(112 % TH1 == 0, "tiles cover the 112-row output exactly");
    static_assert(STAGE <= NP * PPLANE, "output staging fits in the patch area");
    static_assert(PW0 % 4 == 0, "patch rows are whole 16-byte chunks");
};

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// WARPIN (round 6): the kernel builds its input patches itself - cat(img1, warp(img2, H)) of model_to_trace.py:261-263 sampled straight into the LDS patch planes -
// instead of copying them from planes a prep launch wrote: the u8 source box of the NEXT tile's patch under the pair's homography and the patch of img1 are fetched
// by LDS-DMA (4 bytes per lane, out-of-image positions as out-of-range offsets = zeros) while this tile's phase 1 runs, and sampled between phase 1 and phase 2
// with the fast sampler of kernels.hip (same arithmetic, same clamps).  Box: B4W_ROWS rows of B4W_PITCH bytes; a patch whose box does not fit, or whose Z is not
// safely away from 0, takes the exact sampler with direct gathers (as the prep kernel's tiles do).
constexpr int B4W_PITCH = 128, B4W_ROWS = 64, B4W_I1ROWS = 26;
constexpr int B4W_LDS_BYTES = (B4W_ROWS + B4W_I1ROWS) * B4W_PITCH;

// the split-bf16 product group: six partial products, smallest first (igemm_s3.h); NP = 1: the plain bf16 product
template <int NP>
__device__ __forceinline__ f32x4_t b4_mfma(f32x4_t acc, const bf16x8 (&w)[3], const bf16x8 (&a)[3]) {
    return s3_mfma16<NP>(acc, w, a);         // igemm_s3.h: six bf16 / three fp16 / one bf16 product(s)
}

// (the round-1 / round-2 "v2" form of this kernel, kept until round 3 for A/B, was removed in round 4: it lost every measurement since r02_v1)

// ---------------------------------------------------------------------------------------------------------------------
// v3 (round 2): the same algorithm with the VECTOR-ISSUE cost taken out.
//
// rocprofv3 --pmc on v2 (profiles/r02_*): SQ_INSTS_VALU / SQ_INSTS_MFMA = 5.1, VALU busy 49 % of the SIMD cycles next to
// 39 % MFMA busy.  On CDNA4 the two share the SIMD's vector issue (a 16x16x32 MFMA holds it for 8 of its 16 cycles, a VALU
// instruction of one wave for 4: MI355X_MICROARCH.md, per-instruction cycle constants): 1476 MFMAs x 8 + 7470 VALU x 4 cycles
// per tile and SIMD is the measured kernel time to within 25 %.  The kernel was issue-bound on address arithmetic, register
// moves and epilogue math, not on the matrix pipe, the LDS or HBM (ablation of round 2, profiles/r02_v0_b4_ablation.log: every component additive).
//
// So: * every M-tile a wave will ever process is known at compile time (tile = wave + WAVES * j), the j loops are fully
//       unrolled and EVERY LDS address is one lane-invariant VGPR (set up once per workgroup) plus an instruction immediate
//       (ds_read_b64 / ds_write_b64 offset:N) - zero address VALU in the MFMA loops.  The reads are inline asm because hipcc
//       otherwise fuses pairs into ds_read2_b64 (8-bit offsets) and pays a v_add per plane and a v_mov per fragment half;
//     * waits are counted by hand (lgkmcnt, two read groups in flight) with the fragment registers as "+v" operands of the
//       wait statement so that no consumer can be scheduled above it (cdna_hip_programming.md §5.7 form (ii), rule 18);
//     * bias = initial accumulator; LeakyReLU = max(v, 0.1 v); the three planes of a value pair come out of
//       v_cvt_pk_bf16_f32 already packed (no shifts / ors to assemble the 8-byte pieces);
//     * wave-uniform quantities (wave id, tile origin, output row base) live in SGPRs (readfirstlane), global stores use the
//       SGPR-base + VGPR-offset form.
namespace b4v3 {

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int IMM>
__device__ __forceinline__ bf16x4 rd64(uint32_t addr) {
    static_assert(IMM >= 0 && IMM < 65536, "ds offset is 16 bits");
    bf16x4 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
template <int IMM>
__device__ __forceinline__ void wr64(uint32_t addr, uint2 v) {
    static_assert(IMM >= 0 && IMM < 65536, "ds offset is 16 bits");
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(IMM) : "memory");
}
template <int IMM>
__device__ __forceinline__ bf16x8 rd128i(uint32_t addr) {
    static_assert(IMM >= 0 && IMM < 65536, "ds offset is 16 bits");
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
// wait until at most N LDS operations are outstanding; the NP fragment registers named cannot be consumed above the statement
template <int N>
__device__ __forceinline__ void waitQ(bf16x8 (&a)[3]) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(N));
}
__device__ __forceinline__ u32x4 rd128(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// wait until at most N LDS operations of this wave are outstanding; the six (a) / twelve (a, b) fragment halves named here
// cannot be consumed above this statement
template <int N>
__device__ __forceinline__ void wait6(bf16x4 (&a)[6]) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait3(bf16x4 (&a)[3]) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : "n"(N));
}
// the same with only the 2 NP (fragment) / NP (tail) registers a mode loads: naming registers that were never loaded would keep them alive
template <int N, int NP>
__device__ __forceinline__ void waitF(bf16x4 (&a)[6]) {
    if constexpr (NP == 3) wait6<N>(a);
    else if constexpr (NP == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N));
    else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(N));
}
template <int N, int NP>
__device__ __forceinline__ void waitT(bf16x4 (&a)[3]) {
    if constexpr (NP == 3) wait3<N>(a);
    else if constexpr (NP == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(N));
    else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[0]) : "n"(N));
}
using s3p::cvt_pk;
using s3p::split_pair;
using s3p::lrelu;

}  // namespace b4v3

// DMA = true: x_in_v points to the padded bf16-plane input written by the prep kernel (kernels.h B4_*: [plane][B][B4_HP][B4_WP]
// dwords, x_plane dwords per plane); the patch of a tile is copied global -> LDS by global_load_lds_dwordx4 (no registers, no
// split, no bounds logic: the zero border is in memory), the copy of tile t+1 runs under phase 2 of tile t, and phase 2 stores
// its 8-byte pieces straight to global memory (the patch area is busy, and 64 lanes x 8 bytes already cover whole 512-byte
// runs).  DMA = false: fp32 NHWC input, split while staged (the round-2 first version of this kernel).
// REUSE (4-wave geometry): a wave's consecutive phase-1 M-tiles are two region rows apart, so step st of tile j + 1 reads exactly what
// step st + 1 of tile j read (same lanes, same addresses): the fragments stay in registers and a tile fetches ONE new 32-deep fragment plus
// its 16-deep tail instead of three plus tail (LDS reads of phase 1: 7 -> 3 ds_read_b64 per plane and tile), prefetched one tile ahead.
template <int TH1, int THREADS, int NP, bool DMA = false, bool REUSE = false, bool WARPIN = false>
__global__ __launch_bounds__(THREADS, 2) void block4_fused_kernel(const void* __restrict__ x_in_v, size_t x_plane, const u32x4* __restrict__ w0frag,
                                                                  const float* __restrict__ bias0, const u32x4* __restrict__ w1frag,
                                                                  const float* __restrict__ bias1, uint16_t* __restrict__ out16,
                                                                  size_t o_plane, int n_tiles, int flags, B4Warp wp = B4Warp{}) {
    static_assert(!WARPIN || (DMA && NP == 2), "the sampling form replaces the LDS-DMA patch copy of the fp16-plane mode");
    using namespace b4v3;
    typedef B4Cfg<TH1, THREADS, NP, DMA> C;
    const float* const x_in = reinterpret_cast<const float*>(x_in_v);
    constexpr int WAVES = C::WAVES, TW1 = C::TW1, RH = C::RH, PH0 = C::PH0, PW0 = C::PW0, PROW0 = C::PROW0;
    constexpr int PPLANE = C::PPLANE, XH = C::XH, PLANE = C::PLANE, N_MT0 = C::N_MT0, N_MT1 = C::N_MT1;
    constexpr int H0 = 224, W0 = 320, H1 = 112, W1 = 160;
    constexpr int HW = WAVES / 2;                                     // region rows / output rows a wave advances per j
    constexpr int N_REG = 2 * RH;                                     // regular phase-1 M-tiles (row, half)
    constexpr int J1 = (N_REG + WAVES - 1) / WAVES, J2 = (N_MT1 + WAVES - 1) / WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint16_t* patch = reinterpret_cast<uint16_t*>(lds_raw);                    // [NP][PH0][PROW0] bf16
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw;
    const uint32_t img0 = lds0 + NP * PPLANE * 2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // SGPR: everything derived from it is wave-uniform
    const int m = lane & 15, g = lane >> 4;
    const int wrow = wave >> 1, whalf = wave & 1;

    // ---- weights -> registers, once per (persistent) workgroup
    constexpr int NW = s3_wplanes<NP>;                                 // weight planes of the mode (igemm_s3.h)
    bf16x8 w0[4][3], w1[7][3];
#pragma unroll
    for (int st = 0; st < 4; st++)
#pragma unroll
        for (int pl = 0; pl < NW; pl++) w0[st][pl] = __builtin_bit_cast(bf16x8, w0frag[(st * 3 + pl) * 64 + lane]);
#pragma unroll
    for (int st = 0; st < 7; st++)
#pragma unroll
        for (int pl = 0; pl < NW; pl++) w1[st][pl] = __builtin_bit_cast(bf16x8, w1frag[(st * 3 + pl) * 64 + lane]);
    // kernel row 6 alone, as v_mfma_f32_16x16x16_bf16 fragments (K = 8 taps x 2 channels of ONE row): the fourth 32-deep step pairs row 6
    // with a row of zero weights - half of its six MFMAs' work and of its LDS reads (12.5 % of phase 1) multiplied zeros
    bf16x4 w0t[3];
#pragma unroll
    for (int pl = 0; pl < NW; pl++) {
        const u32x4 t = w0frag[(4 * 3 + pl) * 64 + lane];
        w0t[pl] = __builtin_bit_cast(bf16x4, uint2{t[0], t[1]});
    }
    const int dx = g >> 1, co0 = 4 * (g & 1);            // phase 1: D row 4g + r = (dx, co0 + r)
    f32x4_t bv, bv1;                                      // bias = initial accumulator, at the accumulator's scale
#pragma unroll
    for (int r = 0; r < 4; r++) { bv[r] = bias0[co0 + r] * s3_acc_scale<NP>; bv1[r] = bias1[4 * g + r] * s3_acc_scale<NP>; }

    // ---- lane-invariant LDS byte addresses (the per-tile part is an instruction immediate)
    // phase 1, regular M-tile j of this wave: region row wrow + HW*j, half whalf, pixel pair 16*whalf + m;
    // step st reads kernel row min(2st + (g>>1), 6) (row 7 has zero weights), taps 4(g&1)..+3
    uint32_t p1a[4];
#pragma unroll
    for (int st = 0; st < 4; st++)
        p1a[st] = lds0 + 2 * ((wrow + min(2 * st + (g >> 1), 6)) * PROW0 + (16 * whalf + m) * 4 + 8 * (g & 1));
    // the 16-deep tail step: kernel row 6, lane group g reads taps 2g, 2g + 1 (8 bytes)
    const uint32_t p1t = lds0 + 2 * ((wrow + 6) * PROW0 + (16 * whalf + m) * 4 + 4 * g);
    // phase-1 store: pixel column 2m + dx of the half, channels co0..co0+3, region row wrow (+ HW*j as immediate)
    const uint32_t st1a = img0 + 2 * (((wrow * 2 + dx) * XH + 16 * whalf + m) * 8 + co0);
    // phase 1, the leftover M-tile of waves 3, 2, 1 (k = 0, 1, 2): slot 16 k + m = region row (16 k + m) / 2 (clamped to the last), pixel pair 32 + (m & 1)
    static_assert(N_MT0 - N_REG <= WAVES - 1, "one leftover M-tile per wave");
    const bool has_left = wave >= 1 && N_REG + (WAVES - 1 - wave) < N_MT0;
    const int rowL = min((16 * (WAVES - 1 - wave) + m) >> 1, RH - 1), pairL = 32 + (m & 1), colL = 2 * pairL + dx;
    uint32_t pL[3];
#pragma unroll
    for (int st = 0; st < 3; st++) pL[st] = lds0 + 2 * ((rowL + 2 * st + (g >> 1)) * PROW0 + pairL * 4 + 8 * (g & 1));
    const uint32_t pLt = lds0 + 2 * ((rowL + 6) * PROW0 + pairL * 4 + 4 * g);
    const uint32_t stL = img0 + 2 * (((rowL * 2 + dx) * XH + pairL) * 8 + co0);
    // phase 2, M-tile j: output row wrow + HW*j, half whalf, column 16*whalf + m; tap t = 4*step + g; even groups read the
    // low 8 bytes of the chunk first, odd groups the high 8 bytes (bank-conflict free, see the header); p2b = the other half
    uint32_t p2a[7], p2b[7];
#pragma unroll
    for (int st = 0; st < 7; st++) {
        int t = 4 * st + g;
        if constexpr (C::P2B128) {                                    // the tap table of kernels.h; a group without a tap reads its neighbour's pixels (its weights are zeros)
            const int tg = g == 0 ? b41_tap(st, 0) : g == 1 ? b41_tap(st, 1) : g == 2 ? b41_tap(st, 2) : b41_tap(st, 3);
            const int tn = g == 1 ? b41_tap(st, 0) : b41_tap(st, 2);
            t = tg >= 0 ? tg : tn;
        }
        const int kh = t / 5, kw = t - kh * 5;
        const int tap = t < 25 ? ((kh * 2 + (kw & 1)) * XH + (kw >> 1)) * 8 : 0;
        p2a[st] = img0 + 2 * (((2 * wrow) * 2 * XH + 16 * whalf + m) * 8 + tap + (C::P2B128 ? 0 : 4 * (g & 1)));
        p2b[st] = p2a[st] ^ 8u;
    }
    // phase-2 output staging (wave private, overlays the dead patch): write 4 channels of pixel m, read back 16-byte pieces
    const uint32_t st2w = lds0 + 2 * (wave * (NP * 16 * 16) + m * 16 + 4 * g);
    const int pc0 = lane, pc1 = 64 + lane;                            // pieces [plane][16 px][2 halves of 8 channels]
    const uint32_t st2r0 = lds0 + 2 * (wave * (NP * 16 * 16) + ((pc0 >> 5) * 16 + ((pc0 & 31) >> 1)) * 16 + (pc0 & 1) * 8);
    const uint32_t st2r1 = lds0 + 2 * (wave * (NP * 16 * 16) + ((pc1 >> 5) * 16 + ((pc1 & 31) >> 1)) * 16 + (pc1 & 1) * 8);
    const uint32_t gvo0 = (uint32_t)(((size_t)(pc0 >> 5) * o_plane + ((pc0 & 31) >> 1) * 16 + (pc0 & 1) * 8) * 2);   // byte offsets
    const uint32_t gvo1 = (uint32_t)(((size_t)(pc1 >> 5) * o_plane + ((pc1 & 31) >> 1) * 16 + (pc1 & 1) * 8) * 2);

    const bool reverse = (flags & 1) != 0;
    const bool o_pad = (flags & 64) != 0;
    const int o_wp = o_pad ? B42_WP : W1;                             // output row pitch in pixels, pixels per pair, pixel (0, 0)
    const size_t o_img = o_pad ? B42_IMG : (size_t)H1 * W1;
    const int o_org = o_pad ? B42_PADY * B42_WP + B42_PADX : 0;
    // XCD-aware tile order (r02_v9).  Slot t = blockIdx.x + k gridDim.x is processed by workgroup blockIdx.x, and consecutive workgroup ids
    // go to consecutive XCDs, each with a private L2: with tile = slot, the 23 x 80-pixel patches of neighbouring tiles - which overlap by
    // 2.05x in total - were fetched by eight different L2s (FETCH_SIZE, corrected x2: 518 MB per launch for 242 MB of input).  Here XCD x
    // owns the contiguous tile range [x T/8, (x + 1) T/8) and walks it in step with its 64 (x 2) workgroups, so the tiles resident on one
    // XCD at any time are neighbours in the image.  flags bit 4 switches back (A/B).
    const bool xcd_order = (flags & 16) == 0 && (n_tiles & 7) == 0 && (gridDim.x & 7) == 0;
    const int t8 = n_tiles >> 3;
    auto tile_origin = [&](int t, int& b, int& by, int& bx) {
        if (xcd_order) t = (t & 7) * t8 + (t >> 3);
        int bid = reverse ? n_tiles - 1 - t : t;
        bx = bid % (W1 / TW1); bid /= (W1 / TW1);
        by = bid % (H1 / TH1);
        b = bid / (H1 / TH1);
    };
    // !DMA: patch pixels of the NEXT tile are prefetched into registers while the current tile computes
    constexpr int PPT = DMA ? 1 : (PH0 * PW0 + THREADS - 1) / THREADS;     // patch pixels per thread
    float2 pre[PPT];
    uint32_t pre_ok = 0;            // validity bits; applied when the registers are consumed, so the loads stay in flight
    auto patch_load = [&](int t) {
        pre_ok = 0;
        int b, by, bx;
        tile_origin(t, b, by, bx);
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
    // DMA: the patch [plane][PH0 rows][PW0 / 4 chunks of 16 bytes] is one linear run of NCH chunks in LDS; wave-instruction k
    // (64 chunks) is issued by wave k % WAVES, lane l fetching chunk 64 k + l.  Its source offset from the tile's first chunk is
    // lane invariant: set up once (IPW registers).  Patch pixel (0, 0) = image pixel (Ry0 - 3, Rx0 - 3) = padded-array dword
    // (2 by TH1, 64 bx): a multiple of 4 dwords, so every chunk is a 16-byte aligned load.
    constexpr int CPR = PW0 / 4, CPP = PH0 * CPR, NCH = NP * CPP, NINS = (NCH + 63) / 64, IPW = (NINS + WAVES - 1) / WAVES;
    uint32_t dsrc[IPW];
    uint32_t dma_ok = 0;
    if constexpr (DMA) {
#pragma unroll
        for (int i = 0; i < IPW; i++) {
            const int q = 64 * (wave + WAVES * i) + lane;
            const int qq = q < NCH ? q : 0;
            const int pl = qq / CPP, r = qq - pl * CPP, row = r / CPR, c = r - row * CPR;
            dsrc[i] = (uint32_t)((size_t)pl * x_plane * 4 + ((size_t)row * B4_WP + 4 * c) * 4);
            dma_ok |= q < NCH ? (1u << i) : 0u;
        }
    }
    auto dma_issue = [&](int t) {
        int b, by, bx;
        tile_origin(t, b, by, bx);
        const unsigned char* base = reinterpret_cast<const unsigned char*>(x_in_v) + (((size_t)b * B4_HP + 2 * by * TH1) * B4_WP + 64 * bx) * 4;
#pragma unroll
        for (int i = 0; i < IPW; i++) {
            const int k = wave + WAVES * i;                  // wave-uniform
            if (k < NINS && ((dma_ok >> i) & 1u))
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + dsrc[i]),
                                                 (void __attribute__((address_space(3)))*)(lds_raw + k * 1024), 16, 0, 0);
        }
    };
    // ---- WARPIN: box geometry + LDS-DMA issue for the patch of tile t (everything wave-uniform), and the sampling of that patch
    unsigned char* const wbox = lds_raw + C::LDS_BYTES;                       // [B4W_ROWS][B4W_PITCH] u8: img2's box
    unsigned char* const wi1 = wbox + B4W_ROWS * B4W_PITCH;                    // [B4W_I1ROWS][B4W_PITCH] u8: img1, patch row r, byte pc + 3
    [[maybe_unused]] float wh[9];
    [[maybe_unused]] int w_gx0 = 0, w_ry0 = 0, w_pitch = 0, w_rows = 0, w_b = 0, w_iy0 = 0, w_ix0 = 0;
    [[maybe_unused]] bool w_fast = false;
    [[maybe_unused]] auto warp_issue = [&](int t) {
        int nb, nby, nbx;
        tile_origin(t, nb, nby, nbx);
        w_b = nb; w_iy0 = 2 * nby * TH1 - 5; w_ix0 = 64 * nbx - 5;           // image pixel of patch pixel (0, 0)
#pragma unroll
        for (int i = 0; i < 9; i++) wh[i] = wp.H[nb * 9 + i];
        // bounding box of the taps: the four warped corners of the patch clipped to the image (a projective map with Z of one sign takes the rectangle to a convex
        // quadrilateral), exactly as warp_stage_box of kernels.hip: lane c of the first quad evaluates corner c
        const int y0c = max(w_iy0, 0), y1c = min(w_iy0 + PH0 - 1, H0 - 1), x0c = max(w_ix0, 0), x1c = min(w_ix0 + PW0 - 1, W0 - 1);
        const int c = lane & 3;
        float cx, cy, Z;
        warp_coords(wh, (c & 1) ? x1c : x0c, (c >> 1) ? y1c : y0c, cx, cy, Z);
        const unsigned quad0 = 0xFu;
        const bool pos = ((unsigned)__ballot(Z > 0.0f) & quad0) == quad0;
        const bool neg = ((unsigned)__ballot(Z < 0.0f) & quad0) == quad0;
        const bool zs = ((unsigned)__ballot(warp_z_safe(Z)) & quad0) == quad0;
        const bool finite = ((unsigned)__ballot(fabsf(cx) < 1.0e6f && fabsf(cy) < 1.0e6f) & quad0) == quad0;
        const float lo_x = uniform_f(quad_min(cx)), hi_x = uniform_f(quad_max(cx));
        const float lo_y = uniform_f(quad_min(cy)), hi_y = uniform_f(quad_max(cy));
        const bool ok = (pos || neg) && finite;
        const int rx0 = max((int)floorf(ok ? lo_x : 0.0f) - 1, -1), rx1 = min((int)floorf(ok ? hi_x : 0.0f) + 2, W0 + 1);
        const int ry0 = max((int)floorf(ok ? lo_y : 0.0f) - 1, -1), ry1 = min((int)floorf(ok ? hi_y : 0.0f) + 2, H0 + 1);
        w_gx0 = (rx0 + 4) / 4 * 4 - 4;
        w_ry0 = ry0;
        w_pitch = rx1 >= rx0 ? (rx1 - w_gx0 + 4) / 4 * 4 : 0;
        w_rows = ry1 >= ry0 ? ry1 - ry0 + 1 : 0;
        w_fast = ok && zs && w_pitch <= B4W_PITCH && w_rows <= B4W_ROWS && w_rows >= 2 && w_pitch >= 2;
        typedef void __attribute__((address_space(3)))* lds_p;
        const int hrow = lane >> 5, cc = lane & 31;
        if (w_fast) {
            const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(wp.img2 + (size_t)nb * (H0 * W0)), 0, H0 * W0, 0x00020000);
#pragma unroll
            for (int k = 0; k < B4W_ROWS / 2 / WAVES; k++) {
                const int rp = wave + WAVES * k;                             // pair of box rows (wave-uniform)
                if (2 * rp < w_rows) {
                    const int r = 2 * rp + hrow, y = w_ry0 + r, x = w_gx0 + 4 * cc;     // x is a multiple of 4: the four pixels are all in or all out
                    const bool in = r < w_rows && 4 * cc < w_pitch && (unsigned)y < (unsigned)H0 && (unsigned)x < (unsigned)W0;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_p)(wbox + rp * 256), 4, in ? (uint32_t)(y * W0 + x) : S3_OOB, 0, 0, 0);
                }
            }
        }
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(wp.img1 + (size_t)nb * (H0 * W0)), 0, H0 * W0, 0x00020000);
#pragma unroll
        for (int k = 0; k < (B4W_I1ROWS / 2 + WAVES - 1) / WAVES; k++) {
            const int rp = wave + WAVES * k;
            if (rp < B4W_I1ROWS / 2) {
                const int r = 2 * rp + hrow, y = w_iy0 + r, x = w_ix0 - 3 + 4 * cc;     // 64 nbx - 8: a multiple of 4
                const bool in = r < PH0 && cc < (PW0 + 3 + 3) / 4 && (unsigned)y < (unsigned)H0 && (unsigned)x < (unsigned)W0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_p)(wi1 + rp * 256), 4, in ? (uint32_t)(y * W0 + x) : S3_OOB, 0, 0, 0);
            }
        }
    };
    [[maybe_unused]] auto warp_sample = [&]() {
        // = the fast branch of prep_warp_tiled_kernel (kernels.hip) pixel for pixel: X, Y, Z by the same FMAs, one shared reciprocal + Newton step, the position clamped
        // to the box (which acts only where the box was clipped to the frame: the zero padding), three interpolations; the pair (img1, warped img2) split into the two
        // fp16 planes as the prep kernel's stores did.  Patch pixels outside the image are zeros (block_4_0's padding).
        const float bx_lo = (float)max(w_gx0, -1), bx_hi = (float)min(w_gx0 + w_pitch - 2, W0);
        const float by_lo = (float)max(w_ry0, -1), by_hi = (float)min(w_ry0 + w_rows - 2, H0);
        const int ibase = -(w_ry0 * B4W_PITCH + w_gx0);
        const uint8_t* i2b = wp.img2 + (size_t)w_b * (H0 * W0);
#pragma unroll 2
        for (int q = 0; q < (PH0 * PW0 + THREADS - 1) / THREADS; q++) {
            const int i = tid + q * THREADS;
            if (i < PH0 * PW0) {
                const int pr = i / PW0, pc = i - pr * PW0;
                const int iy = w_iy0 + pr, ix = w_ix0 + pc;
                const bool inside = (unsigned)iy < (unsigned)H0 && (unsigned)ix < (unsigned)W0;
                const float a = u8_to_unit((float)wi1[pr * B4W_PITCH + pc + 3]);
                float w;
                if (w_fast) {                                                // workgroup-uniform
                    const float fu = (float)ix, fv = (float)iy;
                    const float X = fmaf(wh[1], fv, fmaf(wh[0], fu, wh[2])), Y = fmaf(wh[4], fv, fmaf(wh[3], fu, wh[5])), Z = fmaf(wh[7], fv, fmaf(wh[6], fu, wh[8]));
                    const float r0 = __builtin_amdgcn_rcpf(Z);
                    const float r1 = fmaf(fmaf(-Z, r0, 1.0f), r0, r0);
                    const float sx = __builtin_amdgcn_fmed3f(X * r1, bx_lo, bx_hi), sy = __builtin_amdgcn_fmed3f(Y * r1, by_lo, by_hi);
                    const float x0f = floorf(sx), y0f = floorf(sy);
                    const float wx1 = sx - x0f, wy1 = sy - y0f;
                    const int e = (int)fmaf(y0f, (float)B4W_PITCH, x0f) + ibase;
                    const float t0 = u8_to_unit((float)wbox[e]), t1 = u8_to_unit((float)wbox[e + 1]);
                    const float t2 = u8_to_unit((float)wbox[e + B4W_PITCH]), t3 = u8_to_unit((float)wbox[e + B4W_PITCH + 1]);
                    const float top = fmaf(wx1, t1 - t0, t0), bot = fmaf(wx1, t3 - t2, t2);
                    w = fmaf(wy1, bot - top, top);
                } else {                                                     // rare: the exact sampler, taps gathered from memory
                    float sx, sy, Z;
                    warp_coords<false>(wh, inside ? ix : 0, inside ? iy : 0, sx, sy, Z);
                    w = warp_taps_global<uint8_t, false>(i2b, sx, sy, nullptr);
                }
                uint32_t pk[3];
                split_pair<NP>(inside ? a : 0.0f, inside ? w : 0.0f, pk);
                const int pe = pr * PROW0 + pc * 2;
#pragma unroll
                for (int pl = 0; pl < NP; pl++) *reinterpret_cast<uint32_t*>(&patch[pl * PPLANE + pe]) = pk[pl];
            }
        }
    };
    if ((int)blockIdx.x < n_tiles) {
        if constexpr (WARPIN) {
            warp_issue(blockIdx.x);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            warp_sample();                                   // (made visible by the barrier at the top of the tile loop)
        } else if constexpr (DMA) dma_issue(blockIdx.x);
        else patch_load(blockIdx.x);
    }
    const uint32_t gv = (uint32_t)(m * 32 + g * 8);          // DMA: byte offset of this lane's 8-byte piece in a 16-pixel output run

    [[maybe_unused]] int tile_no = -1;
    [[maybe_unused]] unsigned long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_prev = 0;      // (6: WARPIN sampling, 7: WARPIN box geometry + DMA issue)
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_no++;
        B4_T(0);
        int b, by, bx;
        tile_origin(tile, b, by, bx);
        const int ty0 = by * TH1, tx0 = bx * TW1;
        const int Ry0 = 2 * ty0 - 2, Rx0 = 2 * tx0 - 2;      // image coordinates of region pixel (0,0)

        if constexpr (DMA) {
            // ---- phase 0: this wave's share of the patch copy has landed (it ran under phase 2 of the previous tile) ...
            if constexpr (WARPIN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the patch was written by LDS stores (the stores of phase 2 stay in flight)
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // ... and everybody else's; the previous tile's phase 2 is done with the image
            asm volatile("" ::: "memory");
            if constexpr (WARPIN) {
                B4_T(1);
                if (tile + (int)gridDim.x < n_tiles) warp_issue(tile + gridDim.x);   // box + img1 of the NEXT tile's patch: in flight during phase 1
                B4_T(7);
            } else
            B4_T(1);
        } else {
            // ---- phase 0: prefetched patch -> bf16 planes in LDS
            __syncthreads();                                     // previous tile's phase 2 is done with the LDS
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * THREADS;
                if (i < PH0 * PW0) {
                    const int pr = i / PW0, pc = i - pr * PW0;
                    const bool ok = (pre_ok >> q) & 1u;
                    uint32_t pk[3];
                    split_pair<NP>(ok ? pre[q].x : 0.f, ok ? pre[q].y : 0.f, pk);
                    const int e = pr * PROW0 + pc * 2;
#pragma unroll
                    for (int pl = 0; pl < NP; pl++) *reinterpret_cast<uint32_t*>(&patch[pl * PPLANE + e]) = pk[pl];
                }
            }
            __syncthreads();
            if (tile + (int)gridDim.x < n_tiles) patch_load(tile + gridDim.x);   // in flight during phases 1 and 2
        }

        // ---- phase 1: block_4_0 over the region, into the S3 image.  Regular M-tiles: fully unrolled, immediate addressing.
        // column validity of this lane's pixel (row validity is wave-uniform per M-tile)
#if HNET_B4_ABLATE != 2
        const bool col_ok = (unsigned)(Rx0 + whalf * 32 + 2 * m + dx) < (unsigned)W0;
        if constexpr (REUSE && HW == 2) {
            constexpr int R = 2 * NP;                        // reads per 32-deep fragment; a tail fragment is NP reads
            bf16x4 F[J1 + 2][6];                             // fragment q = region rows wrow + 2q, 2q + 1 (tile j uses q = j, j + 1, j + 2)
            bf16x4 T[J1][3];                                 // tail fragment of tile j = region row wrow + 2j + 6
            auto rdF = [&](auto qc) {
                constexpr int q = decltype(qc)::value;
                static_for<NP>([&](auto pc) {
                    constexpr int pl = decltype(pc)::value;
                    F[q][2 * pl] = rd64<2 * q * PROW0 * 2 + pl * PPLANE * 2>(p1a[0]);
                    F[q][2 * pl + 1] = rd64<2 * q * PROW0 * 2 + pl * PPLANE * 2 + 8>(p1a[0]);
                });
            };
            auto rdT = [&](auto jc2) {
                constexpr int j2 = decltype(jc2)::value;
                static_for<NP>([&](auto pc) {
                    constexpr int pl = decltype(pc)::value;
                    T[j2][pl] = rd64<HW * j2 * PROW0 * 2 + pl * PPLANE * 2>(p1t);
                });
            };
            // in flight on entry of tile j > 0: F[j + 2] (R reads) then T[j] (NP reads), issued by tile j - 1.  lgkmcnt counts to 15: never more
            // than 2 R + NP = 15 (NP = 3) operations are outstanding when a wait is issued (this tile's LDS stores included, which only makes
            // a wait conservative: LDS operations complete in order)
            rdF(std::integral_constant<int, 0>{}); rdF(std::integral_constant<int, 1>{});
            // Round 4: the epilogue of tile j - 1 (bias is in the accumulator: 2 adds, LeakyReLU + plane split of four values, NP stores: ~26 vector instructions
            // that used to run with nothing beside them) is issued between the first six MFMAs of tile j - two accumulator sets, selected by j & 1.
            f32x4_t accs[2], accts[2];
            auto epi = [&](auto jc) {                        // the epilogue proper of tile j: accs / accts [j & 1] -> the S3 image
                constexpr int j = decltype(jc)::value;
                f32x4_t acc = accs[j & 1] + accts[j & 1];
                const bool ok = col_ok && (unsigned)(Ry0 + wrow + HW * j) < (unsigned)H0;
                uint32_t pa[3], pb[3];
#if HNET_B4_ABLATE == 4
                pa[0] = __builtin_bit_cast(uint32_t, acc[0]); pa[1] = __builtin_bit_cast(uint32_t, acc[1]); pb[0] = __builtin_bit_cast(uint32_t, acc[2]); pb[1] = __builtin_bit_cast(uint32_t, acc[3]) + ok;
#else
                s3p::act_split<NP>(acc[0], acc[1], pa, ok);
                s3p::act_split<NP>(acc[2], acc[3], pb, ok);
#endif
                constexpr int JW = HW * j * 2 * XH * 16;
                static_for<NP>([&](auto pc) {
                    constexpr int pl = decltype(pc)::value;
                    wr64<JW + pl * PLANE * 2>(st1a, make_uint2(pa[pl], pb[pl]));
                });
            };
            static_for<J1>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr bool all_waves = WAVES * (j + 1) <= N_REG;     // tile j exists for every wave
                if (all_waves || wave + WAVES * j < N_REG) {             // wave-uniform
                    // the next tile exists for every wave (compile time) / for some waves only (then nothing is prefetched and the waits
                    // below are the conservative ones: waiting for fewer outstanding operations than there are is always correct)
                    constexpr bool next_all = WAVES * (j + 2) <= N_REG;
                    accs[j & 1] = bv;
                    auto mm = [&](bf16x4 (&fr)[6], const bf16x8 (&w)[3]) {
                        bf16x8 a[3];
#pragma unroll
                        for (int pl = 0; pl < NP; pl++) a[pl] = __builtin_shufflevector(fr[2 * pl], fr[2 * pl + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                        accs[j & 1] = b4_mfma<NP>(accs[j & 1], w, a);
                    };
                    constexpr bool pre = j > 0 && WAVES * (j + 1) <= N_REG;      // F[j + 2], T[j] were prefetched by tile j - 1
                    if constexpr (j == 0) {
                        waitF<R, NP>(F[0]); __builtin_amdgcn_sched_barrier(0); mm(F[0], w0[0]);
                        rdF(std::integral_constant<int, 2>{});
                        rdT(std::integral_constant<int, 0>{});
                        waitF<R + NP, NP>(F[1]); __builtin_amdgcn_sched_barrier(0); mm(F[1], w0[1]);
                        if constexpr (next_all) rdF(std::integral_constant<int, 3>{});
                    } else {
                        if constexpr (!pre) { rdF(std::integral_constant<int, j + 2>{}); rdT(std::integral_constant<int, j>{}); }
                        if constexpr (next_all) rdF(std::integral_constant<int, j + 3>{});
                        __builtin_amdgcn_sched_barrier(0);   // one tile at a time: without it the scheduler interleaves tiles and runs out of registers
                        mm(F[j], w0[0]);
                        mm(F[j + 1], w0[1]);
                        epi(std::integral_constant<int, j - 1>{});
                        // the previous tile's epilogue in the shadows of these MFMAs (an MFMA holds the vector issue for 8 of its 16 cycles), its stores behind them.
                        // (Its LDS stores are issued behind this tile's reads: the counted waits below only get more conservative by them.)
                        constexpr int NM = 2 * (NP == 3 ? 6 : NP == 2 ? 3 : 1);
#pragma unroll
                        for (int q = 0; q < NM; q++) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, (26 + NM - 1) / NM, 0);
                        }
                    }
                    waitF<(next_all ? R : 0) + NP, NP>(F[j + 2]); __builtin_amdgcn_sched_barrier(0); mm(F[j + 2], w0[2]);
                    if constexpr (next_all) rdT(std::integral_constant<int, j + 1>{});
                    waitT<(next_all ? R + NP : 0), NP>(T[j]); __builtin_amdgcn_sched_barrier(0);
                    f32x4_t acct = {0.f, 0.f, 0.f, 0.f};     // own accumulator chain: see the non-REUSE path
                    if constexpr (NP == 2) {
                        acct = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, w0t[2]), __builtin_bit_cast(f16x4, T[j][1]), acct, 0, 0, 0);
                        acct = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, w0t[1]), __builtin_bit_cast(f16x4, T[j][0]), acct, 0, 0, 0);
                        acct = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, w0t[0]), __builtin_bit_cast(f16x4, T[j][0]), acct, 0, 0, 0);
                    } else {
                        if constexpr (NP == 3) {
                            acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[0], T[j][2], acct, 0, 0, 0);
                            acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[2], T[j][0], acct, 0, 0, 0);
                            acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[1], T[j][1], acct, 0, 0, 0);
                            acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[0], T[j][1], acct, 0, 0, 0);
                            acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[1], T[j][0], acct, 0, 0, 0);
                        }
                        acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[0], T[j][0], acct, 0, 0, 0);
                    }
                    accts[j & 1] = acct;
                    if constexpr (j == J1 - 1) epi(std::integral_constant<int, j>{});           // the last tile: nothing to hide behind
                } else {
                    if constexpr (j > 0) epi(std::integral_constant<int, j - 1>{});             // this wave has no tile j: finish its last one
                }
            });
        } else
        static_for<J1>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (wave + WAVES * j < N_REG) {                  // wave-uniform
                constexpr int JR = HW * j * PROW0 * 2;       // bytes: HW region rows down
                bf16x4 f[4][6];                              // [step][plane, half]; lgkmcnt is a 4-bit counter: at most two steps
                auto rd = [&](auto sc) {                     // (12 reads) stay in flight behind the one being waited for
                    constexpr int st = decltype(sc)::value;
                    static_for<NP>([&](auto pc) {
                        constexpr int pl = decltype(pc)::value;
                        f[st][2 * pl] = rd64<JR + pl * PPLANE * 2>(p1a[st]);
                        f[st][2 * pl + 1] = rd64<JR + pl * PPLANE * 2 + 8>(p1a[st]);
                    });
                };
                f32x4_t acc = bv;
                auto mm = [&](bf16x4 (&fr)[6], const bf16x8 (&w)[3]) {
                    bf16x8 a[3];
#pragma unroll
                    for (int pl = 0; pl < NP; pl++) a[pl] = __builtin_shufflevector(fr[2 * pl], fr[2 * pl + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                    acc = b4_mfma<NP>(acc, w, a);
                };
                constexpr int R = 2 * NP;                    // reads per 32-deep step; the tail step reads NP
                bf16x4 ft[3];
                rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{});
                waitF<2 * R, NP>(f[0]); __builtin_amdgcn_sched_barrier(0); mm(f[0], w0[0]);
                static_for<NP>([&](auto pc) {
                    constexpr int pl = decltype(pc)::value;
                    ft[pl] = rd64<JR + pl * PPLANE * 2>(p1t);
                });
                waitF<R + NP, NP>(f[1]); __builtin_amdgcn_sched_barrier(0); mm(f[1], w0[1]);
                waitF<NP, NP>(f[2]); __builtin_amdgcn_sched_barrier(0); mm(f[2], w0[2]);
                waitT<0, NP>(ft); __builtin_amdgcn_sched_barrier(0);
                // its own accumulator: a 16x16x16 MFMA whose SrcC is the result of a 16x16x32 MFMA issued just before returned wrong sums
                // (the compiler inserts no wait states for that mixed pair here; with two s_nop 15 in between the results were right).
                // Two chains of one opcode each have no such dependency; they meet in four v_add.
                f32x4_t acct = {0.f, 0.f, 0.f, 0.f};
                if constexpr (NP == 2) {
                    acct = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, w0t[2]), __builtin_bit_cast(f16x4, ft[1]), acct, 0, 0, 0);
                    acct = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, w0t[1]), __builtin_bit_cast(f16x4, ft[0]), acct, 0, 0, 0);
                    acct = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, w0t[0]), __builtin_bit_cast(f16x4, ft[0]), acct, 0, 0, 0);
                } else {
                if constexpr (NP == 3) {
                    acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[0], ft[2], acct, 0, 0, 0);
                    acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[2], ft[0], acct, 0, 0, 0);
                    acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[1], ft[1], acct, 0, 0, 0);
                    acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[0], ft[1], acct, 0, 0, 0);
                    acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[1], ft[0], acct, 0, 0, 0);
                }
                acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[0], ft[0], acct, 0, 0, 0);
                }
                acc += acct;
                // epilogue: D (transposed) row 4g + r = (dx, co0 + r), column m = pixel pair.  Outside the image = block_4_1's zero padding.
                const bool ok = col_ok && (unsigned)(Ry0 + wrow + HW * j) < (unsigned)H0;
                uint32_t pa[3], pb[3];
                s3p::act_split<NP>(acc[0], acc[1], pa, ok);
                s3p::act_split<NP>(acc[2], acc[3], pb, ok);
                constexpr int JW = HW * j * 2 * XH * 16;     // bytes: HW image rows down
                static_for<NP>([&](auto pc) {
                    constexpr int pl = decltype(pc)::value;
                    wr64<JW + pl * PLANE * 2>(st1a, make_uint2(pa[pl], pb[pl]));
                });
            }
        });
        B4_T(2);
#if HNET_B4_ABLATE != 5
        // the pixel pairs of columns 64..66 (two per region row) form N_MT0 - N_REG = 3 more M-tiles, one each for waves 3, 2, 1: slot 16 k + m = (region row, pair 32 / 33).
        // Same immediate-addressed form as the regular tiles on lane-invariant addresses of their own (round 4; until then a loop with per-tile address arithmetic,
        // reads and MFMAs in lockstep: 1150 cycles per tile for 192 cycles of MFMA - tools/trace_b4.hip)
        if (has_left) {                                      // wave-uniform
            bf16x4 f[3][6], ft[3];
            static_for<3>([&](auto sc) {
                constexpr int st = decltype(sc)::value;
                static_for<NP>([&](auto pc) {
                    constexpr int pl = decltype(pc)::value;
                    f[st][2 * pl] = rd64<pl * PPLANE * 2>(pL[st]);
                    f[st][2 * pl + 1] = rd64<pl * PPLANE * 2 + 8>(pL[st]);
                });
            });
            static_for<NP>([&](auto pc) {
                constexpr int pl = decltype(pc)::value;
                ft[pl] = rd64<pl * PPLANE * 2>(pLt);
            });
            f32x4_t acc = bv;
            auto mm = [&](bf16x4 (&fr)[6], const bf16x8 (&w)[3]) {
                bf16x8 a[3];
#pragma unroll
                for (int pl = 0; pl < NP; pl++) a[pl] = __builtin_shufflevector(fr[2 * pl], fr[2 * pl + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                acc = b4_mfma<NP>(acc, w, a);
            };
            constexpr int R = 2 * NP;
            waitF<2 * R + NP, NP>(f[0]); __builtin_amdgcn_sched_barrier(0); mm(f[0], w0[0]);
            waitF<R + NP, NP>(f[1]); __builtin_amdgcn_sched_barrier(0); mm(f[1], w0[1]);
            waitF<NP, NP>(f[2]); __builtin_amdgcn_sched_barrier(0); mm(f[2], w0[2]);
            waitT<0, NP>(ft); __builtin_amdgcn_sched_barrier(0);
            f32x4_t acct = {0.f, 0.f, 0.f, 0.f};         // (own accumulator chain: see the regular tiles)
            if constexpr (NP == 2) {
                acct = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, w0t[2]), __builtin_bit_cast(f16x4, ft[1]), acct, 0, 0, 0);
                acct = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, w0t[1]), __builtin_bit_cast(f16x4, ft[0]), acct, 0, 0, 0);
                acct = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, w0t[0]), __builtin_bit_cast(f16x4, ft[0]), acct, 0, 0, 0);
            } else {
                if constexpr (NP == 3) {
                    acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[0], ft[2], acct, 0, 0, 0);
                    acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[2], ft[0], acct, 0, 0, 0);
                    acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[1], ft[1], acct, 0, 0, 0);
                    acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[0], ft[1], acct, 0, 0, 0);
                    acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[1], ft[0], acct, 0, 0, 0);
                }
                acct = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w0t[0], ft[0], acct, 0, 0, 0);
            }
            acc += acct;
            // outside the image = block_4_1's zero padding; slots beyond the region's last row repeat it (same address, same value); column 67 lands in an unread chunk
            const bool ok = (unsigned)(Ry0 + rowL) < (unsigned)H0 && (unsigned)(Rx0 + colL) < (unsigned)W0;
            uint32_t pa[3], pb[3];
            s3p::act_split<NP>(acc[0], acc[1], pa, ok);
            s3p::act_split<NP>(acc[2], acc[3], pb, ok);
            static_for<NP>([&](auto pc) {
                constexpr int pl = decltype(pc)::value;
                wr64<pl * PLANE * 2>(stL, make_uint2(pa[pl], pb[pl]));
            });
        }
#endif
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's asm ds_writes (the barrier's own wait does not count them)
        B4_T(3);
        if constexpr (DMA) {
            if constexpr (WARPIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next patch's box has landed (issued a phase 1 ago)
            __builtin_amdgcn_s_barrier();                    // raw barrier: a __syncthreads() would also drain the stores of phase 2
            asm volatile("" ::: "memory");
            B4_T(4);
            if constexpr (WARPIN) {
                if (tile + (int)gridDim.x < n_tiles) warp_sample();              // the patch is dead: the next tile's is sampled into it
                B4_T(6);
            } else if (tile + (int)gridDim.x < n_tiles) dma_issue(tile + gridDim.x);    // the patch is dead: the next tile's copy runs under phase 2
        } else {
            __syncthreads();
        }

        // ---- phase 2: block_4_1 from the S3 image; fully unrolled, immediate addressing, three steps of reads in flight
        // flags bit 6: the bordered layout of kernels.h B42_* (the fused block_4_2 + block_4_3 kernel copies its patches from it by LDS-DMA)
        unsigned char* const obase = reinterpret_cast<unsigned char*>(out16) + ((size_t)b * o_img + o_org + (size_t)(ty0 + wrow) * o_wp + tx0 + whalf * 16) * 32;
#if HNET_B4_ABLATE != 1
        static_for<J2>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (wave + WAVES * j < N_MT1) {                  // wave-uniform
                constexpr int JR = HW * j * 2 * (2 * XH * 16);   // bytes: HW output rows = 2 HW image rows down
                f32x4_t acc = bv1;
                if constexpr (C::P2B128) {
                    // one ds_read_b128 per plane and step (16 bytes = the 8 channels of the lane's tap pixel); four steps in flight
                    bf16x8 q[7][3];
                    auto rd = [&](auto sc) {
                        constexpr int st = decltype(sc)::value;
                        static_for<NP>([&](auto pc) {
                            constexpr int pl = decltype(pc)::value;
                            q[st][pl] = rd128i<JR + pl * PLANE * 2>(p2a[st]);
                        });
                    };
                    rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{}); rd(std::integral_constant<int, 3>{});
                    static_for<7>([&](auto sc) {
                        constexpr int st = decltype(sc)::value;
                        constexpr int ahead = st + 3 < 7 ? 3 : 6 - st;       // steps in flight behind the one consumed
                        waitQ<ahead * NP>(q[st]); __builtin_amdgcn_sched_barrier(0);
                        acc = b4_mfma<NP>(acc, w1[st], q[st]);
                        if constexpr (st + 4 < 7) rd(std::integral_constant<int, st + 4>{});
                    });
                } else {
                bf16x4 f[7][6];
                auto rd = [&](auto sc) {
                    constexpr int st = decltype(sc)::value;
                    static_for<NP>([&](auto pc) {
                        constexpr int pl = decltype(pc)::value;
                        f[st][2 * pl] = rd64<JR + pl * PLANE * 2>(p2a[st]);
                        f[st][2 * pl + 1] = rd64<JR + pl * PLANE * 2>(p2b[st]);
                    });
                };
                auto mm = [&](bf16x4 (&fr)[6], const bf16x8 (&w)[3]) {
                    bf16x8 a[3];
#pragma unroll
                    for (int pl = 0; pl < NP; pl++) a[pl] = __builtin_shufflevector(fr[2 * pl], fr[2 * pl + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                    acc = b4_mfma<NP>(acc, w, a);
                };
                constexpr int R = 2 * NP;                    // reads per step
                rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{});
                waitF<2 * R, NP>(f[0]); __builtin_amdgcn_sched_barrier(0); mm(f[0], w1[0]);
                rd(std::integral_constant<int, 3>{});
                waitF<2 * R, NP>(f[1]); __builtin_amdgcn_sched_barrier(0); mm(f[1], w1[1]);
                rd(std::integral_constant<int, 4>{});
                waitF<2 * R, NP>(f[2]); __builtin_amdgcn_sched_barrier(0); mm(f[2], w1[2]);
                rd(std::integral_constant<int, 5>{});
                waitF<2 * R, NP>(f[3]); __builtin_amdgcn_sched_barrier(0); mm(f[3], w1[3]);
                rd(std::integral_constant<int, 6>{});
                waitF<2 * R, NP>(f[4]); __builtin_amdgcn_sched_barrier(0); mm(f[4], w1[4]);
                waitF<1 * R, NP>(f[5]); __builtin_amdgcn_sched_barrier(0); mm(f[5], w1[5]);
                waitF<0, NP>(f[6]); __builtin_amdgcn_sched_barrier(0); mm(f[6], w1[6]);
                }
                // D (transposed): row 4g + r = cout, column m = output pixel: 8 bytes (4 channels) per lane and plane
                uint32_t pa[3], pb[3];
                s3p::act_split<NP>(acc[0], acc[1], pa);
                s3p::act_split<NP>(acc[2], acc[3], pb);
                unsigned char* const orow = obase + (size_t)(HW * j) * o_wp * 32;    // wave-uniform
                if constexpr (DMA) {
#if HNET_B4_ABLATE == 3
                    asm volatile("" ::"v"(pa[0]), "v"(pb[0]), "v"(pa[1]), "v"(pb[1]), "v"(orow));
#else
#pragma unroll
                    for (int pl = 0; pl < NP; pl++) *reinterpret_cast<uint2*>(orow + (size_t)pl * o_plane * 2 + gv) = make_uint2(pa[pl], pb[pl]);
#endif
                } else {
                    static_for<NP>([&](auto pc) {
                        constexpr int pl = decltype(pc)::value;
                        wr64<pl * 512>(st2w, make_uint2(pa[pl], pb[pl]));
                    });
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    u32x4 o0 = rd128(st2r0), o1;
                    if constexpr (NP == 3) o1 = rd128(st2r1);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o0), "+v"(o1));
                    __builtin_amdgcn_sched_barrier(0);
                    if (NP >= 2 || lane < 32) *reinterpret_cast<u32x4*>(orow + gvo0) = o0;       // NP x 32 pieces: 64 lanes cover two planes
                    if (NP == 3 && lane < 32) *reinterpret_cast<u32x4*>(orow + gvo1) = o1;
                }
            }
        });
#endif
        B4_T(5);
    }   // persistent tile loop
#ifdef HNET_B4_TRACE
    if (blockIdx.x < 8 && lane == 0) for (int k = 0; k < 8; k++) g_b4_trace[(blockIdx.x * 4 + wave) * 9 + k] = tr_acc[k];
    if (blockIdx.x < 8 && lane == 0) g_b4_trace[(blockIdx.x * 4 + wave) * 9 + 8] = (unsigned long long)(tile_no - 1);
#endif
}

}  // namespace hnet
