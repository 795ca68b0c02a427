// igemm_s3.h — implicit-GEMM convolution on the bf16 matrix cores with fp32-grade accuracy ("split-bf16 x3").
//
// CDNA4 has no TF32/xf32; its exact-fp32 MFMA runs at 1/16 of the bf16 rate and, sustained, is power limited
// to ~100-125 TFLOP/s on this part (tools/mfma_peak.hip).  Every fp32 value v is therefore carried as THREE bf16
// planes  v = v1 + v2 + v3  (v1 = bf16(v), v2 = bf16(v - v1), v3 = bf16(v - v1 - v2): 3 x 8 significant bits = the
// 24 bits of fp32, the split is exact), and a product a*b is evaluated as the six partial products
//     a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1        (dropped: a2b3, a3b2, a3b3 <= 2^-24 |ab| each)
// by six v_mfma_f32_32x32x16_bf16 accumulating in fp32.  bf16 x bf16 products are exact in fp32, so the result has
// the accuracy of an fp32 dot product (measured against the oracle in tests/test_gpu_parity.py) at 6/16 of the
// fp32-MFMA issue cycles.
//
// Data layout ("S3"): an activation tensor is three NHWC bf16 planes [3][B][H][W][C]; weights are pre-split on the
// host to [3][Cout][Kp] with the same K order as igemm.h ((kh, [kw, ci]) segments).  Producers write S3 in their
// epilogue (one split per output element, amortised over the 6-25 times each element is consumed), so the staging
// path is pure 16-byte copies: global -> registers -> ds_write_b128, no conversion.
//
// Tile: 256 threads = 4 waves, each wave a 32x32 (x TM x TN) accumulator tile; K-tile 32; LDS per operand and plane
// [rows][32] bf16 = 64-byte rows whose four 16-byte chunks are XOR-swizzled with (row >> 2) & 3 so that the
// ds_read_b128 of an MFMA fragment (rows l&31, chunk 2*step + (l>>5)) is bank-conflict free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "igemm.h"
#include "s3_format.h"
#include "heads_mask.h"

namespace hnet {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// NP = number of ACTIVATION planes = the arithmetic mode (s3_format.h): 3 = split-bf16, 1 = plain bf16, 2 = two fp16 planes.
// s3_wplanes: weight planes of the mode; s3_acc_scale: what the accumulator carries relative to the true sum.
template <int NP> constexpr int s3_wplanes = NP == 1 ? 1 : 3;
template <int NP> constexpr float s3_acc_scale = NP == 2 ? S3_F16_SCALE : 1.0f;
template <int NP> __device__ __forceinline__ float s3_descale(float acc) { return NP == 2 ? acc * S3_F16_INV : acc; }
// The implicit-GEMM kernels of this file run the fp16 mode with TWO weight planes (w = W0 + W1 / 4096, the activation split) and two
// accumulators per tile: hi += W0 A0, lo += W1 A0 + W0 A1, result = hi + lo / 4096 - the same three MFMAs per product, a third less
// weight traffic through LDS (these kernels are LDS bound in this mode) and two shorter dependent MFMA chains.  The kernels that keep
// their weights in registers use the three-plane form above (one accumulator).
template <int NP> constexpr int s3_wplanes_gemm = NP == 1 ? 1 : NP;

struct S3Params {
    const uint16_t* A;     // input planes [3][...] NHWC bf16
    size_t a_plane;        // elements per input plane
    const uint16_t* Wp;    // weights [3][N][Kp] bf16
    size_t w_plane;
    const float* bias;     // [N]
    uint16_t* out16;       // S3 output planes (or nullptr)
    size_t o_plane;
    float* out32;          // fp32 output [M][N] (last layer of a block) (or nullptr)
    int M, N, Kp;
    int H, W, Ho, Wo;
    int k_split;           // gridDim.z slices of the K loop (small-M launches), raw fp32 partials -> partial[z][M][N]
    float* partial;
    // heads only (HeadLoaderS3): keep-mask bits [B][n_local][2 heads][640 bytes], bit i of byte j = element 8j+i (NHWC k)
    const uint8_t* mask;
    int n_local;
    const uint16_t* wfrag; // igemm_region.h layers: the weights as MFMA fragments in consumption order (nullptr: not packed)
    int xcd_remap;         // 1: XCD-aware workgroup -> tile mapping (s3_tile_origin)
    int tile;              // tile-shape experiment of the context (HNET_S3_TILE at hnet_create; 0 = the measured defaults of s3_dispatch.h)
    uint32_t* tickets;     // split-K launches of igemm_s3_lean_kernel (round 5): one zeroed word per (M, N) tile; the LAST of the k_split workgroups of a
                           // tile to arrive sums the partials (s3_splitk_last_arriver) - no splitk_reduce* launch.  nullptr: raw partials only
#ifdef HNET_S3_TRACE
    unsigned long long* trace;   // tools/trace_s3.hip only: [block < 8][wave][S3T_SLOTS] s_memtime stamps
#endif
};

// cache policy of the buffer loads / stores that exchange split-K partials between workgroups on different XCDs (each XCD has a private,
// mutually non-coherent L2): sc0 | sc1 = system scope - the store is written through to memory, the load does not hit a stale L2 line
constexpr int S3_CPOL_SYSTEM = 1 | 16;

// phase timestamps of the K loop for tools/trace_s3.hip (compiled out of the library)
#ifdef HNET_S3_TRACE
#define S3T_SLOTS 512
#define S3T()                                                                                                             \
    do {                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        if (blockIdx.y == 0 && blockIdx.x < 8 && tcount < S3T_SLOTS) {                                                    \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                   \
            if (lane == 0) p.trace[(size_t)(blockIdx.x * 4 + wave) * S3T_SLOTS + tcount] = t_;                            \
            tcount++;                                                                                                     \
        }                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    } while (0)
#else
#define S3T() do {} while (0)
#endif

template <int CIN_, int KS_, int STRIDE_, int SEG_>
struct ConvLoaderS3 {
    static constexpr int CIN = CIN_, KS = KS_, STRIDE = STRIDE_, SEG = SEG_;
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int RL = KS * CIN;
    static constexpr int SPR = (RL + SEG - 1) / SEG;
    static constexpr int TOTAL_SEGS = KS * SPR;
    static constexpr int KP = TOTAL_SEGS * SEG;
    static constexpr int SEGMENT = SEG;
    static constexpr bool WIDE_TAPS = CIN >= 64;        // a 64-wide K tile stays inside one tap: 128 contiguous bytes per row and plane
    template <int BK> static constexpr bool lean_ok() { return CIN % BK == 0 && (KS * CIN) % SEG == 0; }   // igemm_s3_lean_kernel: one tap per K-tile
    static_assert(IG_BK % SEG == 0 && SEG % 8 == 0 && CIN % 8 == 0, "16-byte chunks must stay inside one pixel");
    static_assert(RL % SEG == 0, "no padded segments for Cin >= 8 layers");

    struct Row { int pix0, iy0, ix0; bool valid; };
    static constexpr bool HAS_MASK = false;
    __device__ static inline uint32_t mask_byte(const S3Params&, const Row&, int) { return 0xFFu; }

    __device__ static inline Row make_row(const S3Params& p, int m, int /*n0*/) {
        Row r;
        r.valid = m < p.M;
        const int mm = r.valid ? m : 0;
        const int hw = p.Ho * p.Wo;
        const int b = mm / hw;
        const int rem = mm - b * hw;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        r.iy0 = oy * STRIDE - PAD;
        r.ix0 = ox * STRIDE - PAD;
        r.pix0 = (b * p.H + r.iy0) * p.W + r.ix0;
        return r;
    }
    // element offset (within a plane) of the 8 K-values starting at padded-K index kp, and whether they are real
    __device__ static inline size_t offset(const S3Params& p, const Row& r, int kp, bool& ok) {
        const int sg = kp / SEG, within = kp % SEG;
        const int kh = sg / SPR;
        const int rr = (sg % SPR) * SEG + within;
        const int kw = rr / CIN, ci = rr % CIN;
        const int iy = r.iy0 + kh, ix = r.ix0 + kw;
        ok = r.valid && sg < TOTAL_SEGS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        return ok ? ((size_t)(r.pix0 + kh * p.W + kw)) * CIN + ci : 0;
    }
};

// packed epilogue helpers shared by the direct-convolution kernels (conv_first.h, conv_patch_s2.h, conv_b4_fused.h)
namespace s3p {
// XCD-aware tile order of the persistent / one-tile-per-workgroup kernels: slot t (= workgroup id + k gridDim.x; consecutive workgroup
// ids go to consecutive XCDs, each with a private L2) -> tile id such that XCD x owns the contiguous range [x n/8, (x + 1) n/8) and the
// tiles resident on one XCD at any time are neighbours in the image, whose halos then hit in that L2 (block4_fused_kernel: FETCH_SIZE
// 518 -> 246 MB per launch for 242 MB of input).  Needs n and the grid to be multiples of 8; identity otherwise.
__device__ __forceinline__ int xcd_tile(int t, int n, int grid) { return ((n | grid) & 7) == 0 ? (t & 7) * (n >> 3) + (t >> 3) : t; }
__device__ __forceinline__ uint32_t cvt_pk(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// two fp32 values -> their NP bf16 planes, each plane as one packed dword (lo = v0, hi = v1)
__device__ __forceinline__ uint32_t cvt_pk_f16(float lo, float hi) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 h = {(_Float16)lo, (_Float16)hi};          // round to nearest even (v_cvt_pk_f16_f32 on gfx950)
    return __builtin_bit_cast(uint32_t, h);
}
// the two halves of a packed fp16 pair back to fp32: one instruction each (left to the compiler, the high half was re-converted from
// the fp32 source with a second v_cvt_f16_f32 instead of being read from the packed register)
__device__ __forceinline__ float cvt_f32_f16_lo(uint32_t packed) {
    float f;
    asm("v_cvt_f32_f16_e32 %0, %1" : "=v"(f) : "v"(packed));
    return f;
}
__device__ __forceinline__ float cvt_f32_f16_hi(uint32_t packed) {
    float f;
    asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f) : "v"(packed));
    return f;
}
typedef float f32x2_p __attribute__((ext_vector_type(2)));      // value pairs: v_pk_mul_f32 / v_pk_fma_f32
// Second fp16 plane of a value pair in TWO instructions (round 5): A1 = f16(c - 4096 A0) with A0 taken straight from the packed fp16 register by
// v_fma_mixlo_f16 / v_fma_mixhi_f16 (fp32 fma with an fp16 source, result rounded to fp16 into one half of the destination) - instead of two
// v_cvt_f32_f16, one v_pk_fma_f32 and one v_cvt_pk_f16_f32.  c - 4096 A0 is exact in fp32 (|c - 4096 A0| <= half an fp16 ulp of 4096 A0, on fp32's
// grid), so both forms round the same number once to fp16: same bits (tools/out_hash.py before / after; the goldens).  c0, c1 = the values at
// 4096 x scale, a0 = packed f16(c / 4096).
__device__ __forceinline__ uint32_t f16_residual_pk(uint32_t a0, float c0, float c1) {
    uint32_t r;
    const float ms = -S3_F16_SCALE;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a0), "s"(ms), "v"(c0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(a0), "s"(ms), "v"(c1));
    return r;
}
template <int NP>
__device__ __forceinline__ void split_pair(float v0, float v1, uint32_t (&pl)[3]) {
    if constexpr (NP == 2) {         // fp16 planes: value = A0 + A1 / 4096, A1 = f16((v - A0) 4096) = f16(4096 v - 4096 A0): four instructions (six until round 4)
        pl[0] = cvt_pk_f16(v0, v1);
        const f32x2_p c = f32x2_p{v0, v1} * S3_F16_SCALE;               // exact (a power of two; |v| < 2^15)
        pl[1] = f16_residual_pk(pl[0], c[0], c[1]);
        return;
    }
    pl[0] = cvt_pk(v0, v1);
    if constexpr (NP == 3) {
        const float r0 = v0 - __builtin_bit_cast(float, pl[0] << 16), r1 = v1 - __builtin_bit_cast(float, pl[0] & 0xffff0000u);
        pl[1] = cvt_pk(r0, r1);
        const float s0 = r0 - __builtin_bit_cast(float, pl[1] << 16), s1 = r1 - __builtin_bit_cast(float, pl[1] & 0xffff0000u);
        pl[2] = cvt_pk(s0, s1);
    }
}
// max(a, b) as ONE v_max_f32 (round 5): fmaxf() is llvm.maxnum, which the backend brackets with a canonicalising v_max_f32 x, x per operand that comes out
// of an accumulator (four extra vector instructions per LeakyReLU of a value pair in kernels that are issue bound).  Same result for every non-NaN input;
// a NaN accumulator stays NaN-or-the-other-operand as v_max_f32 defines it - such a forward raises the overflow flag either way.
__device__ __forceinline__ float vmax1(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float lrelu(float v) { return vmax1(v, v * 0.1f); }
// LeakyReLU of an accumulator of mode NP (bias already inside, at the accumulator's scale)
template <int NP> __device__ __forceinline__ float act(float acc) { return lrelu(s3_descale<NP>(acc)); }
// two accumulator values of mode NP (bias inside) -> LeakyReLU -> the planes of the pair; `ok` = false writes zeros.
// fp16 mode: the accumulator carries 4096 x the sum.  u = max(acc, 0.1 acc) at that scale, v = u / 4096 (exact), A0 = f16(v),
// A1 = f16((v - A0) 4096) = f16(u - 4096 A0): bit for bit split_pair(act(acc0), act(acc1)) - scaling by a power of two commutes with the
// rounding of the 0.1 multiple - in 11 vector instructions per pair instead of 18.
template <int NP>
__device__ __forceinline__ void act_split(float a0, float a1, uint32_t (&pl)[3], bool ok = true) {
    if constexpr (NP == 2) {
        f32x2_p u = {a0, a1};
        const f32x2_p t = u * 0.1f;
        u[0] = vmax1(u[0], t[0]);
        u[1] = vmax1(u[1], t[1]);
        if (!ok) u = f32x2_p{0.f, 0.f};
        const f32x2_p v = u * S3_F16_INV;
        pl[0] = cvt_pk_f16(v[0], v[1]);
        pl[1] = f16_residual_pk(pl[0], u[0], u[1]);      // f16(u - 4096 A0): 9 vector instructions per pair (11 until round 4)
    } else {
        split_pair<NP>(ok ? act<NP>(a0) : 0.f, ok ? act<NP>(a1) : 0.f, pl);
    }
}
// one value -> its planes
template <int NP>
__device__ __forceinline__ void split1(float v, uint16_t& a, uint16_t& b, uint16_t& c) {
    if constexpr (NP == 3) split3(v, a, b, c);
    else if constexpr (NP == 2) split2h(v, a, b);
    else a = f32_to_bf16_rn(v);
}

}  // namespace s3p

// MC-dropout input of the heads in S3 form: A[(b, s)][k] = keep(s, k) ? featS3[b][k] : 0, where featS3 are the three
// bf16 planes of feat * 1/(1-p) (heads_prep_kernel) and the keep bits come from a precomputed bit array (one byte per
// 8-element chunk) so that the staging path does no hashing (reference: Dropout -> Linear(5120,256),
// model_to_trace.py:222-225,229-232).
struct HeadLoaderS3 {
    static constexpr int KP = 5120;
    static constexpr int SEGMENT = 32;
    static constexpr bool WIDE_TAPS = true;
    static constexpr bool HAS_MASK = true;
    template <int BK> static constexpr bool lean_ok() { return true; }
    struct Row { int b; size_t mrow; bool valid; };
    __device__ static inline Row make_row(const S3Params& p, int m, int n0) {
        Row r;
        r.valid = m < p.M;
        const int mm = r.valid ? m : 0;
        r.b = mm / p.n_local;
        const int head = n0 >> 8;
        r.mrow = ((size_t)mm * 2 + head) * 640;
        return r;
    }
    __device__ static inline size_t offset(const S3Params&, const Row& r, int kp, bool& ok) {
        ok = r.valid;
        return (size_t)r.b * 5120 + kp;
    }
    __device__ static inline uint32_t mask_byte(const S3Params& p, const Row& r, int kp) { return p.mask[r.mrow + (kp >> 3)]; }
};

// Workgroup -> tile mapping.  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, each with its own
// 4 MB L2.  With the plain mapping neighbouring M-tiles — which share most of their input rows (a 3x3 / 5x5 window re-reads
// every value 2.25 / 6.25 times) — land in eight different L2s and each of them fetches its own copy (FETCH_SIZE was
// 1.8-4.2x the input tensor).  Remapped, XCD k owns a contiguous range of the logical tile order (M-tile major, the N-tiles
// of one M-tile adjacent), so those re-reads are L2 hits.
__device__ __forceinline__ void s3_tile_origin(const S3Params& p, int BM, int BN, int& m0, int& n0) {
    if (!p.xcd_remap) { m0 = blockIdx.x * BM; n0 = blockIdx.y * BN; return; }
    const int nx = gridDim.x, ny = gridDim.y, total = nx * ny;
    const int lin = blockIdx.x + blockIdx.y * nx;
    const int xcd = lin & 7, idx = lin >> 3;
    const int base = total >> 3, rem = total & 7;
    const int L = xcd * base + min(xcd, rem) + idx;           // XCD k holds base + (k < rem) consecutive tiles
    m0 = (L / ny) * BM;
    n0 = (L % ny) * BN;
}

// chunk swizzle of an LDS tile row: 64-byte rows (4 chunks) pair four rows per 256-byte bank row, 128-byte rows (8 chunks) two
template <int CH>
__device__ __forceinline__ int s3_swz(int row, int chunk) {
    return CH == 4 ? (chunk ^ ((row >> 2) & 3)) * 8 : (chunk ^ ((row >> 1) & 7)) * 8;   // bf16 elements
}

// OUT32 = true: fp32 [M][N] output (feeds an FC);  false: S3 planes
// NBUF = 2: double-buffered LDS, one barrier per K-tile.  NBUF = 1: single LDS buffer, two barriers per K-tile but half
// the LDS, i.e. twice the resident workgroups per CU: with 6 MFMAs x 32 cycles per k16-step a K-tile lasts ~400 cycles,
// less than the latency of its own prefetch, so the latency has to be hidden by more workgroups instead.
// BKT = K-tile (32 or 64 K-values).  With 64 every staged row is a full 128-byte line per plane: the texture addresser
// (GRBM_TA_BUSY ~ 90 % on the 32-wide tiles, profiles/r01) handles half as many lines per byte.
// epilogue of the transposed 16x16x32 tiles (shared by the register-staged and the LDS-DMA kernel): lane (m = lane&15,
// g = lane>>4) holds channels 4g .. 4g+3 of GEMM row m of every 16x16 tile; (mw, nw) = origin of the wave's tile
typedef float f32x4_m16 __attribute__((ext_vector_type(4)));
template <int TM16, int TN16, bool OUT32, int NP = 3, bool SCALED = (NP == 2)>
__device__ __forceinline__ void s3_epilogue_m16(f32x4_m16 (&acc16)[TM16][TN16], const S3Params& p, uint16_t* st_wave, int mw, int nw, int lane) {
    // ---- epilogue of the transposed 16x16 tiles: lane (m = lane&15, g = lane>>4) holds channels n = 4g .. 4g+3 of GEMM row m
    typedef float f32x4_e __attribute__((ext_vector_type(4)));
    const int em = lane & 15, eg = lane >> 4;
    if (p.k_split > 1 || OUT32) {
        float* dst = p.k_split > 1 ? p.partial + (size_t)blockIdx.z * p.M * p.N : p.out32;
#pragma unroll
        for (int j = 0; j < TN16; j++) {
            const int n = nw + j * 16 + 4 * eg;
            f32x4_e bv = {0.f, 0.f, 0.f, 0.f};
            if (p.k_split == 1 && n < p.N) bv = *reinterpret_cast<const f32x4_e*>(p.bias + n);
#pragma unroll
            for (int i = 0; i < TM16; i++) {
                const int m = mw + i * 16 + em;
                if (m < p.M && n < p.N) {
                    f32x4_e v = acc16[i][j];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = SCALED ? s3_descale<NP>(v[e]) : v[e];      // split-K partials are written at the true scale too
                    if (p.k_split == 1) {
#pragma unroll
                        for (int e = 0; e < 4; e++) { const float x = v[e] + bv[e]; v[e] = x > 0.0f ? x : x * 0.1f; }
                    }
                    if (p.k_split > 1 && p.tickets)       // read by the tile's last arriver on another XCD: written through to memory (s3_splitk_last_arriver)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, 0x7FFFFFF0, 0x00020000),
                                                               (uint32_t)(((size_t)m * p.N + n) * 4), 0, S3_CPOL_SYSTEM);
                    else
                        *reinterpret_cast<f32x4_e*>(dst + (size_t)m * p.N + n) = v;
                }
            }
        }
    } else {
        // S3 planes: per 16 x 32 piece of the wave's tile, 8-byte pieces (4 channels of one row) into wave-private LDS
        // [plane][16 rows][32 cols], 16-byte chunks XOR-swizzled with (row >> 1) & 3, then 16 bytes per lane to global
        static_assert(TN16 % 2 == 0, "wave tile width is a multiple of 32");
        uint16_t* st = st_wave;
#pragma unroll
        for (int sj = 0; sj < TN16 / 2; sj++)
#pragma unroll
            for (int i = 0; i < TM16; i++) {
#pragma unroll
                for (int dj = 0; dj < 2; dj++) {
                    const int j = 2 * sj + dj;
                    const int nloc = dj * 16 + 4 * eg;
                    const int n = nw + sj * 32 + nloc;
                    f32x4_e bv = {0.f, 0.f, 0.f, 0.f};
                    if (n < p.N) bv = *reinterpret_cast<const f32x4_e*>(p.bias + n);
                    uint16_t sp[3][4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float v = (SCALED ? s3_descale<NP>(acc16[i][j][e]) : acc16[i][j][e]) + bv[e];
                        v = v > 0.0f ? v : v * 0.1f;
                        s3p::split1<NP>(v, sp[0][e], sp[1][e], sp[2][e]);
                    }
                    const int chunk = (nloc >> 3) ^ ((em >> 1) & 3);
                    const int e0 = em * 32 + chunk * 8 + (nloc & 7);
#pragma unroll
                    for (int pl = 0; pl < NP; pl++)
                        *reinterpret_cast<uint2*>(&st[pl * 16 * 32 + e0]) =
                            make_uint2((uint32_t)sp[pl][0] | ((uint32_t)sp[pl][1] << 16), (uint32_t)sp[pl][2] | ((uint32_t)sp[pl][3] << 16));
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                const int mb = mw + i * 16, nb = nw + sj * 32;
#pragma unroll
                for (int q = 0; q < NP; q++) {
                    const int piece = q * 64 + lane;             // NP planes x 16 rows x 4 chunks of 16 B
                    const int pl = piece >> 6, rem = piece & 63, row = rem >> 2, ch = rem & 3;
                    const int m = mb + row, n = nb + ch * 8;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(&st[(pl * 16 + row) * 32 + (ch ^ ((row >> 1) & 3)) * 8]);
                    if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(p.out16 + pl * p.o_plane + (size_t)m * p.N + n) = v;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
            }
    }
}

// product group of the transposed 16x16x32 tiles (weights as A operand): six split-bf16 partial products, smallest first;
// NP = 1: the single plain-bf16 product
template <int NP>
__device__ __forceinline__ f32x4_m16 s3_mfma16(f32x4_m16 acc, const bf16x8 (&w)[3], const bf16x8 (&a)[3]) {
    if constexpr (NP == 2) {     // fp16 planes: W2 A1 + W1 A0 + W0 A0 = 4096 a w (s3_format.h), smallest first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[2]), __builtin_bit_cast(f16x8, a[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[1]), __builtin_bit_cast(f16x8, a[0]), acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[0]), __builtin_bit_cast(f16x8, a[0]), acc, 0, 0, 0);
    }
    if constexpr (NP == 3) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], a[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], a[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], a[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], a[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], a[0], acc, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], a[0], acc, 0, 0, 0);
}

// fp16 mode of the GEMM kernels: two weight planes, two accumulators (see s3_wplanes_gemm)
__device__ __forceinline__ void s3_mfma16_2acc(f32x4_m16& hi, f32x4_m16& lo, const bf16x8 (&w)[3], const bf16x8 (&a)[3]) {
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[0]), __builtin_bit_cast(f16x8, a[1]), lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[1]), __builtin_bit_cast(f16x8, a[0]), lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[0]), __builtin_bit_cast(f16x8, a[0]), hi, 0, 0, 0);
}
// the same product group on 32x32x16 tiles (weights as A operand)
typedef float f32x16_m32 __attribute__((ext_vector_type(16)));
template <int NP>
__device__ __forceinline__ f32x16_m32 s3_mfma32(f32x16_m32 acc, const bf16x8 (&w)[3], const bf16x8 (&a)[3]) {
    if constexpr (NP == 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[2]), __builtin_bit_cast(f16x8, a[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[1]), __builtin_bit_cast(f16x8, a[0]), acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[0]), __builtin_bit_cast(f16x8, a[0]), acc, 0, 0, 0);
    }
    if constexpr (NP == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], a[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], a[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], a[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], a[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], a[0], acc, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], a[0], acc, 0, 0, 0);
}

// MF = MFMA shape: 32 -> v_mfma_f32_32x32x16_bf16; 16 -> v_mfma_f32_16x16x32_bf16 issued with the weights as A operand
// (transposed tile: a lane holds four consecutive output channels of one GEMM row).  Same LDS traffic per flop; the
// 16x16x32 form sustains a higher clock under the package power limit (MI355X_MICROARCH.md: 1.12-1.15x in MFMA-paced loops).
// NP = number of bf16 planes: 3 = split-bf16, 1 = plain bf16 operands (HNET_PREC_BF16; 16x16x32 form only)
template <class L, int BM, int BN, int WGM, bool OUT32, int NBUF = 1, int BKT = 32, int MF = 32, int NP = 3>
static __global__ __launch_bounds__(256) void igemm_s3_kernel(S3Params p) {
    constexpr int BK = BKT;                           // K-values per tile = CH chunks of 8
    constexpr int CH = BK / 8, RPP = 256 / CH;        // chunks per row, rows staged per pass of the 256 threads
    static_assert(BK % L::SEGMENT == 0, "K tile must be a whole number of loader segments");
    constexpr int WGN = 4 / WGM;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int TM16 = WM / 16, TN16 = WN / 16;
    static_assert((MF == 32 && WM % 32 == 0 && WN % 32 == 0) || (MF == 16 && WM % 16 == 0 && WN % 32 == 0 && BK % 32 == 0),
                  "wave tile: multiples of 32x32 (32x32x16 MFMA) or 16x32 (16x16x32 MFMA)");
    constexpr int A_ROWS = (BM + RPP - 1) / RPP, B_ROWS = (BN + RPP - 1) / RPP;   // rows staged per thread and plane
    static_assert(BM % 32 == 0 && BN % 32 == 0, "tile rows");
    constexpr int TILE_A = BM * BK, TILE_B = BN * BK; // bf16 elements per plane and buffer
    static_assert(NP == 3 || MF == 16, "plain bf16 and the fp16 planes exist in the transposed 16x16x32 form");
    constexpr int NW = s3_wplanes_gemm<NP>;           // weight planes

    // [buf][plane][rows][32]; the epilogue reuses it as a store staging area
    constexpr int SMEM_ELEMS = NBUF * (NP * TILE_A + NW * TILE_B) > 4 * 3 * 32 * 32 ? NBUF * (NP * TILE_A + NW * TILE_B) : 4 * 3 * 32 * 32;
    __shared__ __attribute__((aligned(16))) uint16_t smem[SMEM_ELEMS];
    uint16_t* As = smem;
    uint16_t* Bs = smem + NBUF * NP * TILE_A;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    int m0, n0;
    s3_tile_origin(p, BM, BN, m0, n0);
    const int srow = tid / CH, schunk = tid % CH;     // staging: RPP rows x CH chunks per pass

    typename L::Row rows[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; i++) rows[i] = L::make_row(p, (srow + i * RPP) < BM ? m0 + srow + i * RPP : p.M, n0);
    const uint16_t* wsrc[B_ROWS];
    bool wvalid[B_ROWS];
#pragma unroll
    for (int i = 0; i < B_ROWS; i++) {
        const int n = n0 + srow + i * RPP;
        wvalid[i] = n < p.N && (srow + i * RPP) < BN;
        wsrc[i] = p.Wp + (size_t)(wvalid[i] ? n : 0) * p.Kp + schunk * 8;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    f32x4_m16 acc16[TM16][TN16];
#pragma unroll
    for (int i = 0; i < TM16; i++)
#pragma unroll
        for (int j = 0; j < TN16; j++) acc16[i][j] = f32x4_m16{0.f, 0.f, 0.f, 0.f};
    f32x4_m16 acc16l[NP == 2 ? TM16 : 1][NP == 2 ? TN16 : 1];     // fp16 mode: the cross terms, scaled by 4096
#pragma unroll
    for (int i = 0; i < (NP == 2 ? TM16 : 1); i++)
#pragma unroll
        for (int j = 0; j < (NP == 2 ? TN16 : 1); j++) acc16l[i][j] = f32x4_m16{0.f, 0.f, 0.f, 0.f};

    u32x4 areg[A_ROWS][3], breg[B_ROWS][3];
    bool aok[A_ROWS], bok[B_ROWS];
    uint32_t amask[A_ROWS];
    const int n_iter_total = (p.Kp + BK - 1) / BK;
    const int it0 = (int)(((long)blockIdx.z * n_iter_total) / p.k_split);
    const int n_iter = (int)(((long)(blockIdx.z + 1) * n_iter_total) / p.k_split) - it0;

    auto g_load = [&](int it) {
        const int kp = it * BK + schunk * 8;
#pragma unroll
        for (int i = 0; i < A_ROWS; i++) {
            const size_t off = L::offset(p, rows[i], kp, aok[i]);
#pragma unroll
            for (int pl = 0; pl < NP; pl++) areg[i][pl] = *reinterpret_cast<const u32x4*>(p.A + pl * p.a_plane + off);
            if constexpr (L::HAS_MASK) amask[i] = L::mask_byte(p, rows[i], kp < p.Kp ? kp : 0);
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; i++) {
            bok[i] = wvalid[i] && kp < p.Kp;
            const uint16_t* src = bok[i] ? wsrc[i] + it * BK : p.Wp;
#pragma unroll
            for (int pl = 0; pl < NW; pl++) breg[i][pl] = *reinterpret_cast<const u32x4*>(src + pl * p.w_plane);
        }
    };
    auto s_store = [&](int buf) {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < A_ROWS; i++) {
            const int r = srow + i * RPP;
            if (r < BM) {
                u32x4 mk = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                if constexpr (L::HAS_MASK) {   // byte -> 8 x 16-bit lane masks
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t x = (amask[i] >> (2 * j)) & 3u;
                        mk[j] = (x & 1u) * 0xFFFFu | (x >> 1) * 0xFFFF0000u;
                    }
                }
#pragma unroll
                for (int pl = 0; pl < NP; pl++) {
                    u32x4 v = aok[i] ? areg[i][pl] : z;
                    if constexpr (L::HAS_MASK) { v[0] &= mk[0]; v[1] &= mk[1]; v[2] &= mk[2]; v[3] &= mk[3]; }
                    *reinterpret_cast<u32x4*>(&As[(buf * NP + pl) * TILE_A + r * BK + s3_swz<CH>(r, schunk)]) = v;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; i++) {
            const int r = srow + i * RPP;
            if (r < BN) {
#pragma unroll
                for (int pl = 0; pl < NW; pl++)
                    *reinterpret_cast<u32x4*>(&Bs[(buf * NW + pl) * TILE_B + r * BK + s3_swz<CH>(r, schunk)]) = bok[i] ? breg[i][pl] : z;
            }
        }
    };

    g_load(it0);
    s_store(0);
    __syncthreads();

    const int frow = lane & 31, fh = lane >> 5;
#ifdef HNET_S3_TRACE
    int tcount = 0;
#endif
    for (int it = 0; it < n_iter; it++) {
        const int buf = NBUF == 2 ? (it & 1) : 0;
        S3T();                                       // stamp 0: K-tile start
        if (it + 1 < n_iter) g_load(it0 + it + 1);
        S3T();                                       // 1: prefetch issued
        if constexpr (MF == 16) {
            const int r16 = lane & 15, g16 = lane >> 4;
#pragma unroll
            for (int step = 0; step < BK / 32; step++) {
                bf16x8 af[TM16][3], bf[TN16][3];
#pragma unroll
                for (int i = 0; i < TM16; i++) {
                    const int r = wm * WM + i * 16 + r16;
#pragma unroll
                    for (int pl = 0; pl < NP; pl++)
                        af[i][pl] = *reinterpret_cast<const bf16x8*>(&As[(buf * NP + pl) * TILE_A + r * BK + s3_swz<CH>(r, 4 * step + g16)]);
                }
#pragma unroll
                for (int j = 0; j < TN16; j++) {
                    const int r = wn * WN + j * 16 + r16;
#pragma unroll
                    for (int pl = 0; pl < NW; pl++)
                        bf[j][pl] = *reinterpret_cast<const bf16x8*>(&Bs[(buf * NW + pl) * TILE_B + r * BK + s3_swz<CH>(r, 4 * step + g16)]);
                }
#pragma unroll
                for (int i = 0; i < TM16; i++)
#pragma unroll
                    for (int j = 0; j < TN16; j++) { // weights as A operand: D' row 4g + r = output channel, column = GEMM row
                        if constexpr (NP == 2) s3_mfma16_2acc(acc16[i][j], acc16l[i][j], bf[j], af[i]);
                        else acc16[i][j] = s3_mfma16<NP>(acc16[i][j], bf[j], af[i]);
                    }
            }
        } else
#pragma unroll
        for (int step = 0; step < BK / 16; step++) {
            bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int r = wm * WM + i * 32 + frow;
#pragma unroll
                for (int pl = 0; pl < 3; pl++)
                    af[i][pl] = *reinterpret_cast<const bf16x8*>(&As[(buf * NP + pl) * TILE_A + r * BK + s3_swz<CH>(r, 2 * step + fh)]);
            }
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int r = wn * WN + j * 32 + frow;
#pragma unroll
                for (int pl = 0; pl < 3; pl++)
                    bf[j][pl] = *reinterpret_cast<const bf16x8*>(&Bs[(buf * NP + pl) * TILE_B + r * BK + s3_swz<CH>(r, 2 * step + fh)]);
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    // smallest partial products first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the consumers of the prefetched registers behind the MFMAs (see igemm.h)
        S3T();                                       // 2: MFMAs issued
        if constexpr (NBUF == 2) {
            if (it + 1 < n_iter) s_store(buf ^ 1);
            S3T();
            __syncthreads();
            S3T();
            S3T();
        } else {
            __syncthreads();                 // every wave has read tile `it`
            S3T();                                   // 3: past barrier 1
            if (it + 1 < n_iter) s_store(0);
            S3T();                                   // 4: staged registers written to LDS
            __syncthreads();
            S3T();                                   // 5: past barrier 2
        }
    }

    if constexpr (MF == 16) {
        if constexpr (NP == 2) {
#pragma unroll
            for (int i = 0; i < TM16; i++)
#pragma unroll
                for (int j = 0; j < TN16; j++) acc16[i][j] += acc16l[i][j] * S3_F16_INV;
        }
        s3_epilogue_m16<TM16, TN16, OUT32, NP, false>(acc16, p, smem + wave * (3 * 32 * 32), m0 + wm * WM, n0 + wn * WN, lane);
        return;
    }
    // ---- epilogue: bias + LeakyReLU(0.1); D layout: col n = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col = lane & 31, rbase = 4 * fh;
    if (p.k_split > 1) {   // raw partial sums; splitk_reduce*_kernel applies bias / activation / split
        float* dst = p.partial + (size_t)blockIdx.z * p.M * p.N;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + wn * WN + j * 32 + col;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                    if (m < p.M && n < p.N) dst[(size_t)m * p.N + n] = acc[i][j][r];
                }
        }
        return;
    }
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
        // the K loop's last barrier has passed: the staging LDS is free; 6 KB per wave
        igemm_store_s3<TM, TN>(acc, smem + wave * (3 * 32 * 32), p.bias, p.out16, p.o_plane, p.M, p.N, m0 + wm * WM,
                               n0 + wn * WN, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// Lean-staging variant of igemm_s3_kernel (round 2).  rocprofv3 --pmc on the round-1 kernels (profiles/r02_*): 3.3-3.8 VALU
// instructions per MFMA — per K-tile and wave 110 (64x64 conv tiles) to 366 (heads) VALU for 48 / 96 MFMAs — and VALU busy +
// MFMA busy = 87-98 % of the SIMD cycles with almost no overlap: on CDNA4 a VALU instruction and an MFMA share the SIMD's
// vector issue (MI355X_MICROARCH.md, per-instruction cycle constants), so the GEMMs were bound by the ADDRESS ARITHMETIC of
// their operand staging (im2col offsets, 64-bit pointer adds, zero selects on 16-byte registers, mask expansion), not by
// the matrix pipe.  Here
//   * operands are fetched with buffer loads: address = descriptor base (SGPRs) + per-row byte offset (one VGPR, set up
//     once per workgroup) + a SCALAR offset that carries everything that changes with the K-tile and the plane (tap, channel
//     block, plane stride).  A K-tile of these layers lies inside one filter tap (BK divides Cin), so the tap is wave-uniform;
//   * padding taps and rows beyond M / N are an out-of-range offset: the buffer load returns zeros by itself (no selects);
//   * the registers go to LDS with ds_write_b128 at lane-invariant addresses;
//   * the heads' dropout mask of a chunk (one byte) becomes its 16-byte AND mask through a 4 KB table in LDS.
// Same tiles, same K order and same MFMA sequence as igemm_s3_kernel<..., MF = 16>: results are bit-identical.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t S3_OOB = 0x80000000u;        // voffset >= num_records of every descriptor built here (all < 2 GB): reads as zero

template <class L> struct LeanRow;              // per staged row: byte offset of its window origin (may be negative), origin, validity
template <int CIN, int KS, int STRIDE, int SEG>
struct LeanRow<ConvLoaderS3<CIN, KS, STRIDE, SEG>> {
    typedef ConvLoaderS3<CIN, KS, STRIDE, SEG> L;
    int off, iy0, ix0;
    bool valid;
    __device__ static inline LeanRow make(const S3Params& p, int m, int /*n0*/, int schunk) {
        const typename L::Row r = L::make_row(p, m, 0);
        LeanRow o;
        o.valid = r.valid;
        o.iy0 = r.iy0; o.ix0 = r.ix0;
        o.off = (r.pix0 * CIN + schunk * 8) * 2;
        return o;
    }
    // K-tile `it` of width BK: the filter tap it lies in (wave-uniform), the byte offset it adds to every row, validity of a row
    struct Tap { int kh, kw, delta; };
    template <int BK>
    __device__ static inline Tap tap(const S3Params& p, int it) {
        static_assert(CIN % BK == 0 && L::RL % SEG == 0, "a K-tile lies inside one tap");
#ifdef HNET_S3_TRACE
        const int kp = (it * BK) % (KS * KS * CIN), t = kp / CIN, ci0 = kp - t * CIN;      // (tools/trace_pipe.hip: K sweeps beyond the filter wrap around)
#else
        const int kp = it * BK, t = kp / CIN, ci0 = kp - t * CIN;
#endif
        Tap x;
        x.kh = t / KS; x.kw = t - x.kh * KS;
        x.delta = ((x.kh * p.W + x.kw) * CIN + ci0) * 2;
        return x;
    }
    __device__ inline uint32_t voffset(const S3Params& p, const Tap& t) const {
        const bool ok = valid && (unsigned)(iy0 + t.kh) < (unsigned)p.H && (unsigned)(ix0 + t.kw) < (unsigned)p.W;
        return ok ? (uint32_t)(off + t.delta) : S3_OOB;
    }
};
template <>
struct LeanRow<HeadLoaderS3> {
    int off;
    uint32_t moff;          // byte offset of this row's mask bytes (plus the chunk) in p.mask
    bool valid;
    __device__ static inline LeanRow make(const S3Params& p, int m, int n0, int schunk) {
        const HeadLoaderS3::Row r = HeadLoaderS3::make_row(p, m, n0);
        LeanRow o;
        o.valid = r.valid;
        o.off = (r.b * 5120 + schunk * 8) * 2;
        o.moff = (uint32_t)(r.mrow + schunk);       // rows beyond M alias row 0 (make_row): the byte is loaded unconditionally, their data is zero anyway
        return o;
    }
    struct Tap { int delta; };
    template <int BK>
    __device__ static inline Tap tap(const S3Params&, int it) { return Tap{it * BK * 2}; }
    __device__ inline uint32_t voffset(const S3Params&, const Tap& t) const { return valid ? (uint32_t)(off + t.delta) : S3_OOB; }
};

// chunk swizzle of the lean kernel's tiles (fragments are read in the 16x16x32 shape only: lane (row r16, chunk group g16)).
// A ds_read_b128 is served in four passes of 16 lanes: rows {0-3, 12-15} of lane group g with rows {4-11} of group g + 1, then the
// complement (tools/lds_probe.hip); a pass is conflict free when its 16 lanes hit 16 different 16-byte slots of the 256-byte bank row.
//   128-byte rows: chunk ^ ((row >> 1) & 7), as s3_swz (5.3 LDS cycles per wave-instruction);
//   64-byte rows:  chunk ^ (3 * ((row >> 3) & 1)) - 5.4 cycles, where s3_swz<4>'s chunk ^ ((row >> 2) & 3) costs 8.1: rows r and r + 4k of
//   the two groups of a pass met in the same slot (conflict share 0.33 of block_1_2's LDS cycles, profiles/r02_v3).
template <int CH>
__device__ __forceinline__ int s3_swz_m16(int row, int chunk) {
    return CH == 4 ? (chunk ^ (((row >> 3) & 1) * 3)) * 8 : (chunk ^ ((row >> 1) & 7)) * 8;   // bf16 elements
}

// Split-K without a second launch (round 5, latency path).  Every workgroup of a split-K launch has written its raw fp32 partial tile to
// p.partial[z]; the k_split workgroups of one (M, N) tile then take a ticket from the tile's counter, and the one that draws the LAST ticket
// adds the partials in z order, applies bias + LeakyReLU and writes the layer output - the arithmetic of splitk_reduce_kernel /
// splitk_reduce_s3_kernel element for element (same bits), without their launch (4 - 6 us per layer at batch 1, twelve layers per forward).
// Memory ordering WITHOUT agent-scope fences: a __threadfence() here is an L2 write-back + invalidate on this eight-XCD part and costs more than
// the launch it saves (measured: block_2_2 13.0 -> 20.5 us, forward 0.211 -> 0.251 ms).  Instead the partials are stored and loaded with the
// system-scope cache policy (S3_CPOL_SYSTEM: written through to memory / never served from a stale L2 line), every wave waits for the
// acknowledgement of its own stores (vmcnt 0) before the workgroup barrier, and only then thread 0 performs the agent-scope atomic; the last
// arriver's loads are issued after the atomic has returned.  The counter is left at zero for the next launch (stream ordered).
template <bool OUT32, int NP>
__device__ __forceinline__ void s3_splitk_last_arriver(const S3Params& p, int m0, int n0, int bm, int bn, uint32_t* lds_word) {
    typedef float f32x4_e __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    // vmcnt(0): this wave's partial stores have been acknowledged by memory.  Inline asm with a memory clobber, not __builtin_amdgcn_s_waitcnt: the builtin is
    // IntrNoMem - nothing at IR level orders it against the buffer-store intrinsics, and a provably empty scoreboard lets later passes drop waits
    // (MI355X_MICROARCH.md, compiler hazard); the asm statement is opaque to both.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t* ticket = p.tickets + (blockIdx.x + blockIdx.y * gridDim.x);
    if (tid == 0) *lds_word = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*lds_word != (uint32_t)(p.k_split - 1)) return;      // (workgroup-uniform)
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)p.partial, 0, 0x7FFFFFF0, 0x00020000);
    const int rows = min(bm, p.M - m0), cq = min(bn, p.N - n0) >> 2;      // N is a multiple of 4 for every layer
    const uint32_t stride = (uint32_t)((size_t)p.M * p.N * 4);             // bytes per partial plane (the workspace is far below 2 GB)
    for (int i = tid; i < rows * cq; i += blockDim.x) {
        const int r = i / cq, q = i - r * cq;
        const size_t e = (size_t)(m0 + r) * p.N + n0 + 4 * q;
        f32x4_e s = __builtin_bit_cast(f32x4_e, __builtin_amdgcn_raw_buffer_load_b128(rP, (uint32_t)(e * 4), 0, S3_CPOL_SYSTEM));
#pragma unroll 8
        for (int z = 1; z < p.k_split; z++) {
            const f32x4_e t = __builtin_bit_cast(f32x4_e, __builtin_amdgcn_raw_buffer_load_b128(rP, (uint32_t)(e * 4), z * stride, S3_CPOL_SYSTEM));
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        const f32x4_e b = *reinterpret_cast<const f32x4_e*>(p.bias + n0 + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; k++) { const float v = s[k] + b[k]; s[k] = v > 0.0f ? v : v * 0.1f; }
        if (OUT32) {
            *reinterpret_cast<f32x4_e*>(p.out32 + e) = s;
        } else {
            uint16_t pl[3][4];
#pragma unroll
            for (int k = 0; k < 4; k++) split_np(s[k], NP, pl[0][k], pl[1][k], pl[2][k]);
#pragma unroll
            for (int q3 = 0; q3 < 3; q3++)
                if (q3 < 2 || NP != 2)       // (the plain-bf16 mode writes three planes like splitk_reduce_s3_kernel; only plane 0 is read)
                    *reinterpret_cast<uint2*>(p.out16 + q3 * p.o_plane + e) =
                        make_uint2((uint32_t)pl[q3][0] | ((uint32_t)pl[q3][1] << 16), (uint32_t)pl[q3][2] | ((uint32_t)pl[q3][3] << 16));
        }
    }
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (launch bounds: the 128 x 128 tiles hold 64 x 64 per wave - 128 accumulator registers in the fp16 mode; asking for two waves per SIMD keeps
// them at <= 256 registers, 226 without scratch, where the default heuristic took 264 and with it half of the occupancy)
template <class L, int BM, int BN, int WGM, bool OUT32, int BKT = 64, int NP = 3>
__global__ __launch_bounds__(256, (BM * BN > 128 * 64 ? 2 : 1)) void igemm_s3_lean_kernel(S3Params p) {
    constexpr int BK = BKT;
    constexpr int CH = BK / 8, RPP = 256 / CH;
    constexpr int WGN = 4 / WGM;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM16 = WM / 16, TN16 = WN / 16;
    static_assert(WM % 16 == 0 && WN % 32 == 0 && BK % 32 == 0 && BM % RPP == 0 && BN % RPP == 0, "tile shape");
    constexpr int A_ROWS = BM / RPP, B_ROWS = BN / RPP;
    constexpr int TILE_A = BM * BK, TILE_B = BN * BK;
    constexpr int LUT_ELEMS = L::HAS_MASK ? 256 * 8 : 0;          // 256 entries x 16 bytes
    constexpr int NW = s3_wplanes_gemm<NP>;                        // weight planes
    constexpr int TILES = NP * TILE_A + NW * TILE_B > 4 * 3 * 32 * 32 ? NP * TILE_A + NW * TILE_B : 4 * 3 * 32 * 32;
    __shared__ __attribute__((aligned(16))) uint16_t smem[TILES + LUT_ELEMS];
    uint16_t* As = smem;
    uint16_t* Bs = smem + NP * TILE_A;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    int m0, n0;
    s3_tile_origin(p, BM, BN, m0, n0);
    const int srow = tid / CH, schunk = tid % CH;

    if constexpr (L::HAS_MASK) {     // entry x of the table: 8 keep bits -> 8 x 16-bit lane masks
        u32x4 e;
#pragma unroll
        for (int j = 0; j < 4; j++) e[j] = ((tid >> (2 * j)) & 1u) * 0xFFFFu | ((tid >> (2 * j + 1)) & 1u) * 0xFFFF0000u;
        *reinterpret_cast<u32x4*>(&smem[TILES + tid * 8]) = e;
    }

    // descriptors: num_records only has to cover what is addressed; offsets >= S3_OOB read as zero
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, 0x7FFFFFF0, 0x00020000);

    LeanRow<L> rows[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; i++) rows[i] = LeanRow<L>::make(p, m0 + srow + i * RPP, n0, schunk);
    uint32_t wvoff[B_ROWS];
#pragma unroll
    for (int i = 0; i < B_ROWS; i++) {
        const int n = n0 + srow + i * RPP;
        wvoff[i] = n < p.N ? (uint32_t)((n * p.Kp + schunk * 8) * 2) : S3_OOB;
    }
    // lane-invariant LDS element offsets of the staged chunks
    int a_lds[A_ROWS], b_lds[B_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; i++) a_lds[i] = (srow + i * RPP) * BK + s3_swz_m16<CH>(srow + i * RPP, schunk);
#pragma unroll
    for (int i = 0; i < B_ROWS; i++) b_lds[i] = (srow + i * RPP) * BK + s3_swz_m16<CH>(srow + i * RPP, schunk);

    f32x4_m16 acc16[TM16][TN16];
#pragma unroll
    for (int i = 0; i < TM16; i++)
#pragma unroll
        for (int j = 0; j < TN16; j++) acc16[i][j] = f32x4_m16{0.f, 0.f, 0.f, 0.f};
    f32x4_m16 acc16l[NP == 2 ? TM16 : 1][NP == 2 ? TN16 : 1];     // fp16 mode: the cross terms, scaled by 4096
#pragma unroll
    for (int i = 0; i < (NP == 2 ? TM16 : 1); i++)
#pragma unroll
        for (int j = 0; j < (NP == 2 ? TN16 : 1); j++) acc16l[i][j] = f32x4_m16{0.f, 0.f, 0.f, 0.f};

    u32x4 areg[A_ROWS][3], breg[B_ROWS][3];
    uint32_t amask[A_ROWS];
    const int n_iter_total = (p.Kp + BK - 1) / BK;
    const int it0 = (int)(((long)blockIdx.z * n_iter_total) / p.k_split);
    const int n_iter = (int)(((long)(blockIdx.z + 1) * n_iter_total) / p.k_split) - it0;
    const int a_pl = (int)(p.a_plane * 2), w_pl = (int)(p.w_plane * 2);     // plane strides in bytes (< 2 GB)

    auto g_load = [&](int it) {          // `it` is wave-uniform: the tap and every scalar offset live in SGPRs
        const typename LeanRow<L>::Tap t = LeanRow<L>::template tap<BK>(p, it);
#pragma unroll
        for (int i = 0; i < A_ROWS; i++) {
            const uint32_t vo = rows[i].voffset(p, t);
#pragma unroll
            for (int pl = 0; pl < NP; pl++)
                areg[i][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, vo, pl * a_pl, 0));
            if constexpr (L::HAS_MASK) amask[i] = p.mask[rows[i].moff + (size_t)it * (BK / 8)];
        }
        const int ws = it * BK * 2;
#pragma unroll
        for (int i = 0; i < B_ROWS; i++)
#pragma unroll
            for (int pl = 0; pl < NW; pl++)
                breg[i][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, wvoff[i], ws + pl * w_pl, 0));
    };
    auto s_store = [&]() {
#pragma unroll
        for (int i = 0; i < A_ROWS; i++) {
            u32x4 mk = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if constexpr (L::HAS_MASK) mk = *reinterpret_cast<const u32x4*>(&smem[TILES + amask[i] * 8]);
#pragma unroll
            for (int pl = 0; pl < NP; pl++) {
                u32x4 v = areg[i][pl];
                if constexpr (L::HAS_MASK) { v[0] &= mk[0]; v[1] &= mk[1]; v[2] &= mk[2]; v[3] &= mk[3]; }
                *reinterpret_cast<u32x4*>(&As[pl * TILE_A + a_lds[i]]) = v;
            }
        }
#ifdef HNET_S3_ABLATE
        if (p.tile == 91 || p.tile == 92) return;      // ablation (wrong results): no LDS stores of the weight tile
#endif
#pragma unroll
        for (int i = 0; i < B_ROWS; i++)
#pragma unroll
            for (int pl = 0; pl < NW; pl++) *reinterpret_cast<u32x4*>(&Bs[pl * TILE_B + b_lds[i]]) = breg[i][pl];
    };

    if constexpr (L::HAS_MASK) __syncthreads();      // the table is read by s_store
    g_load(it0);
    s_store();
    __syncthreads();

    const int r16 = lane & 15, g16 = lane >> 4;
    for (int it = 0; it < n_iter; it++) {
        if (it + 1 < n_iter) g_load(it0 + it + 1);
#pragma unroll
        for (int step = 0; step < BK / 32; step++) {
            bf16x8 af[TM16][3], bf[TN16][3];
#pragma unroll
            for (int i = 0; i < TM16; i++) {
                const int r = wm * WM + i * 16 + r16;
#pragma unroll
                for (int pl = 0; pl < NP; pl++)
                    af[i][pl] = *reinterpret_cast<const bf16x8*>(&As[pl * TILE_A + r * BK + s3_swz_m16<CH>(r, 4 * step + g16)]);
            }
#pragma unroll
            for (int j = 0; j < TN16; j++) {
                const int r = wn * WN + j * 16 + r16;
#ifdef HNET_S3_ABLATE
                if (p.tile == 92) {                     // ablation (wrong results): no LDS reads of the weight fragments either
#pragma unroll
                    for (int pl = 0; pl < NW; pl++) bf[j][pl] = __builtin_bit_cast(bf16x8, breg[j % B_ROWS][pl]);
                    continue;
                }
#endif
#pragma unroll
                for (int pl = 0; pl < NW; pl++)
                    bf[j][pl] = *reinterpret_cast<const bf16x8*>(&Bs[pl * TILE_B + r * BK + s3_swz_m16<CH>(r, 4 * step + g16)]);
            }
#pragma unroll
            for (int i = 0; i < TM16; i++)
#pragma unroll
                for (int j = 0; j < TN16; j++) {
                    if constexpr (NP == 2) s3_mfma16_2acc(acc16[i][j], acc16l[i][j], bf[j], af[i]);
                    else acc16[i][j] = s3_mfma16<NP>(acc16[i][j], bf[j], af[i]);
                }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the consumers of the prefetched registers behind the MFMAs (see igemm.h)
        __syncthreads();                     // every wave has read tile `it`
        if (it + 1 < n_iter) s_store();
        __syncthreads();
    }
    if constexpr (NP == 2) {
#pragma unroll
        for (int i = 0; i < TM16; i++)
#pragma unroll
            for (int j = 0; j < TN16; j++) acc16[i][j] += acc16l[i][j] * S3_F16_INV;
    }
    s3_epilogue_m16<TM16, TN16, OUT32, NP, false>(acc16, p, smem + wave * (3 * 32 * 32), m0 + wm * WM, n0 + wn * WN, lane);
    if (p.k_split > 1 && p.tickets) {            // (uniform; the epilogue above wrote raw partials and used no LDS)
        __syncthreads();                         // every wave has left the tiles: smem[0] can carry the ticket
        s3_splitk_last_arriver<OUT32, NP>(p, m0, n0, BM, BN, reinterpret_cast<uint32_t*>(smem));
    }
}

// ---------------------------------------------------------------------------------------------
// Eight-wave variant of the lean kernel (experiment, HNET_S3_TILE=12): 128 x 128 workgroup tile, 512 threads as 2 x 4 waves
// of 64 x 32 each, DOUBLE-BUFFERED LDS with ONE barrier per K-tile (tile it + 1 is stored into the other buffer after the MFMAs of tile it; every
// wave left that buffer at the barrier of tile it - 1).  Per MFMA the activation tile is staged once for 128 instead of 64 output channels
// (ds_write bytes - 33 %), the fragment reads are those of the four-wave kernel.  Same K order and MFMA sequence: same bits.  Dynamic LDS
// (2 x 64 KB of tiles + the mask table).
// ---------------------------------------------------------------------------------------------
template <class L, bool OUT32, int NP>
__global__ __launch_bounds__(512, 2) void igemm_s3_lean8_kernel(S3Params p) {
    constexpr int BM = 128, BN = 128, BK = 64, WGM = 2, NT = 512;
    constexpr int CH = BK / 8, RPP = NT / CH;
    constexpr int WGN = (NT / 64) / WGM;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM16 = WM / 16, TN16 = WN / 16;
    static_assert(NP == 2 && WM == 64 && WN == 32, "the fp16-plane mode, 64 x 32 per wave");
    constexpr int A_ROWS = BM / RPP, B_ROWS = BN / RPP;
    constexpr int TILE_A = BM * BK, TILE_B = BN * BK;
    constexpr int NW = s3_wplanes_gemm<NP>;
    constexpr int STAGE = NP * TILE_A + NW * TILE_B;               // elements per buffer
    extern __shared__ __attribute__((aligned(16))) uint16_t smem8[];
    uint16_t* const lut = smem8 + 2 * STAGE;                       // 256 entries x 16 bytes (heads only)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    int m0, n0;
    s3_tile_origin(p, BM, BN, m0, n0);
    const int srow = tid / CH, schunk = tid % CH;

    if constexpr (L::HAS_MASK) {
        if (tid < 256) {
            u32x4 e;
#pragma unroll
            for (int j = 0; j < 4; j++) e[j] = ((tid >> (2 * j)) & 1u) * 0xFFFFu | ((tid >> (2 * j + 1)) & 1u) * 0xFFFF0000u;
            *reinterpret_cast<u32x4*>(&lut[tid * 8]) = e;
        }
    }
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, 0x7FFFFFF0, 0x00020000);

    LeanRow<L> rows[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; i++) rows[i] = LeanRow<L>::make(p, m0 + srow + i * RPP, n0, schunk);
    uint32_t wvoff[B_ROWS];
#pragma unroll
    for (int i = 0; i < B_ROWS; i++) {
        const int n = n0 + srow + i * RPP;
        wvoff[i] = n < p.N ? (uint32_t)((n * p.Kp + schunk * 8) * 2) : S3_OOB;
    }
    int a_lds[A_ROWS], b_lds[B_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; i++) a_lds[i] = (srow + i * RPP) * BK + s3_swz_m16<CH>(srow + i * RPP, schunk);
#pragma unroll
    for (int i = 0; i < B_ROWS; i++) b_lds[i] = (srow + i * RPP) * BK + s3_swz_m16<CH>(srow + i * RPP, schunk);

    f32x4_m16 acc16[TM16][TN16], acc16l[TM16][TN16];
#pragma unroll
    for (int i = 0; i < TM16; i++)
#pragma unroll
        for (int j = 0; j < TN16; j++) { acc16[i][j] = f32x4_m16{0.f, 0.f, 0.f, 0.f}; acc16l[i][j] = f32x4_m16{0.f, 0.f, 0.f, 0.f}; }

    u32x4 areg[A_ROWS][3], breg[B_ROWS][3];
    uint32_t amask[A_ROWS];
    const int n_iter_total = (p.Kp + BK - 1) / BK;
    const int it0 = (int)(((long)blockIdx.z * n_iter_total) / p.k_split);
    const int n_iter = (int)(((long)(blockIdx.z + 1) * n_iter_total) / p.k_split) - it0;
    const int a_pl = (int)(p.a_plane * 2), w_pl = (int)(p.w_plane * 2);

    auto g_load = [&](int it) {
        const typename LeanRow<L>::Tap t = LeanRow<L>::template tap<BK>(p, it);
#pragma unroll
        for (int i = 0; i < A_ROWS; i++) {
            const uint32_t vo = rows[i].voffset(p, t);
#pragma unroll
            for (int pl = 0; pl < NP; pl++)
                areg[i][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, vo, pl * a_pl, 0));
            if constexpr (L::HAS_MASK) amask[i] = p.mask[rows[i].moff + (size_t)it * (BK / 8)];
        }
        const int ws = it * BK * 2;
#pragma unroll
        for (int i = 0; i < B_ROWS; i++)
#pragma unroll
            for (int pl = 0; pl < NW; pl++)
                breg[i][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, wvoff[i], ws + pl * w_pl, 0));
    };
    auto s_store = [&](int buf) {
        uint16_t* As = smem8 + buf * STAGE;
        uint16_t* Bs = As + NP * TILE_A;
#pragma unroll
        for (int i = 0; i < A_ROWS; i++) {
            u32x4 mk = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if constexpr (L::HAS_MASK) mk = *reinterpret_cast<const u32x4*>(&lut[amask[i] * 8]);
#pragma unroll
            for (int pl = 0; pl < NP; pl++) {
                u32x4 v = areg[i][pl];
                if constexpr (L::HAS_MASK) { v[0] &= mk[0]; v[1] &= mk[1]; v[2] &= mk[2]; v[3] &= mk[3]; }
                *reinterpret_cast<u32x4*>(&As[pl * TILE_A + a_lds[i]]) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; i++)
#pragma unroll
            for (int pl = 0; pl < NW; pl++) *reinterpret_cast<u32x4*>(&Bs[pl * TILE_B + b_lds[i]]) = breg[i][pl];
    };

    if constexpr (L::HAS_MASK) __syncthreads();
    g_load(it0);
    s_store(0);
    __syncthreads();

    const int r16 = lane & 15, g16 = lane >> 4;
    for (int it = 0; it < n_iter; it++) {
        if (it + 1 < n_iter) g_load(it0 + it + 1);
        const uint16_t* As = smem8 + (it & 1) * STAGE;
        const uint16_t* Bs = As + NP * TILE_A;
#pragma unroll
        for (int step = 0; step < BK / 32; step++) {
            bf16x8 af[TM16][3], bf[TN16][3];
#pragma unroll
            for (int i = 0; i < TM16; i++) {
                const int r = wm * WM + i * 16 + r16;
#pragma unroll
                for (int pl = 0; pl < NP; pl++)
                    af[i][pl] = *reinterpret_cast<const bf16x8*>(&As[pl * TILE_A + r * BK + s3_swz_m16<CH>(r, 4 * step + g16)]);
            }
#pragma unroll
            for (int j = 0; j < TN16; j++) {
                const int r = wn * WN + j * 16 + r16;
#pragma unroll
                for (int pl = 0; pl < NW; pl++)
                    bf[j][pl] = *reinterpret_cast<const bf16x8*>(&Bs[pl * TILE_B + r * BK + s3_swz_m16<CH>(r, 4 * step + g16)]);
            }
#pragma unroll
            for (int i = 0; i < TM16; i++)
#pragma unroll
                for (int j = 0; j < TN16; j++) s3_mfma16_2acc(acc16[i][j], acc16l[i][j], bf[j], af[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < n_iter) s_store((it + 1) & 1);      // the other buffer: every wave finished reading it before the previous barrier
        __syncthreads();                                 // ONE barrier per K-tile
    }
#pragma unroll
    for (int i = 0; i < TM16; i++)
#pragma unroll
        for (int j = 0; j < TN16; j++) acc16[i][j] += acc16l[i][j] * S3_F16_INV;
    s3_epilogue_m16<TM16, TN16, OUT32, NP, false>(acc16, p, smem8 + wave * (3 * 32 * 32), m0 + wm * WM, n0 + wn * WN, lane);
}

// (The LDS-DMA ring variant of rounds 1 - 3, igemm_s3_dma_kernel - global_load_lds_dwordx4 into a 3 / 4 stage ring of 32-deep K tiles on 64 x 64 tiles - lost
// every in-process measurement (profiles/r02_ab_s3_dma.log, r03_experiments_not_shipped.log item 10) and was removed in round 4; igemm_pipe.h is the
// LDS-DMA design that won: 64-deep tiles = full 128-byte rows, 144 x 128 tiles, fragments double-buffered in registers, staggered issue.)

// heads preparation: (a) featS3 = split3(feat * scale) as three planes [3][B][5120]; (b) the keep bits of both heads'
// first dropout for every local sample, one byte per 8 consecutive NHWC elements.  The hash is evaluated on the
// reference's NCHW flatten index c*20 + pix (include/hnet_rng.h); stream 0 = mean head, 2 = uncertainty head.
static __global__ __launch_bounds__(256) void heads_prep_kernel(const float* __restrict__ feat, int batch, int n_local, int s_begin,
                                                         uint32_t thr, float scale, uint64_t mc_seed, uint64_t pair_seq0,
                                                         const uint64_t* __restrict__ seq_dev,
                                                         uint16_t* __restrict__ feat16, size_t f_plane, uint8_t* __restrict__ mask, int np, int ktile_layout) {
    // (plain-bf16 mode reads plane 0 only; writing all three costs nothing measurable here)
    const size_t nfeat = (size_t)batch * 5120;
    const size_t nmask = (size_t)batch * n_local * 2 * 640;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (feat16 && i < nfeat) {         // (feat16 == nullptr: the keep bits only - heads_fc1_lat_kernel splits the features itself)
        uint16_t a, b, c;
        split_np(feat[i] * scale, np, a, b, c);
        feat16[i] = a; feat16[f_plane + i] = b;
        if (np != 2) feat16[2 * f_plane + i] = c;
    }
    // keep bits: FOUR bytes (32 draws of one row) per thread, one 32-bit store.
    // Row-major layout (ktile_layout = 0: [B][n_local][2 heads][640 bytes]): a workgroup's 1024 bytes span at most three rows (b, sample,
    // head) of 640 bytes: the row's hash prefix (four hnet_mix32 and, with a run-time n_local, an integer division) is formed by three threads
    // and shared through LDS, and all index arithmetic is 32-bit - the 64-bit i % 640, i / 640, t % n_local per thread of the first version
    // cost more than the hashing itself; with one byte per thread (round 2) the per-thread overhead was still half of the instructions.
    // K-tile-major layout (ktile_layout = 1, igemm_heads_pipe_kernel: [2 heads][80 K-tiles][M rows][8 bytes]): consecutive threads are the two
    // halves of consecutive ROWS of one K-tile (coalesced 4-byte stores), every thread forms its own row prefix.
    __shared__ uint32_t pre_row[3];
    if (!ktile_layout) {                                              // (uniform) the row-major layout: heads_mask.h
        heads_mask_block(blockIdx.x, batch, n_local, s_begin, thr, mc_seed, pair_seq0 + (seq_dev ? *seq_dev : 0ull), mask, pre_row);
        return;
    }
    if (4 * i < nmask) {                                              // nmask is a multiple of 640: the four bytes are all in or all out
        uint32_t pre, oidx;
        int chunk;                                                    // multiple of 4: the four bytes lie in one row and in one pixel's channel run
        {
            const uint32_t M = (uint32_t)batch * (uint32_t)n_local, g = (uint32_t)i, half = g & 1u, t = g >> 1;
            const uint32_t q = t / M, m = t - q * M, head = q / 80u, it = q - head * 80u;
            const uint32_t b = m / (uint32_t)n_local, sm = m - b * (uint32_t)n_local;
            pre = hnet_mask_prefix(hnet_pair_key(mc_seed, pair_seq0 + (seq_dev ? *seq_dev : 0ull) + (uint64_t)b), 2u * head, (uint32_t)s_begin + sm);
            chunk = (int)(it * 8u + half * 4u);
            oidx = (q * M + m) * 2u + half;
        }
        const int k0 = chunk * 8, pix = k0 >> 8, c0 = k0 & 255;
        // hnet_mask_keep(pre, element, thr) for the 32 elements (c0 + e) * 20 + pix: element * 0xc2b2ae35 + 0x27d4eb2f (hnet_rng.h,
        // hnet_mask_bits) advances by the constant 20 * 0xc2b2ae35 (mod 2^32) from one to the next - one quarter-rate integer multiply per
        // thread instead of one per element (the two multiplies inside hnet_mix32 remain): same bits
        uint32_t em = (uint32_t)(c0 * 20 + pix) * 0xc2b2ae35U + 0x27d4eb2fU;
        const uint32_t thr8 = thr << 8;                               // (x >> 8) >= thr  <=>  x >= thr << 8 for thr < 2^24; thr = 2^24 (p = 1) keeps nothing
        uint32_t bits = 0;
#pragma unroll
        for (int e = 0; e < 32; e++) {
            const uint32_t x = hnet_mix32(pre ^ em);
            bits |= (thr < (1u << 24) && x >= thr8) ? (1u << e) : 0u;
            em += 20U * 0xc2b2ae35U;
        }
        reinterpret_cast<uint32_t*>(mask)[oidx] = bits;
    }
}

// S3 variant of splitk_reduce_kernel: out planes <- split3(LeakyReLU(bias + sum_z partial[z]))
static __global__ __launch_bounds__(256) void splitk_reduce_s3_kernel(const float* __restrict__ partial, int k_split, int M, int N,
                                                               const float* __restrict__ bias, uint16_t* __restrict__ out16,
                                                               size_t o_plane, int np) {   // three planes also in the plain-bf16 mode (plane 0 = bf16(v) is the one read)
    const size_t total = (size_t)M * N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float s = partial[i];
    for (int z = 1; z < k_split; z++) s += partial[z * total + i];
    float v = s + bias[i % N];
    v = v > 0.0f ? v : v * 0.1f;
    uint16_t a, b, c;
    split_np(v, np, a, b, c);
    out16[i] = a; out16[o_plane + i] = b;
    if (np != 2) out16[2 * o_plane + i] = c;
}

// layout / format conversion helpers (operator-level entry points and debug read-back)
[[maybe_unused]] static __global__ void nchw_f32_to_nhwc_s3_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, size_t o_plane,
                                           int batch, int c, int hw, int np) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)batch * c * hw) return;
    const int ch = (int)(idx % c);
    const long t = idx / c;
    const int px = (int)(t % hw), b = (int)(t / hw);
    uint16_t x, y, z;
    split_np(in[((size_t)b * c + ch) * hw + px], np, x, y, z);
    out[idx] = x; out[o_plane + idx] = y; out[2 * o_plane + idx] = z;
}
[[maybe_unused]] static __global__ void nhwc_s3_to_nchw_f32_kernel(const uint16_t* __restrict__ in, size_t i_plane, float* __restrict__ out,
                                           int batch, int c, int hw, int n_planes) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)batch * c * hw) return;
    const int px = (int)(idx % hw);
    const long t = idx / hw;
    const int ch = (int)(t % c), b = (int)(t / c);
    const size_t src = ((size_t)b * hw + px) * c + ch;
    // plain-bf16 mode: planes 1, 2 are never written by the NP = 1 kernels; fp16 mode: plane 2 is not
    out[idx] = join_np(in[src], n_planes >= 2 ? in[i_plane + src] : (uint16_t)0, n_planes == 3 ? in[2 * i_plane + src] : (uint16_t)0, n_planes);
}

}  // namespace hnet
