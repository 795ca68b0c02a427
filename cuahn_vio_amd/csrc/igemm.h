// igemm.h — implicit-GEMM convolution / masked-FC kernel on fp32 MFMA for gfx950 (CDNA4).
//
// C[M][N] = act(A[M][K] * W[N][K]^T + bias[N])
//   conv : M = B*Ho*Wo output pixels, N = Cout, K = (kh, kw, ci) over an NHWC input; A is gathered on the fly
//          (reference op: nn.Conv2d + LeakyReLU(0.1), trace_pytorch_model/model_to_trace.py:7-15)
//   heads: M = B*n_mc (pair, MC sample), N = 512 (two heads x 256 hidden), K = 5120;
//          A[m][k] = feat[b][k] * keep(mask)/(1-p) generated on the fly
//          (reference op: Dropout -> Linear(5120,256) -> LeakyReLU, model_to_trace.py:222-225,229-232)
//
// Design for CDNA4:
//   * 256 threads = 4 waves (one per SIMD); v_mfma_f32_32x32x2_f32 (or 16x16x4 for Cout <= 16): exact fp32
//     fma chain, 64 FLOP/clk/SIMD.
//   * both operand tiles live in LDS as [rows][BK] fp32 (K contiguous, 16-byte quads XOR-swizzled per row).  Staging is one global float4
//     load + one ds_write_b128 per 4 K-values; fragments are read with ds_read_b128 using a K-permutation:
//     lane half h of the wave reads K = 8q+4h .. 8q+4h+3 of its row and feeds element i to MFMA step 4q+i.
//     Both operands use the same permutation so the contraction is unchanged; one LDS read serves 4 MFMAs and
//     an XOR swizzle of the 16-byte quads (ig_swz) makes the b128 reads bank-conflict free for both MFMA shapes.
//   * register-prefetch double buffering: global loads of tile t+1 are issued before the MFMAs of tile t,
//     written to the other LDS buffer afterwards; one barrier per K-tile.
//   * K order of a conv is (kh, [kw, ci]) so that, in NHWC, every (kh) row is one contiguous run of KS*CIN
//     floats: staging loads are 128-byte contiguous per 8 lanes.  Runs are cut into SEG-float segments
//     (SEG | 32); weights are pre-packed [Cout][KS][SPR*SEG] with zero padding.
//   * deterministic: fixed K order, no atomics, no split-K.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hnet_rng.h"

namespace hnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct IgemmParams {
    const float* A;       // conv: NHWC input [B][H][W][CIN]; heads: feat [B][5120] (NHWC flatten)
    const float* Wp;      // [N][Kp] packed weights (K contiguous)
    const float* bias;    // [N]
    float* out;           // [M][N]
    int M, N, Kp;
    int H, W, Ho, Wo;     // conv geometry (input H,W; output Ho,Wo)
    // heads only
    int n_local;          // MC samples evaluated by this context
    int s_begin;          // global index of the first one
    uint32_t thr;         // drop threshold (24 bit)
    float scale;          // 1/(1-p)
    uint64_t mc_seed, pair_seq0;
    const uint64_t* seq_dev;   // optional device-resident addend to pair_seq0 (lets a captured hipGraph be replayed with a new sequence number)
    // split-K (small-M launches): gridDim.z = k_split slices of the K loop, raw partial sums go to
    // partial[z][M][N]; splitk_reduce_kernel adds them in z order (deterministic) and applies bias + LeakyReLU
    int k_split;
    float* partial;
    // optional S3 output (three bf16 planes, igemm_s3.h) instead of fp32; 32x32 MFMA tiles only, no split-K
    uint16_t* out16;
    size_t o_plane;
};

constexpr int IG_BK = 32;          // K-tile (floats)
constexpr int IG_BKP = IG_BK;      // LDS row stride (floats): 128 B = 8 quads, no padding; quads are XOR-swizzled

// LDS tile layout: row r holds its 8 K-quads (16 B each) at physical quad position (q ^ ((r >> 1) & 7)).
// With 128-B rows two consecutive rows fill one 256-B bank row; the XOR makes the ds_read_b128 of every 16-lane
// group hit 16 distinct 16-B slots for both fragment shapes (32x32x2: rows l&31, quad 2q+h; 16x16x4: rows l&15,
// quad 4q+g) — the padded layout used before was conflict free only for the 32x32 shape (rocprof r01_v1:
// SQ_LDS_BANK_CONFLICT = 80 % of the LDS cycles of the 16x16 kernels).  Staging writes stay 128 contiguous bytes.
__device__ __forceinline__ int ig_swz(int row, int quad) { return (quad ^ ((row >> 1) & 7)) * 4; }

// ---------------------------------------------------------------------------------------------
// A-operand loaders
// ---------------------------------------------------------------------------------------------
template <int CIN_, int KS_, int STRIDE_, int SEG_>
struct ConvLoader {
    static constexpr int CIN = CIN_, KS = KS_, STRIDE = STRIDE_, SEG = SEG_;
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int RL = KS * CIN;                 // floats per (kh) row of the receptive field
    static constexpr int SPR = (RL + SEG - 1) / SEG;    // segments per row
    static constexpr int TOTAL_SEGS = KS * SPR;
    static constexpr int KP = TOTAL_SEGS * SEG;         // padded K
    static_assert(IG_BK % SEG == 0 && SEG % 4 == 0, "segment must divide the K tile");
    static_assert(CIN == 2 || CIN % 4 == 0, "CIN must be 2 or a multiple of 4");

    struct Row {
        int pix0;    // (b*H + iy0)*W + ix0   (may be negative; only used when in range)
        int iy0, ix0;
        bool valid;
    };

    __device__ static inline Row make_row(const IgemmParams& p, int m, int /*n0*/) {
        Row r;
        r.valid = m < p.M;
        int mm = r.valid ? m : 0;
        int hw = p.Ho * p.Wo;
        int b = mm / hw;
        int rem = mm - b * hw;
        int oy = rem / p.Wo;
        int ox = rem - oy * p.Wo;
        r.iy0 = oy * STRIDE - PAD;
        r.ix0 = ox * STRIDE - PAD;
        r.pix0 = (b * p.H + r.iy0) * p.W + r.ix0;
        return r;
    }

    // 4 consecutive K values starting at padded-K index kp (multiple of 4).
    // The global loads are UNCONDITIONAL (the address is redirected to element 0 of the input when the tap is
    // padding / out of range); `mask` says which elements are real and is applied by finalize() when the
    // registers are written to LDS, after the MFMAs of the current tile: a load under a runtime condition makes
    // hipcc branch around it and wait vmcnt(0) right there, which serialises the whole staging.
    __device__ static inline void load(const IgemmParams& p, const Row& r, int kp, f32x4& raw, uint32_t& mask) {
        const int sg = kp / SEG;
        const int within = kp % SEG;
        const int kh = sg / SPR;
        const int rr = (sg % SPR) * SEG + within;
        const int iy = r.iy0 + kh;
        const bool row_ok = r.valid && sg < TOTAL_SEGS && iy >= 0 && iy < p.H;
        if constexpr (CIN >= 4) {
            const int kw = rr / CIN, ci = rr % CIN;
            const int ix = r.ix0 + kw;
            const bool ok = row_ok && (RL % SEG == 0 || rr < RL) && ix >= 0 && ix < p.W;
            const size_t off = ok ? ((size_t)(r.pix0 + kh * p.W + kw)) * CIN + ci : 0;
            raw = *reinterpret_cast<const f32x4*>(p.A + off);
            mask = ok ? 0xFu : 0u;
        } else {  // CIN == 2: the float4 spans two pixels, each float2 has its own bounds
            const int kw = rr / 2;
            const int ix = r.ix0 + kw;
            const bool ok0 = row_ok && rr < RL && ix >= 0 && ix < p.W;
            const bool ok1 = row_ok && rr + 2 < RL && ix + 1 >= 0 && ix + 1 < p.W;
            const size_t base = ((size_t)(r.pix0 + kh * p.W + kw)) * 2;
            const float2 t0 = *reinterpret_cast<const float2*>(p.A + (ok0 ? base : 0));
            const float2 t1 = *reinterpret_cast<const float2*>(p.A + (ok1 ? base + 2 : 0));
            raw[0] = t0.x; raw[1] = t0.y; raw[2] = t1.x; raw[3] = t1.y;
            mask = (ok0 ? 0x3u : 0u) | (ok1 ? 0xCu : 0u);
        }
    }
    __device__ static inline f32x4 finalize(const IgemmParams&, const f32x4& raw, uint32_t mask) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = (mask >> i) & 1u ? raw[i] : 0.0f;
        return v;
    }
};

// MC-dropout input of the heads: A[(b, s)][k] = feat[b][k] * keep(s, k) / (1 - p)
// feat is the NHWC flatten of [4][5][256]; the mask element id is the reference's NCHW flatten index
// c*20 + pix (model_to_trace.py:253 `.view(batch_size, -1)` of [256,4,5]).
struct HeadLoader {
    static constexpr int KP = 5120;
    struct Row {
        const float* feat;
        uint32_t prefix;
        bool valid;
    };
    __device__ static inline Row make_row(const IgemmParams& p, int m, int n0) {
        Row r;
        r.valid = m < p.M;
        int mm = r.valid ? m : 0;
        int b = mm / p.n_local;
        int s = p.s_begin + (mm - b * p.n_local);
        int head = n0 >> 8;     // columns 0..255 = mean head, 256..511 = uncertainty head
        r.feat = p.A + (size_t)b * 5120;
        const uint64_t seq = p.pair_seq0 + (p.seq_dev ? *p.seq_dev : 0ull) + (uint64_t)b;
        r.prefix = hnet_mask_prefix(hnet_pair_key(p.mc_seed, seq), (uint32_t)(2 * head), (uint32_t)s);
        return r;
    }
    __device__ static inline void load(const IgemmParams& p, const Row& r, int kp, f32x4& raw, uint32_t& mask) {
        raw = *reinterpret_cast<const f32x4*>(r.feat + kp);    // unconditional (row 0 when invalid)
        const int pix = kp >> 8, c = kp & 255;
        mask = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t e = (uint32_t)((c + i) * 20 + pix);
            mask |= (r.valid && hnet_mask_keep(r.prefix, e, p.thr)) ? (1u << i) : 0u;
        }
    }
    __device__ static inline f32x4 finalize(const IgemmParams& p, const f32x4& raw, uint32_t mask) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = (mask >> i) & 1u ? raw[i] * p.scale : 0.0f;
        return v;
    }
};

// ---------------------------------------------------------------------------------------------
// S3 epilogue shared by the fp32 and the split-bf16 kernels: bias + LeakyReLU(0.1), split each value into 3 bf16
// planes, stage the wave's 32x32 tile per plane in wave-private LDS and write 16 bytes per lane (64-byte rows).
// ---------------------------------------------------------------------------------------------
typedef uint32_t ig_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint16_t ig_bf16_rn(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }   // v_cvt_pk_bf16_f32 (RNE)
template <int TM, int TN>
__device__ __forceinline__ void igemm_store_s3(f32x16 (&acc)[TM][TN], uint16_t* st, const float* bias, uint16_t* out16,
                                               size_t o_plane, int M, int N, int mw, int nw, int lane) {
    const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int nb = nw + j * 32;
        const float bv = (nb + col) < N ? bias[nb + col] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int mb = mw + i * 32;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + rbase;
                float v = acc[i][j][r] + bv;
                v = v > 0.0f ? v : v * 0.1f;
                const uint16_t a = ig_bf16_rn(v);
                const float r1 = v - __uint_as_float((uint32_t)a << 16);
                const uint16_t b = ig_bf16_rn(r1);
                const float r2 = r1 - __uint_as_float((uint32_t)b << 16);
                const uint16_t c = ig_bf16_rn(r2);
                st[(0 * 32 + row) * 32 + col] = a;
                st[(1 * 32 + row) * 32 + col] = b;
                st[(2 * 32 + row) * 32 + col] = c;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): the wave's own LDS writes have landed
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const int piece = q * 64 + lane;             // 3 planes x 32 rows x 4 chunks of 16 B
                const int pl = piece >> 7, rem = piece & 127, row = rem >> 2, ch = rem & 3;
                const int m = mb + row, n = nb + ch * 8;
                const ig_u32x4 v = *reinterpret_cast<const ig_u32x4*>(&st[(pl * 32 + row) * 32 + ch * 8]);
                if (m < M && n < N) *reinterpret_cast<ig_u32x4*>(out16 + pl * o_plane + (size_t)m * N + n) = v;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
template <int MF> struct AccT;
template <> struct AccT<32> { typedef f32x16 type; };
template <> struct AccT<16> { typedef f32x4 type; };

template <class L, int BM, int BN, int WGM, int MF>
static __global__ __launch_bounds__(256) void igemm_kernel(IgemmParams p) {
    constexpr int BK = IG_BK, BKP = IG_BKP;
    constexpr int WGN = 4 / WGM;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / MF, TN = WN / MF;
    static_assert(WM % MF == 0 && WN % MF == 0 && TM >= 1 && TN >= 1, "bad wave tile");
    constexpr int A_ROWS = BM / 32;                  // rows staged per thread (256 threads x float4 = 32 rows x 32 K)
    constexpr int B_ROWS = (BN + 31) / 32;
    typedef typename AccT<MF>::type acc_t;

    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * BKP];
    float* As = smem;
    float* Bs = smem + 2 * BM * BKP;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    const int srow = tid >> 3;          // staging row within a 32-row slab
    const int skk = (tid & 7) * 4;      // staging K offset within the tile

    typename L::Row rows[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; i++) rows[i] = L::make_row(p, m0 + srow + i * 32, n0);

    const float* wsrc[B_ROWS];
    bool wvalid[B_ROWS];
#pragma unroll
    for (int i = 0; i < B_ROWS; i++) {
        int n = srow + i * 32;
        wvalid[i] = (n < BN) && (n0 + n < p.N);
        wsrc[i] = p.Wp + (size_t)(wvalid[i] ? (n0 + n) : 0) * p.Kp + skk;
    }

    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < (MF == 32 ? 16 : 4); r++) acc[i][j][r] = 0.0f;

    f32x4 areg[A_ROWS], breg[B_ROWS];
    uint32_t amask[A_ROWS];
    bool bok[B_ROWS];
    const int n_iter_total = (p.Kp + BK - 1) / BK;
    const int it0 = (int)(((long)blockIdx.z * n_iter_total) / p.k_split);
    const int n_iter = (int)(((long)(blockIdx.z + 1) * n_iter_total) / p.k_split) - it0;

    // raw, unconditional global loads of K-tile `it` into registers (masks computed, not yet applied)
    auto g_load = [&](int it) {
        const int kp = it * BK + skk;
#pragma unroll
        for (int i = 0; i < A_ROWS; i++) L::load(p, rows[i], kp, areg[i], amask[i]);
#pragma unroll
        for (int i = 0; i < B_ROWS; i++) {
            bok[i] = wvalid[i] && kp < p.Kp;
            breg[i] = *reinterpret_cast<const f32x4*>(bok[i] ? wsrc[i] + it * BK : p.Wp);
        }
    };
    // apply the masks and write the registers to LDS buffer `buf` (first consumer of the loaded data)
    auto s_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_ROWS; i++)
            *reinterpret_cast<f32x4*>(&As[(buf * BM + srow + i * 32) * BKP + ig_swz(srow + i * 32, skk >> 2)]) = L::finalize(p, areg[i], amask[i]);
#pragma unroll
        for (int i = 0; i < B_ROWS; i++)
            if (srow + i * 32 < BN) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = bok[i] ? breg[i][e] : 0.0f;
                *reinterpret_cast<f32x4*>(&Bs[(buf * BN + srow + i * 32) * BKP + ig_swz(srow + i * 32, skk >> 2)]) = v;
            }
    };

    g_load(it0);
    s_store(0);
    __syncthreads();

    for (int it = 0; it < n_iter; it++) {
        const int buf = it & 1;
        if (it + 1 < n_iter) g_load(it0 + it + 1);

        if constexpr (MF == 32) {
            const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; i++) {
                    const int r = wm * WM + i * 32 + frow;
                    af[i] = *reinterpret_cast<const f32x4*>(&As[(buf * BM + r) * BKP + ig_swz(r, 2 * q + fh)]);
                }
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    const int r = wn * WN + j * 32 + frow;
                    bf[j] = *reinterpret_cast<const f32x4*>(&Bs[(buf * BN + r) * BKP + ig_swz(r, 2 * q + fh)]);
                }
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < TN; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
            }
        } else {
            const int frow = lane & 15, fg = lane >> 4;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                f32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; i++) {
                    const int r = wm * WM + i * 16 + frow;
                    af[i] = *reinterpret_cast<const f32x4*>(&As[(buf * BM + r) * BKP + ig_swz(r, 4 * q + fg)]);
                }
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    const int r = wn * WN + j * 16 + frow;
                    bf[j] = *reinterpret_cast<const f32x4*>(&Bs[(buf * BN + r) * BKP + ig_swz(r, 4 * q + fg)]);
                }
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < TN; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
            }
        }

        // keep the consumers of the prefetched registers (mask select + ds_write) behind the MFMAs: without this
        // fence the scheduler hoists the selects next to the loads and the s_waitcnt vmcnt lands before the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < n_iter) s_store(buf ^ 1);
        __syncthreads();
    }

    // epilogue: bias + LeakyReLU(0.1), NHWC store (lanes run along N = channels); split-K: raw partial sums
    const bool split = p.k_split > 1;
    float* const dst = split ? p.partial + (size_t)blockIdx.z * p.M * p.N : p.out;
    if constexpr (MF == 32) {
        if (p.out16 != nullptr) {
            igemm_store_s3<TM, TN>(acc, reinterpret_cast<uint16_t*>(smem) + wave * (3 * 32 * 32), p.bias, p.out16, p.o_plane,
                                   p.M, p.N, m0 + wm * WM, n0 + wn * WN, lane);
            return;
        }
        const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + wn * WN + j * 32 + col;
            const float bv = (n < p.N && !split) ? p.bias[n] : 0.0f;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                    if (m < p.M && n < p.N) {
                        float v = acc[i][j][r] + bv;
                        dst[(size_t)m * p.N + n] = (v > 0.0f || split) ? v : v * 0.1f;
                    }
                }
        }
    } else {
        const int col = lane & 15, rbase = 4 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + wn * WN + j * 16 + col;
            const float bv = (n < p.N && !split) ? p.bias[n] : 0.0f;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int m = m0 + wm * WM + i * 16 + rbase + r;
                    if (m < p.M && n < p.N) {
                        float v = acc[i][j][r] + bv;
                        dst[(size_t)m * p.N + n] = (v > 0.0f || split) ? v : v * 0.1f;
                    }
                }
        }
    }
}

// out[m][n] = LeakyReLU(bias[n] + sum_z partial[z][m][n]), z ascending (fixed order => reproducible)
static __global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int k_split, int M, int N,
                                                            const float* __restrict__ bias, float* __restrict__ out) {
    const size_t total4 = (size_t)M * N / 4;     // N is a multiple of 4 for every layer
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const size_t stride = (size_t)M * N;
    f32x4 s = *reinterpret_cast<const f32x4*>(partial + i * 4);
    for (int z = 1; z < k_split; z++) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(partial + z * stride + i * 4);
        s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
    const int n = (int)((i * 4) % N);
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
    for (int e = 0; e < 4; e++) { const float v = s[e] + b[e]; s[e] = v > 0.0f ? v : v * 0.1f; }
    *reinterpret_cast<f32x4*>(out + i * 4) = s;
}

}  // namespace hnet
