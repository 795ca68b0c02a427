// kernels.hip — the HBM-bound / small kernels of the HomographyNet forward for gfx950:
//   prep   : cat(img1, warp(img2, H)) -> AvgPool(k) -> NHWC   (fused; the concat is never materialised)
//   errmap : |warp(img2, H_total) - img1| * 255
//   block_fc_dlt : Linear(5120,8) + DLT + homography composition, one workgroup per frame pair
//   heads_fc2    : Dropout -> Linear(256,8) of both heads, per-sample outputs, ensemble + transfer
// Reference ops are cited at each kernel.
#include <cstdint>
#include <cstdlib>
#include "kernels.h"
#include "heads_mask.h"
#include "geom.h"
#include "warp_dev.h"
#include "s3_format.h"
#include "../../include/hnet.h"
#include "../../include/hnet_rng.h"

namespace hnet {

// f16(c0 - 4096 A0.lo) | f16(c1 - 4096 A0.hi) << 16 with A0 read straight from the packed fp16 register (v_fma_mixlo / mixhi_f16; igemm_s3.h has the
// same helper for the GEMM epilogues): the difference is exact in fp32, so this is the one rounding of split2h's second plane
__device__ __forceinline__ uint32_t prep_f16_residual_pk(uint32_t a0, float c0, float c1) {
    uint32_t r;
    const float ms = -S3_F16_SCALE;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a0), "s"(ms), "v"(c0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(a0), "s"(ms), "v"(c1));
    return r;
}


// WarpImg.warpSingleImage_H_Mtrx (warp.py:60-79) for one output pixel (u, v):
//   (X,Y,Z) = H (u,v,1); x = X/Z, y = Y/Z; g = x * 2/(W-1) - 1; grid_sample(bilinear, zeros,
//   align_corners=True) un-normalises i = ((g+1)/2)*(W-1) and blends the 4 neighbours; taps outside the
//   image contribute 0.  fp32 throughout like the reference; fp32 division is IEEE (hipcc default).
template <typename PIX>
__device__ inline float warp_sample(const PIX* img, const float* h, int u, int v, const float* lut) {
    float ix, iy, Z;
    warp_coords(h, u, v, ix, iy, Z);
    return warp_taps_global<PIX>(img, ix, iy, lut);
}

// ---------------------------------------------------------------------------------------------
// LDS-tiled warp: a workgroup owns a 64 x 32 tile of output pixels.  A projective map takes the tile to a convex
// quadrilateral (Z keeps its sign), so the taps of every pixel of the tile lie inside the bounding box of the four
// warped corners (+1 px for the second tap, +1 px of slack for fp32 rounding).  That box of img2 is copied ONCE into
// LDS as float — coalesced 4-pixel loads, converted through the u8 -> f32 table, positions outside the image stored
// as 0 — and the four taps of each pixel become LDS reads.  The byte gathers of the direct kernels (64 scattered
// addresses per wave instruction, 4 per pixel) kept the texture addresser busy ~90 % of the time and the kernels at
// ~1.9 TB/s; the arithmetic, and therefore every result bit, is unchanged.
// Tiles whose box does not fit the staging buffer (extreme warps), whose Z changes sign or whose coordinates are
// not finite fall back to the direct gathers; so does any single pixel whose taps fall outside the staged box.
// ---------------------------------------------------------------------------------------------
constexpr int WT_W = 64, WT_H = 32;          // output tile
constexpr int WT_CAP = 7168;                 // staged floats (28 KB): e.g. 112 x 64

// four consecutive pixels: the raw load and its conversion to float are separate steps so that several loads can be in flight
template <typename PIX> struct PxRaw;
template <> struct PxRaw<uint8_t> {
    typedef uint32_t type;
    __device__ static __forceinline__ type zero() { return 0u; }
    __device__ static __forceinline__ type load(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }
    __device__ static __forceinline__ float4 cvt(type q) {
        return make_float4(u8_to_unit((float)(q & 255u)), u8_to_unit((float)((q >> 8) & 255u)), u8_to_unit((float)((q >> 16) & 255u)),
                           u8_to_unit((float)(q >> 24)));
    }
};
template <> struct PxRaw<float> {
    typedef float4 type;
    __device__ static __forceinline__ type zero() { return make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
    __device__ static __forceinline__ type load(const float* p) { return *reinterpret_cast<const float4*>(p); }
    __device__ static __forceinline__ float4 cvt(type q) { return q; }
};

struct WarpBox { int gx0, ry0, pitch, rows; bool ok; bool zsafe; };


template <typename PIX>
__device__ inline WarpBox warp_stage_box(const PIX* __restrict__ img, const float* h, int u0, int v0, float* __restrict__ reg) {
    WarpBox bx;
    // The four warped corners: lane c of every quad evaluates corner c (ONE pass through warp_coords per wave instead of four - the corner
    // arithmetic with its range test and division was a quarter of the kernel's vector instructions), minimum / maximum across the quad,
    // the sign / range / finiteness flags by ballot; everything after that is wave-uniform and lives in SGPRs.  Same values as the
    // sequential form: min / max of non-NaN floats do not depend on the order, and any NaN clears `finite`, which discards the box.
    const int c = (int)(threadIdx.x & 3u);
    float ix, iy, Z;
    warp_coords(h, u0 + (c & 1) * (WT_W - 1), v0 + (c >> 1) * (WT_H - 1), ix, iy, Z);
    const unsigned quad0 = 0xFu;                                            // lanes 0..3 of the wave: one of each corner
    const bool pos = ((unsigned)__ballot(Z > 0.0f) & quad0) == quad0;
    const bool neg = ((unsigned)__ballot(Z < 0.0f) & quad0) == quad0;
    const bool zs = ((unsigned)__ballot(warp_z_safe(Z)) & quad0) == quad0;
    const bool finite = ((unsigned)__ballot(fabsf(ix) < 1.0e6f && fabsf(iy) < 1.0e6f) & quad0) == quad0;      // false for NaN
    const float lo_x = uniform_f(quad_min(ix)), hi_x = uniform_f(quad_max(ix));
    const float lo_y = uniform_f(quad_min(iy)), hi_y = uniform_f(quad_max(iy));
    bx.ok = (pos || neg) && finite;
    bx.zsafe = (pos || neg) && zs;        // Z is affine in (u, v) and keeps its sign over the tile: |Z| inside the tile lies between the corners' values
    // taps of pixels that survive the far-out test lie in [-1, W] x [-1, H] (+1): clip the box to that frame
    const int rx0 = max((int)floorf(bx.ok ? lo_x : 0.0f) - 1, -1), rx1 = min((int)floorf(bx.ok ? hi_x : 0.0f) + 2, IMG_W + 1);
    const int ry0 = max((int)floorf(bx.ok ? lo_y : 0.0f) - 1, -1), ry1 = min((int)floorf(bx.ok ? hi_y : 0.0f) + 2, IMG_H + 1);
    bx.gx0 = (rx0 + 4) / 4 * 4 - 4;                                     // rx0 >= -1: round down to a multiple of 4
    bx.ry0 = ry0;
    bx.pitch = rx1 >= rx0 ? (rx1 - bx.gx0 + 4) / 4 * 4 : 0;
    bx.rows = ry1 >= ry0 ? ry1 - ry0 + 1 : 0;
    if (bx.pitch * bx.rows > WT_CAP) bx.ok = false;
    if (!bx.ok) return bx;
    // eight rows per thread and pass (rows rl + 8 k of a 64-row band): the loads of a pass are all issued before the first conversion, and the usual box (32 tile
    // rows at a scale near 1 -> ~40 rows) is ONE pass = one global round trip.  (Rounds 3 - 4: four rows per pass - every box of more than 32 rows paid a second,
    // dependent round trip for its last rows; one row per pass before that.)  The kernel is bound by this chain of dependent round trips, not by its instruction
    // count (profiles/r05_experiments_not_shipped.log item 11).
    constexpr int RPP = 8;
    const int groups = bx.pitch >> 2;
    const int rl = (int)(threadIdx.x >> 5);
    for (int rb = 0; rb < bx.rows; rb += 8 * RPP) {
        for (int c = (int)(threadIdx.x & 31); c < groups; c += 32) {
            const int x = bx.gx0 + 4 * c;                                // multiple of 4: the group is all in or all out
            const bool xin = x >= 0 && x < IMG_W;
            typename PxRaw<PIX>::type raw[RPP];
#pragma unroll
            for (int k = 0; k < RPP; k++) {
                const int r = rb + rl + 8 * k, y = ry0 + r;
                raw[k] = PxRaw<PIX>::zero();
                if (r < bx.rows && xin && y >= 0 && y < IMG_H) raw[k] = PxRaw<PIX>::load(img + y * IMG_W + x);
            }
#pragma unroll
            for (int k = 0; k < RPP; k++) {
                const int r = rb + rl + 8 * k;
                if (r < bx.rows) *reinterpret_cast<float4*>(&reg[r * bx.pitch + 4 * c]) = PxRaw<PIX>::cvt(raw[k]);   // raw zero -> 0.0f
            }
        }
    }
    return bx;
}

// one pixel from the staged box (same arithmetic as warp_taps_global: zeros stand in for the skipped taps).
// Branch-free: the LDS addresses are clamped into the box and the result is selected afterwards, so that the eight
// pixels of a thread are independent instruction streams the scheduler can interleave (with a branch per pixel the
// LDS and v_rcp latencies of each pixel were exposed one after the other).  `fallback` is set when the pixel has
// taps outside the staged box (or the tile has no box); the caller then recomputes it with warp_taps_global.
template <bool ZSAFE>
__device__ __forceinline__ float warp_sample_box(const float* h, int u, int v, const WarpBox& bx, const float* reg, bool& fallback,
                                                 float& ix, float& iy) {
    float Z;
    warp_coords<ZSAFE>(h, u, v, ix, iy, Z);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const bool near = x0f >= -1.0f && x0f <= (float)IMG_W && y0f >= -1.0f && y0f <= (float)IMG_H;   // false for NaN
    const int cx = (int)(near ? x0f : 0.0f) - bx.gx0, cy = (int)(near ? y0f : 0.0f) - bx.ry0;
    const bool inbox = bx.ok && cx >= 0 && cx + 1 < bx.pitch && cy >= 0 && cy + 1 < bx.rows;
    fallback = near && !inbox;
    const float wx1 = ix - x0f, wx0 = 1.0f - wx1, wy1 = iy - y0f, wy0 = 1.0f - wy1;
    const float* t = reg + (inbox ? cy * bx.pitch + cx : 0);
    const int down = inbox ? bx.pitch : 0;
    float s = 0.0f;
    s = fmaf(t[0], wx0 * wy0, s);
    s = fmaf(t[1], wx1 * wy0, s);
    s = fmaf(t[down], wx0 * wy1, s);
    s = fmaf(t[down + 1], wx1 * wy1, s);
    return near && inbox ? s : 0.0f;
}

// The FAST sampler (round 3, the default; HNET_WARP_EXACT=1 selects the bit-faithful one above).  The contract of the path is 1e-4 px on the
// network's outputs, not bit-identical sampling positions: warp.py:60-79 normalises the coordinates to [-1, 1] and grid_sample un-normalises
// them again (two roundings that are not the identity in fp32), and X / Z, Y / Z are IEEE divisions.  Here
//   * one v_rcp_f32 + one Newton step is shared by both quotients (relative error < 1.5 ulp: 4e-5 px at 320 px) and the normalise /
//     un-normalise round trip is skipped (it moves a position by < 3e-5 px),
//   * X, Y, Z advance down the thread's column with one FMA each (the column part is formed once per thread),
//   * the tile's bounding box holds every tap (a projective map with Z of one sign takes the tile to a convex quadrilateral), so instead of
//     testing each pixel the POSITION is clamped to the box: where the box was clipped to the frame [-1, W] x [-1, H] the clamp gives the
//     zero padding exactly (position -1 or W: the only tap with a non-zero weight is a stored zero), elsewhere it never acts,
//   * weights come from v_fract_f32, the LDS address from one FMA + one conversion.
// ~30 vector instructions per pixel instead of ~85.  Used only for tiles with a staged box and Z safely away from 0 (the others take the
// exact path).  Measured position difference to the exact path: < 6e-5 px; outputs: same golden gates (tests/test_gpu_parity.py).
struct WarpFast { float bx_lo, bx_hi, by_lo, by_hi, fpitch; int ibase; };
__device__ __forceinline__ WarpFast warp_fast_setup(const WarpBox& bx) {
    WarpFast f;
    f.bx_lo = (float)max(bx.gx0, -1);
    f.bx_hi = (float)min(bx.gx0 + bx.pitch - 2, IMG_W);
    f.by_lo = (float)max(bx.ry0, -1);
    f.by_hi = (float)min(bx.ry0 + bx.rows - 2, IMG_H);
    f.fpitch = (float)bx.pitch;
    f.ibase = -(bx.ry0 * bx.pitch + bx.gx0);                           // element offset of image pixel (0, 0) in the staged box: added to the LDS base once per thread
    return f;
}
__device__ __forceinline__ float warp_sample_box_fast(float X, float Y, float Z, const WarpFast& f, int pitch, const float* reg) {
    const float r0 = __builtin_amdgcn_rcpf(Z);
    const float r1 = fmaf(fmaf(-Z, r0, 1.0f), r0, r0);
    const float ix = __builtin_amdgcn_fmed3f(X * r1, f.bx_lo, f.bx_hi);   // NaN -> one of the bounds (a zero-weight corner case, never a fault)
    const float iy = __builtin_amdgcn_fmed3f(Y * r1, f.by_lo, f.by_hi);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float wx1 = ix - x0f, wy1 = iy - y0f;
    const int e = (int)fmaf(y0f, f.fpitch, x0f);                       // exact: integers below 2^24 (the box origin is folded into `reg` by the caller)
    const float* t = reg + e;
    // round 5: the blend as three interpolations (6 instructions) instead of four weight products and four FMAs (10 with the two 1 - w): the fast sampler's
    // contract is the 1e-4 px of the network's outputs, not the reference's order of the four products (that is the exact sampler's)
    const float top = fmaf(wx1, t[1] - t[0], t[0]);
    const float bot = fmaf(wx1, t[pitch + 1] - t[pitch], t[pitch]);
    return fmaf(wy1, bot - top, top);
}

// Linear(5120, 8) of one pair + corner update + DLT + composition (model_to_trace.py:143-150, :163-168, :183-188), by one 256-thread
// workgroup: the 8 dot products in a fixed order (thread t takes elements t + 256 i, wave shuffle tree, 4 partials summed by thread 0),
// geometry in double.  Shared by block_fc_dlt_kernel (one workgroup per pair) and by the small-batch prep kernels, every workgroup of
// which recomputes its pair's homography instead of waiting for a launch of its own (FcArgs below): the same instructions in the same
// order, so the two paths agree bit for bit.  Result: hout[9] (fp32, in LDS); prior != nullptr: H = DLT(p4 + prior) (the :129-130 case).
// corner update + DLT + composition from the eight FC outputs (bias included), by one thread; geometry in double
__device__ __forceinline__ void fc_finish(const float* fc, const float* __restrict__ H_in_b, float* hout) {
    double dst[8], hb[9], hin[9], ho[9];
    for (int o = 0; o < 8; o++) dst[o] = (double)(float)(p4(o) + (double)fc[o]);
    dlt_solve(dst, hb);
    if (H_in_b) {
        for (int i = 0; i < 9; i++) hin[i] = (double)H_in_b[i];
        mat3_mul(hin, hb, ho);
    } else {
        for (int i = 0; i < 9; i++) ho[i] = hb[i];
    }
    for (int i = 0; i < 9; i++) hout[i] = (float)ho[i];
}
__device__ __forceinline__ void fc_dlt_block(const float* __restrict__ f, const float* __restrict__ wfc, const float* __restrict__ bfc,
                                             const float* __restrict__ H_in_b, float (*part)[8], float* hout) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; o++) acc[o] = 0.0f;
    // round 5: four consecutive elements per thread and step (k = 4 tid + 1024 i): 45 sixteen-byte loads, ALL in flight before the first FMA - one memory
    // round trip (rounds 3 - 4: 180 four-byte loads in two batches of 90).  Another order of the 5120 products than before; the block-tail launch and the
    // prep launches of the latency path share this function, so the two paths still agree bit for bit.
    float4 xv[5], wv[5][8];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int k = 4 * tid + 1024 * i;
        xv[i] = *reinterpret_cast<const float4*>(f + k);
#pragma unroll
        for (int o = 0; o < 8; o++) wv[i][o] = *reinterpret_cast<const float4*>(wfc + o * 5120 + k);
    }
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int o = 0; o < 8; o++) {
            acc[o] = fmaf(xv[i].x, wv[i][o].x, acc[o]);
            acc[o] = fmaf(xv[i].y, wv[i][o].y, acc[o]);
            acc[o] = fmaf(xv[i].z, wv[i][o].z, acc[o]);
            acc[o] = fmaf(xv[i].w, wv[i][o].w, acc[o]);
        }
#pragma unroll
    for (int o = 0; o < 8; o++) {
        float v = acc[o];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0) part[wave][o] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float fc[8];
        for (int o = 0; o < 8; o++) fc[o] = ((part[0][o] + part[1][o]) + (part[2][o] + part[3][o])) + bfc[o];
        fc_finish(fc, H_in_b, hout);
    }
}
// the same from the 32 partial sums per output that the previous block's tail chain left (chain_lat.h, one per item of its last layer: 8 channels x 20 pixels each):
// threads 0 .. 7 add them in item order - 1 KB instead of the 184 KB of features and weights every workgroup of the launch read
__device__ __forceinline__ void fc_part_dlt_block(const float* __restrict__ part32, const float* __restrict__ bfc, const float* __restrict__ H_in_b, float (*part)[8], float* hout) {
    const int tid = threadIdx.x;
    if (tid < 8) {
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = part32[i * 8 + tid];
        float s = v[0];
#pragma unroll
        for (int i = 1; i < 32; i++) s += v[i];
        part[0][tid] = s + bfc[tid];
    }
    __syncthreads();
    if (tid == 0) fc_finish(part[0], H_in_b, hout);
}
__device__ __forceinline__ void prior_dlt_block(const float* __restrict__ prior_b, float* hout) {     // dlt_kernel with add_corners
    if (threadIdx.x == 0) {
        double d[8], hb[9];
        for (int k = 0; k < 8; k++) d[k] = (double)(float)(p4(k) + (double)prior_b[k]);
        dlt_solve(d, hb);
        for (int k = 0; k < 9; k++) hout[k] = (float)hb[k];
    }
}

// prep with warp, LDS-tiled: AvgPool_K(cat(img1, warp(img2, H))) -> NHWC [B][224/K][320/K][2].
// Lane = column of the tile, wave w = rows 8w .. 8w+7, eight pixels of one column per thread: the taps of a wave are
// (nearly) consecutive LDS words, the K = 1 stores are 512 contiguous bytes per wave and row.  Pooling: the rows of a
// window are summed in the thread, its columns across lanes (xor 1, 2, 4); lane % K == 0 stores.
// OUTS3 (K = 1 only): the block-4 input is written as bf16 planes with a zero border, [plane][B][B4_HP][B4_WP] dwords
// (lo = img1, hi = warped img2 of one pixel), the layout the fused block-4 kernel stages by LDS-DMA (kernels.h B4_*)
// FC (small batches only): H is not read from memory but recomputed by every workgroup from the previous block's trunk output (fc_dlt_block,
// FcArgs) - the block-tail launch disappears from the dependent chain of a batch-1 forward; the first workgroup of a pair stores it.
template <typename PIX, int K, bool OUTS3 = false, bool EXACT = true, bool FC = false>
__global__ __launch_bounds__(256) void prep_warp_tiled_kernel(const PIX* __restrict__ img1, const PIX* __restrict__ img2,
                                                              const float* __restrict__ H, float* __restrict__ out,
                                                              uint32_t* __restrict__ out_s3 = nullptr, size_t s3_plane = 0, int n_planes = 3,
                                                              FcArgs fc = FcArgs{}) {
    __shared__ __attribute__((aligned(16))) float reg[WT_CAP];
    constexpr int TX = IMG_W / WT_W, TY = IMG_H / WT_H, HO = IMG_H / K, WO = IMG_W / K;
    unsigned n_main = gridDim.x;
    if constexpr (FC) {
        if (fc.mask) {      // the surplus workgroups of the grid: keep bits of the heads (FcArgs::mask)
            n_main -= (unsigned)fc.mask_blocks;
            if (blockIdx.x >= n_main) {
                heads_mask_block(blockIdx.x - n_main, (int)(n_main / (TX * TY)), fc.n_local, fc.s_begin, fc.thr, fc.mc_seed, fc.pair_seq0 + (fc.seq_dev ? *fc.seq_dev : 0ull),
                                 fc.mask, reinterpret_cast<uint32_t*>(reg));
                return;
            }
        }
    }
    // XCD-aware tile order: consecutive workgroup ids run on different XCDs (private L2s) and the staged boxes of neighbouring tiles
    // overlap: with tile = workgroup id the warped image was fetched 2.4 times (FETCH_SIZE x 2: 89 MB for 37 MB of images)
    int bid = ((n_main & 7) == 0) ? (int)((blockIdx.x & 7) * (n_main >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int tx = bid % TX; bid /= TX;
    const int ty = bid % TY;
    const int b = bid / TY;
    const int u0 = tx * WT_W, v0 = ty * WT_H;
    const PIX* i1 = img1 + (size_t)b * NPIX;
    const PIX* i2 = img2 + (size_t)b * NPIX;
    const int lane = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * 8;
    const int u = u0 + lane;
    // (round 6: img1's pixels are requested BEFORE the homography is formed - they do not depend on it, and on the latency path the FC / DLT in front of the warp is
    // a memory round trip + a serial double-precision tail of its own)
    PIX araw[8];
    const PIX* p1 = i1 + (v0 + r0) * IMG_W + u;                          // one address, the rows as instruction offsets
#pragma unroll
    for (int i = 0; i < 8; i++) araw[i] = p1[i * IMG_W];
    float h[9];
    if constexpr (FC) {
        __shared__ float fc_part[4][8];
        __shared__ float fc_h[9];
        if (fc.fc_part) fc_part_dlt_block(fc.fc_part + (size_t)b * 256, fc.bfc, fc.H_in ? fc.H_in + b * 9 : nullptr, fc_part, fc_h);
        else if (fc.feat) fc_dlt_block(fc.feat + (size_t)b * 5120, fc.wfc, fc.bfc, fc.H_in ? fc.H_in + b * 9 : nullptr, fc_part, fc_h);
        else prior_dlt_block(fc.prior + b * 8, fc_h);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 9; i++) h[i] = fc_h[i];
        if (tx == 0 && ty == 0 && threadIdx.x < 9) fc.H_out[b * 9 + threadIdx.x] = h[threadIdx.x];
    } else {
#pragma unroll
        for (int i = 0; i < 9; i++) h[i] = H[b * 9 + i];
    }
    // round 3: the kernel was bound by its own dependent latencies (img1 tile -> LDS, then H -> box -> img2 box -> LDS, barrier: two global
    // round trips in sequence at four workgroups per CU).  A thread's eight img1 pixels (one column of the tile: 64 consecutive bytes per
    // wave and row) are now loaded straight into registers BEFORE the box is staged and are consumed after the barrier: one round trip,
    // no LDS tile for img1 (28 instead of 37 KB per workgroup), no u8 -> f32 table.
    const WarpBox bx = warp_stage_box<PIX>(i2, h, u0, v0, reg);
    __syncthreads();
    float a[8], w[8], fx[8], fy[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = PixRead<PIX>::cvt(araw[i]);
    uint32_t fb = 0;
    if (!EXACT && bx.zsafe && bx.ok && bx.rows >= 2 && bx.pitch >= 2) {   // workgroup-uniform: the fast sampler (see warp_sample_box_fast)
        const WarpFast wf = warp_fast_setup(bx);
        const float fu = (float)u;
        const float Xc = fmaf(h[0], fu, h[2]), Yc = fmaf(h[3], fu, h[5]), Zc = fmaf(h[6], fu, h[8]);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float fv = (float)(v0 + r0 + i);
            w[i] = warp_sample_box_fast(fmaf(h[1], fv, Xc), fmaf(h[4], fv, Yc), fmaf(h[7], fv, Zc), wf, bx.pitch, reg + wf.ibase);
        }
    } else if (bx.zsafe) {                                              // workgroup-uniform: the shared-reciprocal division without its range test
#pragma unroll
        for (int i = 0; i < 8; i++) {
            bool f;
            w[i] = warp_sample_box<true>(h, u, v0 + r0 + i, bx, reg, f, fx[i], fy[i]);
            fb |= (uint32_t)f << i;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            bool f;
            w[i] = warp_sample_box<false>(h, u, v0 + r0 + i, bx, reg, f, fx[i], fy[i]);
            fb |= (uint32_t)f << i;
        }
    }
    if (__builtin_expect(__any(fb != 0), 0)) {                          // rare: taps outside the staged box -> direct gathers
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (fb & (1u << i)) w[i] = warp_taps_global<PIX, false>(i2, fx[i], fy[i], nullptr);
    }
    if constexpr (K == 1 && OUTS3) {
        if (n_planes == 2) {                                            // fp16 planes (the default mode): both channels of a pixel split as ONE packed pair
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            typedef float f2 __attribute__((ext_vector_type(2)));
            uint32_t* o0 = out_s3 + ((size_t)b * B4_HP + (v0 + r0 + B4_PADY)) * B4_WP + u + B4_PADX;
#pragma unroll
            for (int i = 0; i < 8; i++) {                               // = split2h(a), split2h(w) of s3_format.h: A0 = f16(v), A1 = f16((v - A0) 4096)
                const f2 v = {a[i], w[i]};
                const h2 hi = {(_Float16)v[0], (_Float16)v[1]};
                const uint32_t hi_u = __builtin_bit_cast(uint32_t, hi);
                const f2 c = v * S3_F16_SCALE;                          // (exact) - then A1 = f16(c - 4096 A0) in two mixed-precision FMAs (round 5: 4 instructions, 6 before)
                o0[(size_t)i * B4_WP] = hi_u;
                o0[s3_plane + (size_t)i * B4_WP] = prep_f16_residual_pk(hi_u, c[0], c[1]);
            }
        } else
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint16_t a0, a1, a2, w0, w1, w2;
            split_np(a[i], n_planes, a0, a1, a2);
            split_np(w[i], n_planes, w0, w1, w2);
            const size_t idx = ((size_t)b * B4_HP + (v0 + r0 + i + B4_PADY)) * B4_WP + u + B4_PADX;
            out_s3[idx] = (uint32_t)a0 | ((uint32_t)w0 << 16);
            if (n_planes >= 2) out_s3[s3_plane + idx] = (uint32_t)a1 | ((uint32_t)w1 << 16);
            if (n_planes == 3) out_s3[2 * s3_plane + idx] = (uint32_t)a2 | ((uint32_t)w2 << 16);
        }
    } else if constexpr (K == 1) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            *reinterpret_cast<float2*>(out + ((size_t)b * NPIX + (size_t)(v0 + r0 + i) * IMG_W + u) * 2) = make_float2(a[i], w[i]);
    } else {
        constexpr int NW = 8 / K;                                       // window rows this thread covers
        float s1[NW], s2[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) {
            s1[j] = 0.0f; s2[j] = 0.0f;
#pragma unroll
            for (int i = 0; i < K; i++) { s1[j] += a[j * K + i]; s2[j] += w[j * K + i]; }
        }
#pragma unroll
        for (int m = 1; m < K; m <<= 1)
#pragma unroll
            for (int j = 0; j < NW; j++) { s1[j] += __shfl_xor(s1[j], m); s2[j] += __shfl_xor(s2[j], m); }
        if ((lane & (K - 1)) == 0) {
            constexpr float inv = 1.0f / (float)(K * K);                // power of two: exact
#pragma unroll
            for (int j = 0; j < NW; j++)
                *reinterpret_cast<float2*>(out + (((size_t)b * HO + (v0 + r0) / K + j) * WO + u / K) * 2) = make_float2(s1[j] * inv, s2[j] * inv);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// prep: block input = AvgPool_k(cat(img1, warp(img2, H)))   (model_to_trace.py:138-139,153-157,171-175,261-263)
// K/2 lanes cooperate on one output pixel: lane j sums rows 2j, 2j+1 of the k x k window, then a shuffle tree.
// Output NHWC [B][224/k][320/k][2] — channel 0 = img1, channel 1 = (warped) img2.
// ---------------------------------------------------------------------------------------------
template <typename PIX, int K, bool WARP>
__global__ __launch_bounds__(256) void prep_kernel(const PIX* __restrict__ img1, const PIX* __restrict__ img2,
                                                   const float* __restrict__ H, float* __restrict__ out, int batch) {
    __shared__ float lut[256];
    if (PixRead<PIX>::kNeedLut) fill_lut(lut);
    constexpr int HO = IMG_H / K, WO = IMG_W / K;
    constexpr int RPL = K >= 2 ? 2 : 1;          // window rows per lane: 2K independent samples in flight per lane
    constexpr int G = K / RPL;                   // lanes per output pixel
    // 32-bit index arithmetic (the launcher refuses batches whose pixel count does not fit): 64-bit divisions by HO * WO are ~20 instructions
    const uint32_t total = (uint32_t)batch * HO * WO * G;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = idx < total;
    const uint32_t ii = active ? idx : 0u;
    const int j = (int)(ii % G);
    const uint32_t opix = ii / G;
    const int b = (int)(opix / (uint32_t)(HO * WO));
    const int rem = (int)(opix - (uint32_t)b * (HO * WO));
    const int oy = rem / WO, ox = rem - oy * WO;
    const PIX* i1 = img1 + (size_t)b * NPIX;
    const PIX* i2 = img2 + (size_t)b * NPIX;
    float h[9];
    if (WARP) {
#pragma unroll
        for (int i = 0; i < 9; i++) h[i] = H[b * 9 + i];
    }
    float s1 = 0.0f, s2 = 0.0f;
    // (the 8-byte form needs 8-byte aligned images - each pair is 71 680 bytes further, a multiple of 8 - which hnet_infer_batch*_device does not promise:
    // an odd buffer takes the byte path below, like the other vectorised prep paths; kernel-uniform)
    const bool vec8 = !WARP && K == 8 && sizeof(PIX) == 1 && ((((uintptr_t)img1 | (uintptr_t)img2) & 7) == 0);
    if (vec8) {
        // block 1 of the full model (no warp, 8 x 8 pool, u8 images): the eight pixels of a window row are ONE aligned 8-byte load per image (round 4; as 32 byte
        // loads + 32 table look-ups per lane the kernel ran at half the memory rate with 54 % of its LDS cycles bank conflicts).  Same values (u8_to_unit is
        // bit-identical to the table), same order of the sums.
#pragma unroll
        for (int rr = 0; rr < RPL; rr++) {
            const int v = oy * K + j * RPL + rr;
            const uint2 q1 = *reinterpret_cast<const uint2*>(i1 + v * IMG_W + ox * K), q2 = *reinterpret_cast<const uint2*>(i2 + v * IMG_W + ox * K);
#pragma unroll
            for (int a = 0; a < K; a++) {
                const uint32_t w1 = a < 4 ? q1.x : q1.y, w2 = a < 4 ? q2.x : q2.y;
                s1 += u8_to_unit((float)((w1 >> (8 * (a & 3))) & 255u));
                s2 += u8_to_unit((float)((w2 >> (8 * (a & 3))) & 255u));
            }
        }
    } else {
#pragma unroll
    for (int rr = 0; rr < RPL; rr++) {
        const int v = oy * K + j * RPL + rr;
#pragma unroll
        for (int a = 0; a < K; a++) {
            const int u = ox * K + a;
            s1 += PixRead<PIX>::get(i1, v * IMG_W + u, lut);
            s2 += WARP ? warp_sample<PIX>(i2, h, u, v, lut) : PixRead<PIX>::get(i2, v * IMG_W + u, lut);
        }
    }
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) {
        s1 += __shfl_xor(s1, m);
        s2 += __shfl_xor(s2, m);
    }
    if (active && j == 0) {
        constexpr float inv = 1.0f / (float)(K * K);   // power of two: exact
        float2 o = make_float2(s1 * inv, s2 * inv);
        *reinterpret_cast<float2*>(out + (size_t)opix * 2) = o;
    }
}

// K = 1 (block 4: full resolution, no pooling): four consecutive pixels per thread so that sixteen independent
// byte gathers are in flight per lane (one pixel per thread left the kernel waiting on memory 80 % of the time);
// img1 is read as one 4-pixel vector, the output is two float4 stores.
template <typename PIX>
__global__ __launch_bounds__(256) void prep_k1_kernel(const PIX* __restrict__ img1, const PIX* __restrict__ img2,
                                                      const float* __restrict__ H, float* __restrict__ out, int batch) {
    __shared__ float lut[256];
    if (PixRead<PIX>::kNeedLut) fill_lut(lut);
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // group of 4 pixels
    if (idx >= (long)batch * (NPIX / 4)) return;
    const int b = (int)(idx / (NPIX / 4));
    const int pix0 = (int)(idx - (long)b * (NPIX / 4)) * 4;
    const int v = pix0 / IMG_W, u0 = pix0 - v * IMG_W;                   // IMG_W % 4 == 0: the 4 pixels share a row
    const PIX* i1 = img1 + (size_t)b * NPIX;
    const PIX* i2 = img2 + (size_t)b * NPIX;
    float h[9];
#pragma unroll
    for (int i = 0; i < 9; i++) h[i] = H[b * 9 + i];
    float a[4], w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = warp_sample<PIX>(i2, h, u0 + i, v, lut);
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = PixRead<PIX>::get(i1, pix0 + i, lut);
    float4* o = reinterpret_cast<float4*>(out + ((size_t)b * NPIX + pix0) * 2);
    o[0] = make_float4(a[0], w[0], a[1], w[1]);
    o[1] = make_float4(a[2], w[2], a[3], w[3]);
}

// fp32 NHWC [B][224][320][2] <-> the padded bf16-plane layout of the block-4 input (operator entry points, debug read-back and the
// fallback when the images are not 16-byte aligned)
__global__ void f32_nhwc_to_s3pad_kernel(const float* __restrict__ x, uint32_t* __restrict__ out, size_t s3_plane, int batch, int n_planes) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)batch * NPIX) return;
    const int b = (int)(i / NPIX), pix = (int)(i - (long)b * NPIX), v = pix / IMG_W, u = pix - v * IMG_W;
    const float2 f = *reinterpret_cast<const float2*>(x + (size_t)i * 2);
    uint16_t a0, a1, a2, w0, w1, w2;
    split_np(f.x, n_planes, a0, a1, a2);
    split_np(f.y, n_planes, w0, w1, w2);
    const size_t idx = ((size_t)b * B4_HP + v + B4_PADY) * B4_WP + u + B4_PADX;
    out[idx] = (uint32_t)a0 | ((uint32_t)w0 << 16);
    if (n_planes >= 2) out[s3_plane + idx] = (uint32_t)a1 | ((uint32_t)w1 << 16);
    if (n_planes == 3) out[2 * s3_plane + idx] = (uint32_t)a2 | ((uint32_t)w2 << 16);
}
__global__ void s3pad_to_f32_nhwc_kernel(const uint32_t* __restrict__ in, size_t s3_plane, float* __restrict__ x, int batch, int n_planes) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)batch * NPIX) return;
    const int b = (int)(i / NPIX), pix = (int)(i - (long)b * NPIX), v = pix / IMG_W, u = pix - v * IMG_W;
    const size_t idx = ((size_t)b * B4_HP + v + B4_PADY) * B4_WP + u + B4_PADX;
    float lo = 0.f, hi = 0.f;
    if (n_planes == 2) {                                  // fp16 planes: A0 + A1 / 4096
        const uint32_t d0 = in[idx], d1 = in[s3_plane + idx];
        lo = join2h((uint16_t)(d0 & 0xffffu), (uint16_t)(d1 & 0xffffu));
        hi = join2h((uint16_t)(d0 >> 16), (uint16_t)(d1 >> 16));
    } else
    for (int pl = n_planes - 1; pl >= 0; pl--) {         // smallest plane first: the sum of the three planes is the exact fp32 value
        const uint32_t d = in[pl * s3_plane + idx];
        lo += bf16_to_f32((uint16_t)(d & 0xffffu));
        hi += bf16_to_f32((uint16_t)(d >> 16));
    }
    *reinterpret_cast<float2*>(x + (size_t)i * 2) = make_float2(lo, hi);
}
hipError_t launch_f32_nhwc_to_s3pad(const float* x, uint32_t* out, size_t s3_plane, int batch, int n_planes, hipStream_t s) {
    const long n = (long)batch * NPIX;
    hipLaunchKernelGGL(f32_nhwc_to_s3pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, out, s3_plane, batch, n_planes);
    return hipGetLastError();
}
hipError_t launch_s3pad_to_f32_nhwc(const uint32_t* in, size_t s3_plane, float* x, int batch, int n_planes, hipStream_t s) {
    const long n = (long)batch * NPIX;
    hipLaunchKernelGGL(s3pad_to_f32_nhwc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, s3_plane, x, batch, n_planes);
    return hipGetLastError();
}

template <typename PIX>
static hipError_t prep_dispatch(const PIX* i1, const PIX* i2, const float* H, int k, float* out, int batch, hipStream_t s,
                                uint32_t* out_s3, size_t s3_plane, int n_planes, bool exact) {
    if (out_s3) {     // block-4 input as padded bf16 planes (k = 1, with warp): the tiled kernel writes them directly
        if (k != 1 || !H) return hipErrorInvalidValue;
        if ((((uintptr_t)i1 | (uintptr_t)i2) & 15) == 0) {
            const unsigned blocks = (unsigned)batch * (IMG_W / WT_W) * (IMG_H / WT_H);
            if (exact) hipLaunchKernelGGL((prep_warp_tiled_kernel<PIX, 1, true, true>), dim3(blocks), dim3(256), 0, s, i1, i2, H, out, out_s3, s3_plane, n_planes);
            else hipLaunchKernelGGL((prep_warp_tiled_kernel<PIX, 1, true, false>), dim3(blocks), dim3(256), 0, s, i1, i2, H, out, out_s3, s3_plane, n_planes);
            return hipGetLastError();
        }
        // unaligned images: direct-gather kernel into the fp32 buffer, then convert
        const long groups = (long)batch * (NPIX / 4);
        hipLaunchKernelGGL(prep_k1_kernel<PIX>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, i1, i2, H, out, batch);
        return launch_f32_nhwc_to_s3pad(out, out_s3, s3_plane, batch, n_planes, s);
    }
    // measured at batch 256 (ms, tiled vs direct, exact sampler): K=1 0.085 / 0.094, K=2 0.076 / 0.080, K=4 0.068 / 0.062 -> the bit-faithful sampler runs
    // tiled for K <= 2 only; the fast sampler exists in the tiled kernel only and serves every K
    if (H && (k <= 2 || !exact) && (((uintptr_t)i1 | (uintptr_t)i2) & 15) == 0) {   // 4-pixel groups: u8 4 B, f32 16 B loads
        const unsigned blocks = (unsigned)batch * (IMG_W / WT_W) * (IMG_H / WT_H);
#define HNET_TILED(KK)                                                                                                                 \
    if (exact) hipLaunchKernelGGL((prep_warp_tiled_kernel<PIX, KK, false, true>), dim3(blocks), dim3(256), 0, s, i1, i2, H, out);       \
    else hipLaunchKernelGGL((prep_warp_tiled_kernel<PIX, KK, false, false>), dim3(blocks), dim3(256), 0, s, i1, i2, H, out);
        switch (k) {
            case 1: HNET_TILED(1); break;
            case 2: HNET_TILED(2); break;
            case 4: HNET_TILED(4); break;
            case 8: HNET_TILED(8); break;
            default: return hipErrorInvalidValue;
        }
#undef HNET_TILED
        return hipGetLastError();
    }
    if (k == 1 && H) {
        const long groups = (long)batch * (NPIX / 4);
        hipLaunchKernelGGL(prep_k1_kernel<PIX>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, i1, i2, H, out, batch);
        return hipGetLastError();
    }
    const long total = (long)batch * NPIX / k / (k >= 2 ? 2 : 1);   // (224/k)*(320/k) outputs x k/2 lanes each
    if (total + 256 >= (1l << 32)) return hipErrorInvalidValue;     // prep_kernel indexes in 32 bits (batch < ~59 000 pairs)
    const int blocks = (int)((total + 255) / 256);
#define HNET_PREP(KK)                                                                                     \
    if (H) hipLaunchKernelGGL((prep_kernel<PIX, KK, true>), dim3(blocks), dim3(256), 0, s, i1, i2, H, out, batch); \
    else hipLaunchKernelGGL((prep_kernel<PIX, KK, false>), dim3(blocks), dim3(256), 0, s, i1, i2, H, out, batch);
    switch (k) {
        case 1: HNET_PREP(1); break;
        case 2: HNET_PREP(2); break;
        case 4: HNET_PREP(4); break;
        case 8: HNET_PREP(8); break;
        default: return hipErrorInvalidValue;
    }
#undef HNET_PREP
    return hipGetLastError();
}

hipError_t launch_prep(const void* img1, const void* img2, int pix_fmt, const float* H, int k, float* out,
                       int batch, hipStream_t s, uint32_t* out_s3, size_t s3_plane, int n_planes, bool exact) {
    if (pix_fmt == HNET_PIX_U8) return prep_dispatch<uint8_t>((const uint8_t*)img1, (const uint8_t*)img2, H, k, out, batch, s, out_s3, s3_plane, n_planes, exact);
    return prep_dispatch<float>((const float*)img1, (const float*)img2, H, k, out, batch, s, out_s3, s3_plane, n_planes, exact);
}

// small-batch form: the homography of each pair is recomputed by every workgroup of the tiled kernel from the previous block's trunk
// output (or from the prior), see FcArgs / fc_dlt_block.  Tiled kernel only: needs 16-byte aligned images and K in {1 (padded planes), 2, 4}.
bool prep_fc_supported(const void* img1, const void* img2, int k, bool has_out_s3) {
    return ((((uintptr_t)img1 | (uintptr_t)img2) & 15) == 0) && ((k == 1 && has_out_s3) || k == 2 || k == 4);
}
template <typename PIX>
static hipError_t prep_fc_dispatch(const PIX* i1, const PIX* i2, const FcArgs& fc, int k, float* out, int batch, hipStream_t s, uint32_t* out_s3,
                                   size_t s3_plane, int n_planes, bool exact) {
    const unsigned blocks = (unsigned)batch * (IMG_W / WT_W) * (IMG_H / WT_H) + (fc.mask ? (unsigned)fc.mask_blocks : 0u);
#define HNET_TILED_FC(KK, S3)                                                                                                                        \
    if (exact) hipLaunchKernelGGL((prep_warp_tiled_kernel<PIX, KK, S3, true, true>), dim3(blocks), dim3(256), 0, s, i1, i2, (const float*)nullptr, out, out_s3, s3_plane, n_planes, fc); \
    else hipLaunchKernelGGL((prep_warp_tiled_kernel<PIX, KK, S3, false, true>), dim3(blocks), dim3(256), 0, s, i1, i2, (const float*)nullptr, out, out_s3, s3_plane, n_planes, fc);
    switch (k) {
        case 1: HNET_TILED_FC(1, true); break;
        case 2: HNET_TILED_FC(2, false); break;
        case 4: HNET_TILED_FC(4, false); break;
        default: return hipErrorInvalidValue;
    }
#undef HNET_TILED_FC
    return hipGetLastError();
}
hipError_t launch_prep_fc(const void* img1, const void* img2, int pix_fmt, const FcArgs& fc, int k, float* out, int batch, hipStream_t s,
                          uint32_t* out_s3, size_t s3_plane, int n_planes, bool exact) {
    if (!prep_fc_supported(img1, img2, k, out_s3 != nullptr) || !fc.H_out || (!fc.feat && !fc.prior)) return hipErrorInvalidValue;
    if (pix_fmt == HNET_PIX_U8) return prep_fc_dispatch<uint8_t>((const uint8_t*)img1, (const uint8_t*)img2, fc, k, out, batch, s, out_s3, s3_plane, n_planes, exact);
    return prep_fc_dispatch<float>((const float*)img1, (const float*)img2, fc, k, out, batch, s, out_s3, s3_plane, n_planes, exact);
}

// ---------------------------------------------------------------------------------------------
// error map: |warp(img2, H_total) - img1| * 255   (model_to_trace.py:324-327; u8 clamp HomographyNet.cpp:201)
// ---------------------------------------------------------------------------------------------
template <typename PIX>
__global__ __launch_bounds__(256) void errmap_kernel(const PIX* __restrict__ img1, const PIX* __restrict__ img2,
                                                     const float* __restrict__ H, float* __restrict__ out,
                                                     uint8_t* __restrict__ out_u8, int batch) {
    __shared__ float lut[256];
    if (PixRead<PIX>::kNeedLut) fill_lut(lut);
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)batch * NPIX) return;
    const int b = (int)(idx / NPIX), pix = (int)(idx - (long)b * NPIX);
    const int v = pix / IMG_W, u = pix - v * IMG_W;
    float h[9];
#pragma unroll
    for (int i = 0; i < 9; i++) h[i] = H[b * 9 + i];
    const float w = warp_sample<PIX>(img2 + (size_t)b * NPIX, h, u, v, lut);
    const float e = fabsf(w - PixRead<PIX>::get(img1 + (size_t)b * NPIX, pix, lut)) * 255.0f;
    if (out) out[idx] = e;
    if (out_u8) out_u8[idx] = (uint8_t)fminf(fmaxf(e, 0.0f), 255.0f);   // clamp(0,255).to(kU8) truncates
}

hipError_t launch_errmap(const void* img1, const void* img2, int pix_fmt, const float* H, float* out,
                         uint8_t* out_u8, int batch, hipStream_t s) {
    const int blocks = (int)(((long)batch * NPIX + 255) / 256);
    if (pix_fmt == HNET_PIX_U8)
        hipLaunchKernelGGL(errmap_kernel<uint8_t>, dim3(blocks), dim3(256), 0, s, (const uint8_t*)img1, (const uint8_t*)img2, H, out, out_u8, batch);
    else
        hipLaunchKernelGGL(errmap_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)img1, (const float*)img2, H, out, out_u8, batch);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// undistort + resize (CamBase::undistort_and_resize_img = cv::remap(INTER_LINEAR, BORDER_CONSTANT 0), CamBase.h:182-186):
// out(v, u) = bilinear(raw, map_x(v, u), map_y(v, u)).  Sample positions are quantised to 1/32 px as cv::remap does
// (INTER_BITS = 5, round half to even), the blend is exact integer arithmetic: weights (32-ax)(32-ay) ... sum 1024,
// result (sum + 512) >> 10.  71 680 output pixels per frame: a latency-bound gather, one pixel per thread.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void undistort_kernel(const uint8_t* __restrict__ raw, int rows, int cols, int stride,
                                                        const float* __restrict__ map_x, const float* __restrict__ map_y,
                                                        uint8_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NPIX) return;
    const float fx = map_x[i] * 32.0f, fy = map_y[i] * 32.0f;
    // saturate like cv::saturate_cast<short> of the integer part does, and keep NaN / huge positions out of the image
    const bool sane = fabsf(fx) < 1.0e9f && fabsf(fy) < 1.0e9f;
    const int sx = sane ? __float2int_rn(fx) : -(1 << 20), sy = sane ? __float2int_rn(fy) : -(1 << 20);
    const int x0 = sx >> 5, y0 = sy >> 5, ax = sx & 31, ay = sy & 31;
    auto tap = [&](int y, int x) -> int { return ((unsigned)y < (unsigned)rows && (unsigned)x < (unsigned)cols) ? (int)raw[(size_t)y * stride + x] : 0; };
    const int v = tap(y0, x0) * (32 - ax) * (32 - ay) + tap(y0, x0 + 1) * ax * (32 - ay) + tap(y0 + 1, x0) * (32 - ax) * ay +
                  tap(y0 + 1, x0 + 1) * ax * ay;
    out[i] = (uint8_t)((v + 512) >> 10);
}

hipError_t launch_undistort(const uint8_t* raw, int rows, int cols, int stride, const float* map_x, const float* map_y, uint8_t* out,
                            hipStream_t s) {
    hipLaunchKernelGGL(undistort_kernel, dim3((NPIX + 255) / 256), dim3(256), 0, s, raw, rows, cols, stride, map_x, map_y, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void warp_f32_kernel(const float* __restrict__ img, const float* __restrict__ H,
                                                       float* __restrict__ out, int batch) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)batch * NPIX) return;
    const int b = (int)(idx / NPIX), pix = (int)(idx - (long)b * NPIX);
    const int v = pix / IMG_W, u = pix - v * IMG_W;
    float h[9];
#pragma unroll
    for (int i = 0; i < 9; i++) h[i] = H[b * 9 + i];
    out[idx] = warp_sample<float>(img + (size_t)b * NPIX, h, u, v, nullptr);
}

hipError_t launch_warp_f32(const float* img, const float* H, float* out, int batch, hipStream_t s) {
    const int blocks = (int)(((long)batch * NPIX + 255) / 256);
    hipLaunchKernelGGL(warp_f32_kernel, dim3(blocks), dim3(256), 0, s, img, H, out, batch);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// DLT kernels (model_to_trace.py:42-61, :129-130)
// ---------------------------------------------------------------------------------------------
__global__ void dlt_kernel(const float* __restrict__ dst_or_prior, float* __restrict__ H, int n, int add_corners) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d[8], h[9];
    for (int k = 0; k < 8; k++) {
        const float x = dst_or_prior[i * 8 + k];
        d[k] = add_corners ? (double)(float)(p4(k) + (double)x) : (double)x;   // p4 + prior is an fp32 tensor op
    }
    dlt_solve(d, h);
    for (int k = 0; k < 9; k++) H[i * 9 + k] = (float)h[k];
}

hipError_t launch_prior_dlt(const float* prior, float* H, int batch, hipStream_t s) {
    hipLaunchKernelGGL(dlt_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, prior, H, batch, 1);
    return hipGetLastError();
}
hipError_t launch_dlt(const float* dst, float* H, int n, hipStream_t s) {
    hipLaunchKernelGGL(dlt_kernel, dim3((n + 63) / 64), dim3(64), 0, s, dst, H, n, 0);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// block tail: fc_block_k = Linear(5120, 8) on the flattened trunk output, corner update, DLT, compose
//   (model_to_trace.py:143-150, :163-168, :183-188).  One 256-thread workgroup per frame pair; the 8 dot
//   products are reduced in a fixed order (wave shuffle tree, then 4 partials summed by lane 0).
//   feat: NHWC flatten [20][256]; wfc is pre-permuted to the same order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void block_fc_dlt_kernel(const float* __restrict__ feat, const float* __restrict__ wfc,
                                                           const float* __restrict__ bfc, const float* __restrict__ H_in,
                                                           float* __restrict__ H_out) {
    __shared__ float part[4][8];
    __shared__ float hout[9];
    const int b = blockIdx.x;
    fc_dlt_block(feat + (size_t)b * 5120, wfc, bfc, H_in ? H_in + b * 9 : nullptr, part, hout);
    if (threadIdx.x == 0)
        for (int i = 0; i < 9; i++) H_out[b * 9 + i] = hout[i];
}

hipError_t launch_block_fc_dlt(const float* feat, const float* wfc, const float* bfc, const float* H_in,
                               float* H_out, int batch, hipStream_t s) {
    hipLaunchKernelGGL(block_fc_dlt_kernel, dim3(batch), dim3(256), 0, s, feat, wfc, bfc, H_in, H_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// heads, second half: Dropout -> Linear(256, 8) for both heads and every local MC sample
//   (model_to_trace.py:226-227,233-234; run_fc :252-256 scales the uncertainty head by 1e-3).
//   hidden [B*n_local][512] holds LeakyReLU(Linear(5120,256)) of both heads (columns 0..255 mean head).
//   One workgroup per (pair, chunk of 4 samples): 64 dot products of length 256, four lanes per dot; every lane
//   walks its 64 elements in a rotated order so that the 32 lanes served per LDS cycle hit 32 different banks.
//   Per-sample outputs go to mean_s / logvar_s [B][n_local][8]; the ensemble is a separate launch (mc_finish).
// ---------------------------------------------------------------------------------------------
constexpr int FC2_CHUNK = 4;

// one chunk (up to FC2_CHUNK samples) of one pair by 256 threads (t = 0..255 inside the chunk's thread group); w2s [2][8][256] already staged;
// hid / pre_row: this group's LDS slices.  Every thread of the workgroup must call it (workgroup barriers inside); nc <= 0: nothing stored.
// The per-sample results go to mean_o / logvar_o [n_local][8] of the pair (global memory, or LDS in the merged kernel).
// pre: the thread's FC2_CHUNK * 512 / 256 = 8 elements of the chunk (i = t, t + 256, ...) loaded ahead of time (nullptr: loaded here)
__device__ __forceinline__ void heads_fc2_chunk(const float* __restrict__ hidden_b, int n_local, int s_begin, int c0, int nc, int t, uint32_t thr,
                                                float scale, uint64_t key, const float* w2s, const float* __restrict__ b2, float* hid,
                                                uint32_t* pre_row, float* mean_o, float* logvar_o, uint32_t* __restrict__ flag, const float* pre = nullptr) {
    // the hash prefix of a (sample, head) row - four hnet_mix32 - once per row, not once per element
    if (t < FC2_CHUNK * 2) pre_row[t] = hnet_mask_prefix(key, (uint32_t)(2 * (t & 1) + 1), (uint32_t)(s_begin + c0 + (t >> 1)));
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FC2_CHUNK * 512 / 256; k++) {
        const int i = t + 256 * k;
        const int sl = i >> 9, col = i & 511, head = col >> 8, j = col & 255;
        float v = 0.0f;
        if (sl < nc) {
            const uint32_t prefix = pre_row[sl * 2 + head];
            const float x = pre ? pre[k] : hidden_b[(size_t)(c0 + sl) * 512 + col];
            v = hnet_mask_keep(prefix, (uint32_t)j, thr) ? x * scale : 0.0f;
        }
        hid[i] = v;
    }
    __syncthreads();
    const int d = t >> 2, part = t & 3;      // dot index (sample, output), quarter of the dot
    const int sl = d >> 4, o = d & 15, head = o >> 3, oi = o & 7;
    const float* hrow = hid + sl * 512 + head * 256 + part * 64;
    const float* wrow = w2s + (head * 8 + oi) * 256 + part * 64;
    // 16-byte LDS reads (round 4: one value at a time the loop was 128 ds_read_b32 + the index arithmetic per thread); the start is rotated per (output, quarter)
    // so that the threads of a wave - which read rows 1 KB apart - do not all start on one bank
    const int rot = ((o & 7) * 4 + part) & 15;
    float acc = 0.0f;
#pragma unroll 8
    for (int j = 0; j < 16; j++) {
        const int jj = ((j + rot) & 15) * 4;
        const float4 hv = *reinterpret_cast<const float4*>(hrow + jj), wv = *reinterpret_cast<const float4*>(wrow + jj);
        acc = fmaf(hv.x, wv.x, acc);
        acc = fmaf(hv.y, wv.y, acc);
        acc = fmaf(hv.z, wv.z, acc);
        acc = fmaf(hv.w, wv.w, acc);
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (part == 0 && sl < nc) {
        float v = acc + b2[head * 8 + oi];
        float* dst = head == 0 ? mean_o : logvar_o;
        if (head == 1) v *= 1e-3f;
        dst[(size_t)(c0 + sl) * 8 + oi] = v;
        if (flag && !(fabsf(v) <= 3.0e38f)) atomicOr(flag, 1u);       // hnet_overflow_flag: a non-finite per-sample output (the *_partial path ends here)
    }
}

__global__ __launch_bounds__(256) void heads_fc2_kernel(const float* __restrict__ hidden, int n_local, int s_begin,
                                                        uint32_t thr, float scale, uint64_t mc_seed, uint64_t pair_seq0,
                                                        const uint64_t* __restrict__ seq_dev, const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ mean_s, float* __restrict__ logvar_s, uint32_t* __restrict__ flag) {
    __shared__ __attribute__((aligned(16))) float w2s[4096];                  // [2][8][256]
    __shared__ __attribute__((aligned(16))) float hid[FC2_CHUNK * 512];       // after dropout
    __shared__ uint32_t pre_row[FC2_CHUNK * 2];
    const int n_chunks = (n_local + FC2_CHUNK - 1) / FC2_CHUNK;
    const int b = blockIdx.x / n_chunks, c0 = (blockIdx.x % n_chunks) * FC2_CHUNK;
    const int nc = min(FC2_CHUNK, n_local - c0);
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) w2s[i] = w2[i];
    const uint64_t key = hnet_pair_key(mc_seed, pair_seq0 + (seq_dev ? *seq_dev : 0ull) + (uint64_t)b);
    heads_fc2_chunk(hidden + (size_t)b * n_local * 512, n_local, s_begin, c0, nc, tid, thr, scale, key, w2s, b2, hid, pre_row,
                    mean_s + (size_t)b * n_local * 8, logvar_s + (size_t)b * n_local * 8, flag);
}

hipError_t launch_heads_fc2(const float* hidden, int batch, int n_local, int s_begin, float p, uint64_t mc_seed,
                            uint64_t pair_seq0, const float* w2, const float* b2, float* mean_s, float* logvar_s,
                            hipStream_t s, const uint64_t* seq_dev, uint32_t* flag) {
    if (n_local < 1 || !mean_s || !logvar_s) return hipErrorInvalidValue;
    const int n_chunks = (n_local + FC2_CHUNK - 1) / FC2_CHUNK;
    hipLaunchKernelGGL(heads_fc2_kernel, dim3((unsigned)(batch * n_chunks)), dim3(256), 0, s, hidden, n_local, s_begin,
                       hnet_drop_threshold(p), 1.0f / (1.0f - p), mc_seed, pair_seq0, seq_dev, w2, b2, mean_s, logvar_s, flag);
    return hipGetLastError();
}

// ensemble + transfer + assembly from per-sample outputs [B][n][8] (model_to_trace.py:274-280): one wave per pair.  Lane (component
// i = lane & 7, sample chunk c = lane >> 3) sums samples c, c + 8, ... in double, the eight chunks are combined by xor-shuffles (a fixed
// tree: the result does not depend on the batch size or the slot), then the reference's second pass over the squared deviations
// from the fp32 mean; lane 0 finishes (geom.h).  (Round 1 ran 8 lanes per pair over all samples in sequence: 11.4 us at batch 1, most
// of it 2 x 32 dependent double-precision exp / add chains per lane.)
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}
// the ensemble + transfer of ONE pair by one wave (t = lane); ms / lv: the pair's per-sample outputs [n][8] (global memory or LDS)
// n_local / rank_stride (round 5): sample s of the pair lives at (s / n_local) * rank_stride + (s % n_local) * 8 floats - the layout an all-gather of every
// rank's [B][n_local][8] block produces ([rank][...]); rank_stride = 0: all n samples consecutive (n_local unused)
__device__ __forceinline__ void mc_finish_wave(const float* ms, const float* lv, int n, const float* __restrict__ H1_b, float* __restrict__ mean_b,
                                               float* __restrict__ cov_b, float* __restrict__ Htot_b, uint32_t* __restrict__ flag, int t,
                                               int n_local = 0, size_t rank_stride = 0) {
    const int i = t & 7, c = t >> 3;
    auto at = [&](int s) -> size_t { return rank_stride ? (size_t)(s / n_local) * rank_stride + (size_t)(s % n_local) * 8 + i : (size_t)s * 8 + i; };
    double sm = 0, sv = 0;
    for (int s = c; s < n; s += 8) {
        sm += (double)ms[at(s)];
        sv += exp((double)lv[at(s)]);
    }
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) { sm += shfl_xor_f64(sm, m); sv += shfl_xor_f64(sv, m); }
    const float mb = (float)(sm / n), vb = (float)(sv / n);
    double se = 0;
    for (int s = c; s < n; s += 8) { const double d = (double)mb - (double)ms[at(s)]; se += d * d; }
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) se += shfl_xor_f64(se, m);
    const double en = (double)(float)((double)(float)(se / n) + (double)vb);
    const double pb = (double)(float)(p4(i) + (double)mb);
    auto bcast = [](double v, int src) { return __hiloint2double(__shfl(__double2hiint(v), src), __shfl(__double2loint(v), src)); };
    double pbar[8];                                // every lane: the eight corner coordinates (lane 4 solves the DLT with them)
#pragma unroll
    for (int k = 0; k < 8; k++) pbar[k] = bcast(pb, k);
    // lane c < 4 works on corner c: its coordinates and ensemble variances come from lanes 2c, 2c + 1 (no lane-indexed arrays)
    const int cs = 2 * (t & 3);
    const double pu = bcast(pb, cs), pv = bcast(pb, cs + 1), vu = bcast(en, cs), vv = bcast(en, cs + 1);
    // transfer + assembly (geom.h transfer_pair), spread over the wave: lane c < 4 takes corner c (its two means and its 2 x 2 covariance
    // block), lane 4 the total homography, all 64 lanes the zero fill of the block-diagonal 8 x 8 matrix.  Same formulas per entry as the
    // one-thread form (the CPU restatement of the tests evaluates that one).
    double H1[9];
#pragma unroll
    for (int k = 0; k < 9; k++) H1[k] = (double)H1_b[k];
    const bool diag = ((t >> 3) >> 1) == ((t & 7) >> 1);           // entry (row t >> 3, column t & 7) lies in a 2 x 2 diagonal block
    if (!diag) cov_b[t] = 0.0f;
    bool bad = false;
    if (t < 4) {
        const int c = t;
        const double X = H1[0] * pu + H1[1] * pv + H1[2];
        const double Y = H1[3] * pu + H1[4] * pv + H1[5];
        const double S = H1[6] * pu + H1[7] * pv + H1[8];
        const float m0 = (float)(X / S - p4(2 * c)), m1 = (float)(Y / S - p4(2 * c + 1));
        const double g00 = H1[0] / S, g01 = H1[1] / S, g10 = H1[3] / S, g11 = H1[4] / S;
        const float c00 = (float)(g00 * vu * g00 + g01 * vv * g01), c01 = (float)(g00 * vu * g10 + g01 * vv * g11);
        const float c10 = (float)(g10 * vu * g00 + g11 * vv * g01), c11 = (float)(g10 * vu * g10 + g11 * vv * g11);
        mean_b[2 * c] = m0; mean_b[2 * c + 1] = m1;
        cov_b[(2 * c) * 8 + 2 * c] = c00; cov_b[(2 * c) * 8 + 2 * c + 1] = c01;
        cov_b[(2 * c + 1) * 8 + 2 * c] = c10; cov_b[(2 * c + 1) * 8 + 2 * c + 1] = c11;
        bad = !(fabsf(m0) <= 3.0e38f && fabsf(m1) <= 3.0e38f && fabsf(c00) <= 3.0e38f && fabsf(c01) <= 3.0e38f && fabsf(c10) <= 3.0e38f &&
                fabsf(c11) <= 3.0e38f);
    } else if (t == 4 && Htot_b) {
        double Hb[9], Ht[9];
        dlt_solve(pbar, Hb);
        mat3_mul(H1, Hb, Ht);
        for (int k = 0; k < 9; k++) Htot_b[k] = (float)Ht[k];
    }
    if (flag && bad) atomicOr(flag, 1u);           // hnet_overflow_flag: a non-finite output of this pair
}
// out_stride: floats between the records of consecutive pairs in `mean` and in `cov` (8 / 64 for separate arrays; 72 / 72 for the packed
// [B][72] record mean | cov that a multi-GPU caller all-gathers: cov = mean + 8)
__global__ __launch_bounds__(64) void mc_finish_kernel(const float* __restrict__ mean_s, const float* __restrict__ logvar_s, int n,
                                                       const float* __restrict__ H1, int batch, float* __restrict__ mean,
                                                       float* __restrict__ cov, float* __restrict__ Htot, uint32_t* __restrict__ flag, int mean_stride, int cov_stride,
                                                       int n_local, size_t rank_stride) {
    const int b = blockIdx.x;
    const size_t pair = (size_t)b * (rank_stride ? n_local : n) * 8;
    mc_finish_wave(mean_s + pair, logvar_s + pair, n, H1 + b * 9, mean + (size_t)b * mean_stride, cov + (size_t)b * cov_stride,
                   Htot ? Htot + b * 9 : nullptr, flag, threadIdx.x, n_local, rank_stride);
}

// Small batches (latency path): second FC of the heads for ALL samples of a pair and the ensemble in ONE launch, one 1024-thread workgroup
// per pair: four 256-thread groups each take a chunk of FC2_CHUNK samples per round (heads_fc2_chunk, same arithmetic as heads_fc2_kernel),
// the per-sample outputs stay in LDS, wave 0 finishes the pair (mc_finish_wave): bit-identical to the two launches it replaces.
constexpr int FC2M_MAX_N = 64;
__global__ __launch_bounds__(1024) void heads_fc2_finish_kernel(const float* __restrict__ hidden, int n_local, int s_begin, uint32_t thr, float scale,
                                                                uint64_t mc_seed, uint64_t pair_seq0, const uint64_t* __restrict__ seq_dev,
                                                                const float* __restrict__ w2, const float* __restrict__ b2,
                                                                const float* __restrict__ H1, float* __restrict__ mean, float* __restrict__ cov,
                                                                float* __restrict__ Htot, uint32_t* __restrict__ flag, int mean_stride, int cov_stride) {
    __shared__ __attribute__((aligned(16))) float w2s[4096];
    __shared__ __attribute__((aligned(16))) float hid[4][FC2_CHUNK * 512];
    __shared__ uint32_t pre_row[4][FC2_CHUNK * 2];
    __shared__ float ms_l[FC2M_MAX_N * 8], lv_l[FC2M_MAX_N * 8];
    __shared__ float h1_l[9];
    const int b = blockIdx.x, tid = threadIdx.x, grp = tid >> 8, t = tid & 255;
    // round 5: every global load of the kernel is issued before the first dependent instruction - the sequence number, the weights, the homography
    // and the thread's hidden activations of the first two rounds (all of them up to N = 32) - instead of one memory round trip per step
    // (15.3 -> ... us at batch 1; same values, same arithmetic)
    const float* hidden_b = hidden + (size_t)b * n_local * 512;
    const int n_chunks = (n_local + FC2_CHUNK - 1) / FC2_CHUNK;
    const uint64_t seqv = seq_dev ? *seq_dev : 0ull;
    float w2r[4], hpre[2][FC2_CHUNK * 512 / 256];
#pragma unroll
    for (int k = 0; k < 4; k++) w2r[k] = w2[tid + 1024 * k];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int c0 = (r * 4 + grp) * FC2_CHUNK, nc = min(FC2_CHUNK, n_local - c0);
#pragma unroll
        for (int k = 0; k < FC2_CHUNK * 512 / 256; k++) {
            const int i = t + 256 * k, sl = i >> 9;
            hpre[r][k] = sl < nc ? hidden_b[(size_t)(c0 + sl) * 512 + (i & 511)] : 0.0f;
        }
    }
    if (tid < 9) h1_l[tid] = H1[b * 9 + tid];
#pragma unroll
    for (int k = 0; k < 4; k++) w2s[tid + 1024 * k] = w2r[k];
    const uint64_t key = hnet_pair_key(mc_seed, pair_seq0 + seqv + (uint64_t)b);
    // (the first barrier inside heads_fc2_chunk also covers the w2s / h1_l fill)
    heads_fc2_chunk(hidden_b, n_local, s_begin, grp * FC2_CHUNK, min(FC2_CHUNK, n_local - grp * FC2_CHUNK), t, thr, scale, key, w2s, b2, hid[grp],
                    pre_row[grp], ms_l, lv_l, nullptr, hpre[0]);
    if (n_chunks > 4)
        heads_fc2_chunk(hidden_b, n_local, s_begin, (4 + grp) * FC2_CHUNK, min(FC2_CHUNK, n_local - (4 + grp) * FC2_CHUNK), t, thr, scale, key, w2s, b2, hid[grp],
                        pre_row[grp], ms_l, lv_l, nullptr, hpre[1]);
    for (int r = 2; r * 4 < n_chunks; r++) {
        const int c0 = (r * 4 + grp) * FC2_CHUNK;
        heads_fc2_chunk(hidden_b, n_local, s_begin, c0, min(FC2_CHUNK, n_local - c0), t, thr, scale, key, w2s, b2, hid[grp],
                        pre_row[grp], ms_l, lv_l, nullptr);
    }
    __syncthreads();
    if (tid < 64) mc_finish_wave(ms_l, lv_l, n_local, h1_l, mean + (size_t)b * mean_stride, cov + (size_t)b * cov_stride, Htot ? Htot + b * 9 : nullptr, flag, tid);
}
hipError_t launch_heads_fc2_finish(const float* hidden, int batch, int n_local, int s_begin, float p, uint64_t mc_seed, uint64_t pair_seq0, const float* w2,
                                   const float* b2, const float* H1, float* mean, float* cov, float* Htot, hipStream_t s, const uint64_t* seq_dev,
                                   uint32_t* flag, int mean_stride, int cov_stride) {
    if (n_local < 1 || n_local > FC2M_MAX_N) return hipErrorInvalidValue;
    hipLaunchKernelGGL(heads_fc2_finish_kernel, dim3((unsigned)batch), dim3(1024), 0, s, hidden, n_local, s_begin, hnet_drop_threshold(p), 1.0f / (1.0f - p),
                       mc_seed, pair_seq0, seq_dev, w2, b2, H1, mean, cov, Htot, flag, mean_stride, cov_stride);
    return hipGetLastError();
}

hipError_t launch_mc_finish(const float* mean_s, const float* logvar_s, int n, const float* H1, int batch,
                            float* mean, float* cov, float* Htot, hipStream_t s, uint32_t* flag, int mean_stride, int cov_stride, int n_local, size_t rank_stride) {
    if (rank_stride && (n_local < 1 || n % n_local)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mc_finish_kernel, dim3(batch), dim3(64), 0, s, mean_s, logvar_s, n, H1, batch, mean, cov, Htot, flag, mean_stride, cov_stride, n_local, rank_stride);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// layout helpers (operator-level entry points and debug read-back only)
// ---------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int batch, int c, int hw) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)batch * c * hw) return;
    const int ch = (int)(idx % c);
    const long t = idx / c;
    const int p = (int)(t % hw), b = (int)(t / hw);
    out[idx] = in[((size_t)b * c + ch) * hw + p];
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int batch, int c, int hw) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)batch * c * hw) return;
    const int p = (int)(idx % hw);
    const long t = idx / hw;
    const int ch = (int)(t % c), b = (int)(t / c);
    out[idx] = in[((size_t)b * hw + p) * c + ch];
}
hipError_t launch_nchw_to_nhwc(const float* in, float* out, int batch, int c, int h, int w, hipStream_t s) {
    const long n = (long)batch * c * h * w;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, batch, c, h * w);
    return hipGetLastError();
}
hipError_t launch_nhwc_to_nchw(const float* in, float* out, int batch, int c, int h, int w, hipStream_t s) {
    const long n = (long)batch * c * h * w;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, batch, c, h * w);
    return hipGetLastError();
}

}  // namespace hnet
