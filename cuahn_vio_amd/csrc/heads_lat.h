// heads_lat.h — first FC of both heads for SMALL batches (latency path, round 5; fp16-plane mode): Dropout -> Linear(5120, 256) -> LeakyReLU of the
// mean and the uncertainty head for every MC sample of a pair (reference model_to_trace.py:222-225, 229-232, run_fc :252-256), as ONE launch.
//
// At batch 1 the GEMM is 32 (samples) x 512 x 5120: the split-K form (igemm_s3_lean_kernel, 64 x 64 tiles x 20 K-slices, then splitk_reduce_kernel, after
// heads_prep_kernel had written the feature planes) is three dependent launches, 22 us.  Here the OUTPUT is split instead: a workgroup owns HL_UN = 4 of the
// 512 hidden units and the whole K; it copies its 4 x 5120 x 2 planes of weights (80 KB) into LDS by LDS-DMA - every load of the kernel in flight at once, one
// memory round trip - while its threads split the pair's 5120 features into fp16 planes (20 KB of LDS; no feature planes in memory).  The sixteen waves each
// take a 320-wide K-slice: ten 16x16x32 MFMA steps with the weights as A operand (rows = hidden units: 4 of the 16 rows are real, the others repeat them and are
// ignored - the kernel is bound by the round trip, not by the matrix pipe) and the masked features as B operand (columns = 16 MC samples; keep bits from the bit
// array of heads_prep_kernel through a 16-entry LDS table).  The sixteen partial tiles are summed through LDS in wave order, then
// bias + LeakyReLU -> hidden [B * n_local][512].  No split-K, no reduce launch, no workspace.  Arithmetic: the two-plane fp16 form of igemm_s3.h
// (hi += W0 A0, lo += W0 A1 + W1 A0, result = hi + lo / 4096), K summed in a different order than the split-K kernels: results agree to fp32 rounding (tested).
#pragma once
#include "igemm_s3.h"

namespace hnet {

constexpr int HL_UN = 4;                       // hidden units per workgroup
constexpr int HL_NT = 1024;                    // sixteen waves
constexpr int HL_KW = 5120 / 16;               // K-slice of a wave: 320 = ten MFMA steps
constexpr int HL_ROW = 5120 * 2 + 64;          // bytes of one (plane, unit) weight row in LDS: the 64-byte pad puts the four rows of a fragment read in different banks
constexpr int HL_MAXG = 4;                     // most sample groups of 16: n_local <= 64 (the kernel is instantiated for 2 and 4 groups)
constexpr int HL_W_BYTES = 2 * HL_UN * HL_ROW;
constexpr int HL_F_BYTES = 2 * 5120 * 2;
constexpr int HL_LUT_BYTES = 256 * 16;
constexpr int HL_RED_BYTES = 16 * HL_MAXG * 16 * 4 * 4;
constexpr int HL_LDS_BYTES = HL_W_BYTES + HL_F_BYTES + HL_LUT_BYTES + HL_RED_BYTES;

typedef __attribute__((address_space(3))) void* hl_lds_ptr_t;

// feat [B][5120] fp32 (NHWC flatten, LeakyReLU applied); w1planes [2][512][5120] fp16 (W0 = f16(w), W1 = f16((w - W0) 4096): wsplit_gemm of hnet_create);
// mask [B][n_local][2 heads][640] keep bits (heads_prep_kernel, row-major layout); hidden [B * n_local][512]
// Round 6: a workgroup keeps its weights in LDS for `ppw` consecutive pairs (grid.y = ceil(batch / ppw)): at batch 8 the 1 024 workgroups of (unit group, pair)
// each fetched 80 KB - four rounds of one-workgroup-per-CU residents, 36.5 us; with four pairs per workgroup the launch is one round of 256 and the weights cross
// the CU's port once: 31 us (N = 32; what remains is ~ 5 us per pair of MFMA issue at a quarter of the rows + three barriers).  The next pair's features and keep
// bits are requested before this pair's MFMAs.  Per pair the arithmetic is unchanged (same bits).
template <int MAXG>
__global__ __launch_bounds__(HL_NT) void heads_fc1_lat_kernel(const float* __restrict__ feat, const uint16_t* __restrict__ w1planes, size_t w_plane,
                                                             const float* __restrict__ b1, const uint8_t* __restrict__ mask, int n_local, float scale,
                                                             float* __restrict__ hidden, int batch, int ppw) {
    extern __shared__ __attribute__((aligned(16))) uint8_t hl_smem[];
    uint8_t* w_lds = hl_smem;
    uint16_t* f_lds = reinterpret_cast<uint16_t*>(hl_smem + HL_W_BYTES);
    uint8_t* lut = hl_smem + HL_W_BYTES + HL_F_BYTES;
    float* red = reinterpret_cast<float*>(hl_smem + HL_W_BYTES + HL_F_BYTES + HL_LUT_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u0 = blockIdx.x * HL_UN, head = u0 >> 8;
    const int b_first = blockIdx.y * ppw, b_end = min(batch, b_first + ppw);
    const int groups = (n_local + 15) >> 4;

    // ---- every global load of the first pair is issued here
    // (1) weights: 2 planes x 4 units x 10 KB as 80 one-KB LDS-DMA copies, five per wave
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)w1planes, 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int c = wave + 16 * j, pl = c / 40, u = (c / 10) % HL_UN, seg = c % 10;      // (wave-uniform)
        const uint32_t vo = (uint32_t)((((size_t)(u0 + u) * 5120 + seg * 512) * 2) + lane * 16);
        const int so = (int)(pl * w_plane * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (hl_lds_ptr_t)(w_lds + (pl * HL_UN + u) * HL_ROW + seg * 1024), 16, vo, so, 0, 0);
    }
    // (2) keep bits of this lane's sample (column n of every group) for the wave's K-slice: 40 bytes per (sample, head) row; (3) the pair's features (two float4 per
    // thread at most: 5 120 = 1 024 x 4 + 256 x 4)
    const int n16 = lane & 15, kg = lane >> 4;
    uint2 mbits[MAXG][5];
    float4 fv[2];
    auto mask_loads = [&](int b) {
#pragma unroll
        for (int g = 0; g < MAXG; g++)
            if (g < groups) {
                const int sm = min(g * 16 + n16, n_local - 1);                              // (columns beyond n_local repeat the last sample and are not stored)
                const uint8_t* mrow = mask + (((size_t)b * n_local + sm) * 2 + head) * 640 + wave * 40;
#pragma unroll
                for (int q = 0; q < 5; q++) mbits[g][q] = *reinterpret_cast<const uint2*>(mrow + 8 * q);
            }
    };
    auto feat_loads = [&](int b) {
        fv[0] = *reinterpret_cast<const float4*>(feat + (size_t)b * 5120 + tid * 4);
        fv[1] = tid < 256 ? *reinterpret_cast<const float4*>(feat + (size_t)b * 5120 + HL_NT * 4 + tid * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    mask_loads(b_first);
    feat_loads(b_first);
    if (tid < 16) {        // entry x of the table: 4 keep bits -> 4 x 16-bit lane masks (8 bytes); sixteen 8-byte entries = banks 0 .. 31 once: a lookup is conflict free
        // whatever the lanes' bits (equal entries broadcast).  (Round 5 used 256 entries x 16 bytes; measured the same - the kernel's time at batch 8 is the matrix pipe:
        // 4 of an MFMA's 16 rows are real, 240 MFMAs x 16 cycles per SIMD and pair.)
        uint2 e;
        e.x = ((tid >> 0) & 1u) * 0xFFFFu | ((tid >> 1) & 1u) * 0xFFFF0000u;
        e.y = ((tid >> 2) & 1u) * 0xFFFFu | ((tid >> 3) & 1u) * 0xFFFF0000u;
        *reinterpret_cast<uint2*>(lut + tid * 8) = e;
    }
    const uint8_t* wrow = w_lds + (n16 & (HL_UN - 1)) * HL_ROW + (wave * HL_KW + 8 * kg) * 2;
    const uint16_t* frow = f_lds + wave * HL_KW + 8 * kg;

    for (int b = b_first; b < b_end; b++) {
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this pair's loads (and, the first time, the DMA copies of this wave) have landed
        // the pair's features: scaled by 1 / (1 - p), split into fp16 planes, to LDS (the previous pair's MFMAs are behind the barrier that followed its reduction)
        {
            uint32_t q0[3], q1[3];
            s3p::split_pair<2>(fv[0].x * scale, fv[0].y * scale, q0);
            s3p::split_pair<2>(fv[0].z * scale, fv[0].w * scale, q1);
            *reinterpret_cast<uint2*>(f_lds + tid * 4) = make_uint2(q0[0], q1[0]);
            *reinterpret_cast<uint2*>(f_lds + 5120 + tid * 4) = make_uint2(q0[1], q1[1]);
            if (tid < 256) {
                s3p::split_pair<2>(fv[1].x * scale, fv[1].y * scale, q0);
                s3p::split_pair<2>(fv[1].z * scale, fv[1].w * scale, q1);
                *reinterpret_cast<uint2*>(f_lds + HL_NT * 4 + tid * 4) = make_uint2(q0[0], q1[0]);
                *reinterpret_cast<uint2*>(f_lds + 5120 + HL_NT * 4 + tid * 4) = make_uint2(q0[1], q1[1]);
            }
        }
        __syncthreads();
        // the next pair's features and keep bits: in flight under this pair's MFMAs and reduction (this pair's keep bits move to a copy).  With four sample groups -
        // N > 32 - there are no registers for that (128 per thread at sixteen waves): both follow the MFMAs
        uint2 mb[MAXG <= 2 ? MAXG : 1][5];
        if constexpr (MAXG <= 2) {
#pragma unroll
            for (int g = 0; g < MAXG; g++)
#pragma unroll
                for (int q = 0; q < 5; q++) mb[g][q] = mbits[g][q];
            if (b + 1 < b_end) { mask_loads(b + 1); feat_loads(b + 1); }
        }

        // ---- ten MFMA steps over the wave's K-slice
        f32x4_m16 hi[MAXG], lo[MAXG];
#pragma unroll
        for (int g = 0; g < MAXG; g++) { hi[g] = f32x4_m16{0.f, 0.f, 0.f, 0.f}; lo[g] = f32x4_m16{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int t = 0; t < 10; t++) {
            bf16x8 w[3], f[2];
            w[0] = *reinterpret_cast<const bf16x8*>(wrow + t * 64);
            w[1] = *reinterpret_cast<const bf16x8*>(wrow + HL_UN * HL_ROW + t * 64);
            f[0] = *reinterpret_cast<const bf16x8*>(frow + t * 32);
            f[1] = *reinterpret_cast<const bf16x8*>(frow + 5120 + t * 32);
#pragma unroll
            for (int g = 0; g < MAXG; g++)
                if (g < groups) {
                    // byte (4 t + kg) of the row's 40: dword t of the ten, byte kg
                    uint32_t dw;
                    if constexpr (MAXG <= 2) dw = (t & 1) ? mb[g][t >> 1].y : mb[g][t >> 1].x;
                    else dw = (t & 1) ? mbits[g][t >> 1].y : mbits[g][t >> 1].x;
                    const uint32_t mbyte = (dw >> (8 * kg)) & 0xFFu;
                    const uint2 m0 = *reinterpret_cast<const uint2*>(lut + (mbyte & 15u) * 8), m1 = *reinterpret_cast<const uint2*>(lut + (mbyte >> 4) * 8);
                    const u32x4 mk = {m0.x, m0.y, m1.x, m1.y};
                    bf16x8 a[3];
                    a[0] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, f[0]) & mk);
                    a[1] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, f[1]) & mk);
                    s3_mfma16_2acc(hi[g], lo[g], w, a);
                }
        }
        // ---- the sixteen K-slices through LDS, in wave order: rows 0 .. 3 of the tile (lanes 0 .. 15) are the workgroup's units
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MAXG > 2) { if (b + 1 < b_end) { mask_loads(b + 1); feat_loads(b + 1); } }
#pragma unroll
        for (int g = 0; g < MAXG; g++)
            if (g < groups && lane < 16) {
                const f32x4_m16 v = hi[g] + lo[g] * S3_F16_INV;
                *reinterpret_cast<f32x4_m16*>(red + ((wave * MAXG + g) * 16 + lane) * 4) = v;
            }
        __syncthreads();                         // (also: every wave is done with this pair's feature planes)
        if (tid < groups * 64) {
            const int g = tid >> 6, n = (tid >> 2) & 15, j = tid & 3, sm = g * 16 + n;
            float s = red[((0 * MAXG + g) * 16 + n) * 4 + j];
#pragma unroll
            for (int wv = 1; wv < 16; wv++) s += red[((wv * MAXG + g) * 16 + n) * 4 + j];
            const float v = s + b1[u0 + j];
            if (sm < n_local) hidden[((size_t)b * n_local + sm) * 512 + u0 + j] = v > 0.0f ? v : v * 0.1f;
        }
        // (the next pair's partial tiles are written to `red` behind the barrier that follows its feature planes: the readers above have passed it by then)
    }
}

}  // namespace hnet
