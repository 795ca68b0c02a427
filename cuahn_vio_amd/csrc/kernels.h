// kernels.h — launch wrappers of the HIP kernels (internal to libhnet_hip.so)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hnet {

struct ConvDesc {
    const char* name;
    int cin, cout, ks, stride;
    int seg;     // K-segment length used by the packed weights (see igemm.h)
    int block;   // 1..4
};
extern const ConvDesc kConvs[20];
int conv_padded_k(int layer);    // Kp of the packed weight matrix [Cout][Kp] (0 for the direct first layers)
bool conv_is_first_direct(int layer);   // layers served by conv_first.h (weights = MFMA fragments)
inline int conv_out_dim(int n, int ks, int stride) { return (n + 2 * ((ks - 1) / 2) - ks) / stride + 1; }

// conv + bias + LeakyReLU on NHWC fp32: in [B][H][W][Cin] -> out [B][Ho][Wo][Cout]
//   ws / ws_floats: optional split-K workspace (nullptr = never split)
//   out16 / o_plane: when set, the output is written as three bf16 planes (S3, igemm_s3.h) instead of fp32
hipError_t launch_conv(int layer, const float* in, int batch, int h, int w, const float* wpacked,
                       const float* bias, float* out, hipStream_t s, float* ws = nullptr, size_t ws_floats = 0,
                       uint16_t* out16 = nullptr, size_t o_plane = 0);

// n_planes (all matrix-core launchers) = the arithmetic mode (s3_format.h): 3 = split-bf16 (fp32-grade), 1 = plain bf16 operands (only plane 0
// is read / written), 2 = two fp16 planes (fp32-grade, three MFMAs per product; planes 0 and 1)
// split-bf16 (S3) convolution of the layers with Cin >= 8: in / out16 are [3][B][H][W][C] bf16 planes
// (plane stride in elements), wplanes [3][Cout][Kp]; out32 != nullptr selects an fp32 [B][Ho][Wo][Cout] output
bool conv_is_s3_layer(int layer);
// wfrag (fp16-plane mode, conv_region_layer(layer)): the weights as MFMA fragments in the order igemm_region.h consumes them (packed by hnet_create), or nullptr
bool conv_region_layer(int layer);            // block_1_2, block_1_3, block_2_4 / 3_5 / 4_6: input region resident in LDS, weights straight to registers
int conv_region_taps_padded(int layer);       // taps per 64-channel chunk in the packed fragments (the K-split form pads its second half with a zero tap)
// Round 5: optional extras of launch_conv_s3 / launch_heads_fc1_s3 for the split-K launches of small batches.
// tickets: SPLITK_TICKETS zeroed 32-bit counters; a split-K tile of at most 40 GEMM rows is then finished by its last-arriving workgroup (igemm_s3.h
// s3_splitk_last_arriver: same sums in the same order as splitk_reduce*: same bits) and the counters are left at zero; launches sharing them are stream ordered.
constexpr int SPLITK_TICKETS = 4096;
struct LatIO {
    uint32_t* tickets;        // nullptr: splitk_reduce* launches
    int kernels;              // result: kernels this call launched (a split-K layer with a splitk_reduce* launch: 2; the heads' first FC: 1 - 3)
    bool heads_one_launch;    // launch_heads_fc1_s3: take the one-launch kernel of heads_lat.h where it applies (heads_fc1_one_launch)
    bool mask_ready;          // launch_heads_fc1_s3: the keep bits have been written already (by the surplus workgroups of block 4's prep launch, FcArgs::mask)
};
bool heads_fc1_one_launch(int batch, int n_local, int n_planes);      // whether launch_heads_fc1_s3 with a LatIO takes the one-launch kernel of heads_lat.h
hipError_t launch_conv_s3(int layer, const uint16_t* in, size_t in_plane, int batch, int h, int w, const uint16_t* wplanes,
                          size_t w_plane, const float* bias, uint16_t* out16, size_t o_plane, float* out32, hipStream_t s,
                          float* ws = nullptr, size_t ws_floats = 0, const uint16_t* wfrag = nullptr, int n_planes = 3, int tile = 0, LatIO* lat = nullptr);
bool conv_is_patch_layer(int layer);      // block_3_1 / block_4_2 (conv_patch_s2.h), split-bf16 mode
bool conv_is_patch32_layer(int layer);    // block_3_2 / block_4_3 at their network size 56x80 (conv_patch32_s2_kernel); same launcher, wfrag [4][9][3][64] x 16 B
hipError_t launch_conv_patch(int layer, const uint16_t* in, size_t i_plane, int batch, int h, int w, const void* wfrag,
                             const float* bias, uint16_t* out16, size_t o_plane, hipStream_t s, int n_planes = 3,
                             bool b128 = false /* block_3_1 / block_4_2: interleaved-half region layout + ds_read_b128 (wfrag packed without the odd-group rotation) */,
                             int rb5 = 1 /* 5x5 kernel, fp16 mode: region rows staged per batch of loads (1, 2 or 5) */);
hipError_t launch_conv_first_s3(const float* x_in, const void* wfrag, const float* bias, uint16_t* out16, size_t o_plane, int batch,
                                int h, int w, hipStream_t s, int n_planes = 3);
// block_1_1 (layer 0) / block_2_1 (layer 3): 7x7 stride 2, Cin 2, fp32 [B][h][w][2] in -> S3 planes out (conv_first.h)
inline bool conv_is_first_s2(int layer) { return layer == 0 || layer == 3; }
hipError_t launch_conv_first_s2(int layer, const float* x_in, const void* wfrag, const float* bias, uint16_t* out16, size_t o_plane, int batch,
                                hipStream_t s, int n_planes = 3);
// block_4_0 + block_4_1 in one launch (conv_b4_fused.h): x_in = the padded 16-bit planes B4_* below (x_plane dwords per plane)
// flags: bit 0 walk the tiles from the end, bit 4 plain tile order, bit 5 the 7 x 32 tiles of rounds 2 - 3, bit 6 write the bordered B42_* layout (fp16-plane mode)
// warp != nullptr (fp16-plane mode, u8 images, 4-byte aligned): the kernel samples cat(img1, warp(img2, H)) itself and x_in is not read (conv_b4_fused.h WARPIN, round 6)
struct B4Warp { const uint8_t* img1; const uint8_t* img2; const float* H; };
inline bool block4_warp_in_supported(const void* img1, const void* img2, int pix_fmt_is_u8, int n_planes) {
    return pix_fmt_is_u8 && n_planes == 2 && ((((uintptr_t)img1 | (uintptr_t)img2) & 3) == 0);
}
hipError_t launch_block4_fused(const void* x_in, size_t x_plane, const void* w0frag, const float* bias0, const void* w1frag, const float* bias1,
                               uint16_t* out16, size_t o_plane, int batch, hipStream_t s, int flags = 0 /* bit 0: reverse tile walk */, int n_planes = 3,
                               const B4Warp* warp = nullptr);
// block_3_0 + block_3_1 in one launch (conv_b3_fused.h), fp16-plane mode (n_planes == 2) only: x_in fp32 NHWC [B][112][160][2] (the block's prep output),
// w0frag [7][2][64] x 16 B, w1frag [2][13][2][64] x 16 B (packed by hnet_create), out16 [2][B][56][80][32]
hipError_t launch_block3_fused(const float* x_in, const void* w0frag, const float* bias0, const void* w1frag, const float* bias1, uint16_t* out16,
                               size_t o_plane, int batch, hipStream_t s, int n_planes);
// block_4_2 + block_4_3 in one launch (conv_b42_fused.h), fp16-plane mode only: in16 = block_4_1's planes [2][B][112][160][16],
// w2frag [2][5][2][64] x 16 B, w3frag [4][9][2][64] x 16 B (packed by hnet_create), out16 [2][B][28][40][64]
hipError_t launch_block42_fused(const uint16_t* in16, size_t i_plane, const void* w2frag, const float* bias2, const void* w3frag, const float* bias3,
                                uint16_t* out16, size_t o_plane, int batch, hipStream_t s, int n_planes);
// dynamic-LDS limits of the kernels that use more than 64 KB; once per device (hnet_create)
hipError_t conv_kernels_init_device();
hipError_t launch_heads_fc1_s3(const float* feat, int batch, int n_local, int s_begin, float p, uint64_t mc_seed,
                               uint64_t pair_seq0, const uint16_t* w1planes, const float* b1, float* hidden,
                               uint16_t* feat16, size_t f_plane, uint8_t* mask, hipStream_t s,
                               float* ws = nullptr, size_t ws_floats = 0, const uint64_t* seq_dev = nullptr, int n_planes = 3, int tile = 0,
                               LatIO* lat = nullptr /* lat->heads_one_launch, fp16-plane mode, batch <= 8, n_local <= 64: the one-launch kernel of heads_lat.h */);
hipError_t launch_nchw_f32_to_nhwc_s3(const float* in, uint16_t* out, size_t o_plane, int batch, int c, int h, int w, hipStream_t s, int n_planes = 3);
hipError_t launch_nhwc_s3_to_nchw_f32(const uint16_t* in, size_t i_plane, float* out, int batch, int c, int h, int w, hipStream_t s, int n_planes = 3);

// first FC of both heads with MC-dropout on the input: feat [B][5120] (NHWC flatten) -> hidden [B*n_local][512]
hipError_t launch_heads_fc1(const float* feat, int batch, int n_local, int s_begin, float p, uint64_t mc_seed,
                            uint64_t pair_seq0, const float* w1packed, const float* b1, float* hidden, hipStream_t s,
                            float* ws = nullptr, size_t ws_floats = 0, const uint64_t* seq_dev = nullptr);

// Input of the fused block-4 kernel (conv_b4_fused.h, DMA staging): cat(img1, warp(img2, H)) as bf16 planes with a zero border,
// [plane][B][B4_HP][B4_WP] dwords (lo half = channel 0, hi half = channel 1), pixel (u, v) at row v + B4_PADY, column u + B4_PADX.
// The border (never written after the allocation was zeroed) IS the zero padding of block_4_0, and the patch of every tile
// starts at a column that is a multiple of 4 dwords (16-byte chunks for the LDS-DMA) while staying an odd pixel column.
constexpr int B4_PADX = 5, B4_PADY = 5, B4_WP = 336, B4_HP = 235;
// Input of the fused block_4_2 + block_4_3 kernel (conv_b42_fused.h, round 4: LDS-DMA staging) = output of the fused block-4 kernel in the fp16-plane mode:
// block_4_1's map as fp16 planes with a zero border, [plane][B][B42_HP][B42_WP][16 ch]; pixel (x, y) at row y + B42_PADY, column x + B42_PADX.  The border (never
// written after the allocation was zeroed) IS block_4_2's zero padding: the 19 x 35-pixel patch of every tile lies inside the array, no bounds logic in the copy.
constexpr int B42_PADX = 3, B42_PADY = 3, B42_WP = 164, B42_HP = 115;
constexpr size_t B42_IMG = (size_t)B42_HP * B42_WP;             // pixels per pair and plane
// fp16 / bf16 planes [np][B][h][w][c] (src_plane elements per plane) <-> the same with a border: dst pixel (x, y) of src at (y + pady) wp + x + padx of an hp x wp image.
// to_padded = false copies the interior back.  c a multiple of 8.  (inspection / test paths; the forward writes the padded form directly)
hipError_t launch_s3_repitch(const uint16_t* src, size_t src_plane, uint16_t* dst, size_t dst_plane, int batch, int h, int w, int c, int hp, int wp, int pady,
                             int padx, bool to_padded, hipStream_t s, int n_planes);
// fp16-plane mode, phase 2 of that kernel: filter tap (kh * 5 + kw) of lane group g in MFMA step st; -1 = no tap (zero weights, the group re-reads its neighbour's
// pixels).  Groups (0, 1) and (2, 3) are served together by ds_read_b128: they hold taps of ONE kernel column, whose pixels are whole image rows (a multiple of
// 256 bytes) apart - conflict-free; kernel row 4 has no partner (steps 5, 6).  Shared by the kernel's address setup and hnet_create's fragment packing.
constexpr int b41_tap(int st, int g) {
    if (st < 5) return g * 5 + st;                               // (kh, kw) = (g, st)
    if (st == 5) return g == 0 ? 20 : g == 2 ? 21 : -1;          // (4, 0) | - | (4, 1) | -
    return g == 0 ? 22 : g == 2 ? 23 : g == 3 ? 24 : -1;         // (4, 2) | - | (4, 3) | (4, 4)
}

// cat(img1, warp(img2,H)) -> AvgPool(k) -> NHWC [B][224/k][320/k][2]; H == nullptr: no warp
// out_s3 != nullptr (k = 1 only): write the padded bf16 planes above instead (s3_plane = dwords per plane)
//   exact = true: sampling positions bit-identical to grid_sample (IEEE divisions, the normalise / un-normalise round trip of warp.py:70);
//   false (the library default, HNET_WARP_EXACT=0): shared reciprocal + Newton step, no round trip (positions within 6e-5 px; kernels.hip)
hipError_t launch_prep(const void* img1, const void* img2, int pix_fmt, const float* H, int k, float* out,
                       int batch, hipStream_t s, uint32_t* out_s3 = nullptr, size_t s3_plane = 0, int n_planes = 3, bool exact = true);
// Small-batch form of launch_prep (latency path, batch <= 8): the block-tail launch (Linear(5120,8) + DLT + composition, launch_block_fc_dlt)
// or the prior's DLT (launch_prior_dlt) disappears from the dependent chain - every workgroup of the tiled warp kernel recomputes its
// pair's homography with the same instructions in the same order (bit-identical), the first workgroup of a pair stores it to H_out.
struct FcArgs {
    const float* feat;      // [B][5120] NHWC-flattened trunk output of the previous block, or nullptr
    const float* wfc;       // [8][5120], same order
    const float* bfc;       // [8]
    const float* H_in;      // [B][9] homography so far (nullptr: identity)
    const float* prior;     // [B][8] corner-offset prior: H = DLT(p4 + prior); used when feat == nullptr
    float* H_out;           // [B][9]: written by the first workgroup of each pair (a buffer other than H_in)
    // round 5 (block 4's launch on the latency path): mask_blocks surplus workgroups at the end of the grid draw the keep bits of the heads' first dropout beside
    // the warp (heads_mask.h; they depend on the seeds only) - mask == nullptr: none
    uint8_t* mask;          // [B][n_local][2][640]
    int mask_blocks, n_local, s_begin;
    uint32_t thr;           // hnet_drop_threshold(p)
    uint64_t mc_seed, pair_seq0;
    const uint64_t* seq_dev;
    // round 6: the FC as 32 partial sums per pair, written by the one-XCD tail chain of the previous block (chain_lat.h): [B][32][8]; feat is then not read
    const float* fc_part;
};
bool prep_fc_supported(const void* img1, const void* img2, int k, bool has_out_s3);
hipError_t launch_prep_fc(const void* img1, const void* img2, int pix_fmt, const FcArgs& fc, int k, float* out, int batch, hipStream_t s,
                          uint32_t* out_s3, size_t s3_plane, int n_planes, bool exact);
hipError_t launch_f32_nhwc_to_s3pad(const float* x, uint32_t* out, size_t s3_plane, int batch, int n_planes, hipStream_t s);
hipError_t launch_s3pad_to_f32_nhwc(const uint32_t* in, size_t s3_plane, float* x, int batch, int n_planes, hipStream_t s);

// |warp(img2,H) - img1| * 255 -> float [B][224][320]  (and optional u8 clamp copy)
hipError_t launch_undistort(const uint8_t* raw, int rows, int cols, int stride, const float* map_x, const float* map_y, uint8_t* out,
                            hipStream_t s);
hipError_t launch_errmap(const void* img1, const void* img2, int pix_fmt, const float* H, float* out,
                         uint8_t* out_u8, int batch, hipStream_t s);

// plain warp of a float image (operator-level entry point)
hipError_t launch_warp_f32(const float* img, const float* H, float* out, int batch, hipStream_t s);

// H[b] = DLT(p4 + prior[b])
hipError_t launch_prior_dlt(const float* prior, float* H, int batch, hipStream_t s);
// H[b] = DLT(dst[b])
hipError_t launch_dlt(const float* dst, float* H, int n, hipStream_t s);

// fc (5120 -> 8) + DLT + compose: H_out = (H_in ? H_in : I) * DLT(p4 + fc)
hipError_t launch_block_fc_dlt(const float* feat, const float* wfc, const float* bfc, const float* H_in,
                               float* H_out, int batch, hipStream_t s);

// second FC of both heads (Dropout -> Linear(256,8)) -> per-sample outputs mean_s / logvar_s [B][n_local][8]
//   hidden [B*n_local][512]
hipError_t launch_heads_fc2(const float* hidden, int batch, int n_local, int s_begin, float p, uint64_t mc_seed,
                            uint64_t pair_seq0, const float* w2, const float* b2, float* mean_s, float* logvar_s,
                            hipStream_t s, const uint64_t* seq_dev = nullptr, uint32_t* flag = nullptr /* bit 0 is ORed when an output is not finite */);

// ensemble/transfer from gathered per-sample outputs [B][n][8]
hipError_t launch_mc_finish(const float* mean_s, const float* logvar_s, int n, const float* H1, int batch,
                            float* mean, float* cov, float* Htot, hipStream_t s, uint32_t* flag = nullptr,
                            int mean_stride = 8, int cov_stride = 64 /* floats between consecutive pairs: 72 / 72 with cov = mean + 8 for the packed [B][72] record */,
                            int n_local = 0, size_t rank_stride = 0 /* != 0: the samples as an all-gather leaves them - rank r's [B][n_local][8] block starts r * rank_stride floats in */);

// latency path (n_local <= 64): both launches above in one, one 1024-thread workgroup per pair; bit-identical results
constexpr int HEADS_FC2_FINISH_MAX_N = 64;
hipError_t launch_heads_fc2_finish(const float* hidden, int batch, int n_local, int s_begin, float p, uint64_t mc_seed, uint64_t pair_seq0, const float* w2,
                                   const float* b2, const float* H1, float* mean, float* cov, float* Htot, hipStream_t s,
                                   const uint64_t* seq_dev = nullptr, uint32_t* flag = nullptr, int mean_stride = 8, int cov_stride = 64);

// layout helpers for the operator-level entry points
hipError_t launch_nchw_to_nhwc(const float* in, float* out, int batch, int c, int h, int w, hipStream_t s);
hipError_t launch_nhwc_to_nchw(const float* in, float* out, int batch, int c, int h, int w, hipStream_t s);

}  // namespace hnet
