// heads_mask.h — the keep bits of the heads' first dropout (reference: nn.Dropout in front of Linear(5120, 256), model_to_trace.py:222-223, 229-230; the mask
// function is include/hnet_rng.h), row-major layout [B][n_local][2 heads][640 bytes]: bit i of byte j = NHWC element 8 j + i of the flattened trunk output.
// One virtual block = 256 threads x 4 bytes (32 draws of one row, one 32-bit store per thread).  Shared by heads_prep_kernel (igemm_s3.h) and - round 5, latency
// path - by the block-4 prep launch, whose surplus workgroups draw the bits beside the warp (kernels.hip prep_warp_tiled_kernel: the bits depend on the seeds
// only, so they leave the dependent chain of a batch-1 forward without a second stream; a forked graph branch was measured 30 us SLOWER than the chain).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hnet_rng.h"

namespace hnet {

// vb: virtual block index; pre_row: three words of LDS.  Every thread of the 256-thread block must call it (one workgroup barrier inside).
// A block's 1024 bytes span at most three rows (b, sample, head) of 640 bytes: the row's hash prefix (four hnet_mix32 and, with a run-time n_local, an integer
// division) is formed by three threads and shared through LDS, and all index arithmetic is 32-bit.
__device__ __forceinline__ void heads_mask_block(uint32_t vb, int batch, int n_local, int s_begin, uint32_t thr, uint64_t mc_seed, uint64_t pair_seq,
                                                 uint8_t* __restrict__ mask, uint32_t* pre_row) {
    const size_t nmask = (size_t)batch * n_local * 2 * 640;
    const size_t i = (size_t)vb * 256 + threadIdx.x;
    const uint32_t i0 = vb * 1024u, row0 = i0 / 640u, rem0 = i0 - row0 * 640u;
    if (threadIdx.x < 3) {
        const uint32_t row = row0 + threadIdx.x;                      // rows beyond the end are never read
        const uint32_t head = row & 1u, t = row >> 1;
        const uint32_t b = t / (uint32_t)n_local, sm = t - b * (uint32_t)n_local;
        pre_row[threadIdx.x] = hnet_mask_prefix(hnet_pair_key(mc_seed, pair_seq + (uint64_t)b), 2u * head, (uint32_t)s_begin + sm);
    }
    __syncthreads();
    if (4 * i >= nmask) return;                                       // nmask is a multiple of 640: the four bytes are all in or all out
    const uint32_t off = rem0 + 4u * threadIdx.x, wrap = (off >= 640u ? 1u : 0u) + (off >= 1280u ? 1u : 0u);
    const int chunk = (int)(off - 640u * wrap);                       // multiple of 4: the four bytes lie in one row and in one pixel's channel run
    const uint32_t pre = pre_row[wrap];
    const int k0 = chunk * 8, pix = k0 >> 8, c0 = k0 & 255;
    // hnet_mask_keep(pre, element, thr) for the 32 elements (c0 + e) * 20 + pix: element * 0xc2b2ae35 + 0x27d4eb2f (hnet_rng.h, hnet_mask_bits) advances by the
    // constant 20 * 0xc2b2ae35 (mod 2^32) from one to the next - one quarter-rate integer multiply per thread instead of one per element: same bits
    uint32_t em = (uint32_t)(c0 * 20 + pix) * 0xc2b2ae35U + 0x27d4eb2fU;
    const uint32_t thr8 = thr << 8;                                   // (x >> 8) >= thr  <=>  x >= thr << 8 for thr < 2^24; thr = 2^24 (p = 1) keeps nothing
    uint32_t bits = 0;
#pragma unroll
    for (int e = 0; e < 32; e++) {
        const uint32_t x = hnet_mix32(pre ^ em);
        bits |= (thr < (1u << 24) && x >= thr8) ? (1u << e) : 0u;
        em += 20U * 0xc2b2ae35U;
    }
    reinterpret_cast<uint32_t*>(mask)[i] = bits;
}
inline unsigned heads_mask_blocks(int batch, int n_local) { return (unsigned)(((size_t)batch * n_local * 2 * 160 + 255) / 256); }

}  // namespace hnet
