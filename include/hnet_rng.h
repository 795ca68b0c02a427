/* hnet_rng.h — the MC-dropout mask function of the hnet C-ABI (part of the interface contract).
 *
 * The reference draws its four dropout masks per forward from PyTorch's global generator
 * (nn.Dropout forced to train mode, reference trace_pytorch_model/model_to_trace.py:222-235,266-268;
 * `noise ~ Bernoulli(1-p); noise /= (1-p); out = x * noise`).  That stream is not reproducible outside
 * PyTorch, so this interface defines the mask as a pure function of
 *   (mc_seed, pair_seq, stream, sample, element)
 * which the HIP kernels, the CPU oracle and the golden-vector generator all evaluate identically.
 * With the mask given, the forward is deterministic and comparable to the reference model run with
 * the same mask injected in place of nn.Dropout (tools/gen_golden.py).
 *
 * stream: 0 = mean head, input of Linear(5120,256)      1 = mean head, input of Linear(256,8)
 *         2 = uncertainty head, input of Linear(5120,256) 3 = uncertainty head, input of Linear(256,8)
 * sample: MC-dropout sample index 0..N-1 (GLOBAL index, so sharding N over ranks does not change masks)
 * element: index into the flattened activation (NCHW flatten order of the reference, model_to_trace.py:253)
 */
#ifndef HNET_RNG_H
#define HNET_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define HNET_RNG_FN __host__ __device__ static inline
#else
#define HNET_RNG_FN static inline
#endif

HNET_RNG_FN uint32_t hnet_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

/* 64-bit key of one frame pair */
HNET_RNG_FN uint64_t hnet_pair_key(uint64_t mc_seed, uint64_t pair_seq) {
    return mc_seed ^ (pair_seq * 0x9E3779B97F4A7C15ULL);
}

/* per (pair, stream, sample) prefix; hoist out of element loops */
HNET_RNG_FN uint32_t hnet_mask_prefix(uint64_t pair_key, uint32_t stream, uint32_t sample) {
    uint32_t klo = (uint32_t)pair_key, khi = (uint32_t)(pair_key >> 32);
    uint32_t h0 = hnet_mix32(klo ^ hnet_mix32(khi ^ 0x5bd1e995U));
    return hnet_mix32(h0 + stream * 0x9e3779b9U + sample * 0x85ebca6bU + 1U);
}

/* 24-bit threshold for drop probability p: element is KEPT iff hnet_mask_bits(...) >= threshold */
HNET_RNG_FN uint32_t hnet_drop_threshold(float p) {
    double t = (double)p * 16777216.0;
    if (t <= 0.0) return 0U;
    if (t >= 16777216.0) return 16777216U;
    return (uint32_t)t;
}

HNET_RNG_FN uint32_t hnet_mask_bits(uint32_t prefix, uint32_t element) {
    return hnet_mix32(prefix ^ (element * 0xc2b2ae35U + 0x27d4eb2fU)) >> 8;
}

HNET_RNG_FN int hnet_mask_keep(uint32_t prefix, uint32_t element, uint32_t threshold) {
    return hnet_mask_bits(prefix, element) >= threshold;
}

#endif /* HNET_RNG_H */
