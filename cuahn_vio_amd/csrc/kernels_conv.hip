// kernels_conv.hip — instantiations and dispatch of the implicit-GEMM kernel (igemm.h) for the 20
// convolution layers of HomographyNet (reference model_to_trace.py:88-113, :210-216) and the heads' first FC.
#include "igemm.h"
#include "kernels.h"

namespace hnet {

// execution order; seg = K-segment of the packed weights: largest of {32,16,8} dividing KS*CIN, else 16 (padded)
const ConvDesc kConvs[20] = {
    {"block_1_1", 2, 128, 7, 2, 16, 1}, {"block_1_2", 128, 128, 5, 2, 32, 1}, {"block_1_3", 128, 256, 3, 2, 32, 1},
    {"block_2_1", 2, 64, 7, 2, 16, 2},  {"block_2_2", 64, 128, 5, 2, 32, 2},  {"block_2_3", 128, 256, 3, 2, 32, 2},
    {"block_2_4", 256, 256, 3, 2, 32, 2},
    {"block_3_0", 2, 16, 7, 1, 16, 3},  {"block_3_1", 16, 32, 5, 2, 16, 3},   {"block_3_2", 32, 64, 3, 2, 32, 3},
    {"block_3_3", 64, 128, 3, 2, 32, 3}, {"block_3_4", 128, 256, 3, 2, 32, 3}, {"block_3_5", 256, 256, 3, 2, 32, 3},
    {"block_4_0", 2, 8, 7, 1, 16, 4},   {"block_4_1", 8, 16, 5, 2, 8, 4},     {"block_4_2", 16, 32, 3, 2, 16, 4},
    {"block_4_3", 32, 64, 3, 2, 32, 4}, {"block_4_4", 64, 128, 3, 2, 32, 4},  {"block_4_5", 128, 256, 3, 2, 32, 4},
    {"block_4_6", 256, 256, 3, 2, 32, 4},
};

int conv_padded_k(int layer) {
    const ConvDesc& d = kConvs[layer];
    const int rl = d.ks * d.cin;
    const int spr = (rl + d.seg - 1) / d.seg;
    return d.ks * spr * d.seg;
}

template <class L, int BM, int BN, int WGM, int MF>
static hipError_t run(const IgemmParams& p, hipStream_t s) {
    dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN);
    hipLaunchKernelGGL((igemm_kernel<L, BM, BN, WGM, MF>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// tile choice: Cout <= 16 -> 16x16x4 MFMA, BN = 16;  Cout = 32 -> BN = 32;  else BN = 64.
// BM = 128 when that still gives >= 2 workgroups per CU worth of tiles, else 64.
template <int CIN, int KS, int STRIDE, int SEG, int COUT>
static hipError_t run_conv(const IgemmParams& p, hipStream_t s) {
    typedef ConvLoader<CIN, KS, STRIDE, SEG> L;
    if constexpr (COUT <= 16) {
        return run<L, 128, 16, 4, 16>(p, s);
    } else if constexpr (COUT == 32) {
        return run<L, 128, 32, 4, 32>(p, s);
    } else {
        const long tiles128 = (long)((p.M + 127) / 128) * (COUT / 64);
        if (tiles128 >= 512) return run<L, 128, 64, 2, 32>(p, s);
        return run<L, 64, 64, 2, 32>(p, s);
    }
}

hipError_t launch_conv(int layer, const float* in, int batch, int h, int w, const float* wpacked,
                       const float* bias, float* out, hipStream_t s) {
    if (layer < 0 || layer >= 20) return hipErrorInvalidValue;
    const ConvDesc& d = kConvs[layer];
    IgemmParams p = {};
    p.A = in; p.Wp = wpacked; p.bias = bias; p.out = out;
    p.H = h; p.W = w;
    p.Ho = conv_out_dim(h, d.ks, d.stride);
    p.Wo = conv_out_dim(w, d.ks, d.stride);
    p.M = batch * p.Ho * p.Wo;
    p.N = d.cout;
    p.Kp = conv_padded_k(layer);
    switch (layer) {
        case 0:  return run_conv<2, 7, 2, 16, 128>(p, s);
        case 1:  return run_conv<128, 5, 2, 32, 128>(p, s);
        case 2: case 5: case 11: case 18: return run_conv<128, 3, 2, 32, 256>(p, s);
        case 3:  return run_conv<2, 7, 2, 16, 64>(p, s);
        case 4:  return run_conv<64, 5, 2, 32, 128>(p, s);
        case 6: case 12: case 19: return run_conv<256, 3, 2, 32, 256>(p, s);
        case 7:  return run_conv<2, 7, 1, 16, 16>(p, s);
        case 8:  return run_conv<16, 5, 2, 16, 32>(p, s);
        case 9: case 16: return run_conv<32, 3, 2, 32, 64>(p, s);
        case 10: case 17: return run_conv<64, 3, 2, 32, 128>(p, s);
        case 13: return run_conv<2, 7, 1, 16, 8>(p, s);
        case 14: return run_conv<8, 5, 2, 8, 16>(p, s);
        case 15: return run_conv<16, 3, 2, 16, 32>(p, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_heads_fc1(const float* feat, int batch, int n_local, int s_begin, float p_drop, uint64_t mc_seed,
                            uint64_t pair_seq0, const float* w1packed, const float* b1, float* hidden, hipStream_t s) {
    IgemmParams p = {};
    p.A = feat; p.Wp = w1packed; p.bias = b1; p.out = hidden;
    p.M = batch * n_local;
    p.N = 512;
    p.Kp = 5120;
    p.n_local = n_local;
    p.s_begin = s_begin;
    p.thr = hnet_drop_threshold(p_drop);
    p.scale = 1.0f / (1.0f - p_drop);
    p.mc_seed = mc_seed;
    p.pair_seq0 = pair_seq0;
    const long tiles128 = (long)((p.M + 127) / 128) * 8;
    if (tiles128 >= 512) return run<HeadLoader, 128, 64, 2, 32>(p, s);
    return run<HeadLoader, 64, 64, 2, 32>(p, s);
}

}  // namespace hnet
