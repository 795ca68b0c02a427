// kernels_chain.hip — the one-XCD tail chains of the latency path (chain_lat.h): instantiations, launcher and the host-side fragment packer.
// Its own translation unit: the build stays parallel and the chain can be rebuilt in seconds.
#include "chain_lat.h"
#include "kernels.h"

#include <vector>

namespace hnet {

namespace {

template <class C>
void pack_layer(const float* w, std::vector<uint16_t>& out) {
    // [tile][step][plane][lane group g][row i][8 halves]: lane (i, g) of the A operand holds K = 8 g .. 8 g + 7 of step s for output channel tile NCH + i;
    // step s = (tap t = s / CPS, 32-channel chunk c = s % CPS); planes = the two-plane weight split of the implicit-GEMM kernels (w = W0 + W1 / 4096)
    out.assign(C::WFRAG_HALVES, 0);
    for (int tile = 0; tile < C::NTILES; tile++)
        for (int s = 0; s < C::NSTEP; s++) {
            const int t = s / C::CPS, c32 = s % C::CPS, kh = t / C::KS, kw = t % C::KS;
            for (int g = 0; g < 4; g++)
                for (int i = 0; i < C::NCH; i++)
                    for (int e = 0; e < 8; e++) {
                        const int co = tile * C::NCH + i, ci = c32 * 32 + 8 * g + e;
                        uint16_t a0, a1;
                        split2h(w[(((size_t)co * C::CIN + ci) * C::KS + kh) * C::KS + kw], a0, a1);
                        const size_t base = ((size_t)(tile * C::NSTEP + s) * 2) * (C::NCH * 32);
                        out[base + ((size_t)g * C::NCH + i) * 8 + e] = a0;
                        out[base + (size_t)C::NCH * 32 + ((size_t)g * C::NCH + i) * 8 + e] = a1;
                    }
        }
}

template <class C0, class C1, class C2>
hipError_t run_chain(const ChainArgs& a, uint32_t* sync, uint32_t* next_sync, int batch, hipStream_t s, int grid) {
    hipLaunchKernelGGL((tail_chain_kernel<C0, C1, C2>), dim3((unsigned)grid), dim3(CH_NT), CH_LDS_BYTES, s, a, sync, next_sync, batch);
    return hipGetLastError();
}

}  // namespace

bool chain_layer(int layer) { return layer == 1 || layer == 2 || layer == 4 || layer == 5 || layer == 6 || layer == 10 || layer == 11 || layer == 12 || layer == 17 || layer == 18 || layer == 19; }

bool chain_pack_weights(int layer, const float* w, std::vector<uint16_t>& out) {
    switch (layer) {
        case 1: pack_layer<ChainL12>(w, out); return true;
        case 2: pack_layer<ChainL13>(w, out); return true;
        case 4: pack_layer<ChainL22>(w, out); return true;
        case 10: case 17: pack_layer<ChainLx4>(w, out); return true;
        case 5: case 11: case 18: pack_layer<ChainLx5>(w, out); return true;
        case 6: case 12: case 19: pack_layer<ChainLx6>(w, out); return true;
    }
    return false;
}

hipError_t chain_init_device() {
    hipError_t e = hipFuncSetAttribute((const void*)tail_chain_kernel<ChainL12, ChainL13, ChainNone>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tail_chain_kernel<ChainL22, ChainLx5, ChainLx6>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tail_chain_kernel<ChainLx4, ChainLx5, ChainLx6>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BYTES);
    return e;
}

// block 1..4 -> its tail chain; d_args = the block's ChainArgs (L = its tail layers in order; block 1: two, L[2] unused)
hipError_t launch_tail_chain(int block, const ChainArgs& d_args, uint32_t* sync, uint32_t* next_sync, int batch, hipStream_t s, int grid) {
    if (batch < 1 || batch > CH_MAX_PAIRS || !sync || !next_sync || sync == next_sync || grid < 1) return hipErrorInvalidValue;
    switch (block) {
        case 1: return run_chain<ChainL12, ChainL13, ChainNone>(d_args, sync, next_sync, batch, s, grid);
        case 2: return run_chain<ChainL22, ChainLx5, ChainLx6>(d_args, sync, next_sync, batch, s, grid);
        case 3: case 4: return run_chain<ChainLx4, ChainLx5, ChainLx6>(d_args, sync, next_sync, batch, s, grid);
    }
    return hipErrorInvalidValue;
}

}  // namespace hnet
