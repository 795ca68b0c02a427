// chain_args.h — host / device interface of the one-XCD tail chains (chain_lat.h, kernels_chain.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace hnet {

constexpr int CH_MAX_PAIRS = 8;
// One counter AREA per chain launch of a forward (uint32 words; all zero when its launch starts: workgroup 0 of the PREVIOUS chain launch of the stream zeroes it):
//   [0..7] claim of pair p (0 free, xcc + 1 owner) - agent-scope compare-and-swap, the only words that XCDs share;
//   from word 32 on, 32 words (one 128-byte line) per pair, touched by ONE XCD each (L2-local atomics, sc1 polls): [xcc] the item counter of the pair as drawn by
//   XCD xcc (three 10-bit fields, one per layer; only the owner's is ever read), [8 + layer] items of the layer that are complete
constexpr int CH_SYNC_WORDS = 32 + 32 * CH_MAX_PAIRS;
constexpr int CH_AREAS = 4;
constexpr uint32_t CH_SPIN_LIMIT = 1u << 20;
constexpr int CH_FLAG_TIMEOUT = 2;                 // bit ORed into the context's flag word when a bounded spin gave up (results of that forward are invalid)

struct ChainLayer {
    const uint16_t* in;      // fp16 planes [2][MB][HI][WI][CIN]
    size_t in_plane;         // elements per plane
    const uint16_t* wfrag;   // packed fragments (chain_pack_weights)
    const float* bias;       // [COUT]
    uint16_t* out16;         // fp16 planes [2][MB][HO][WO][COUT], or
    size_t out_plane;
    float* out32;            // fp32 [MB][HO WO][COUT] (last layer of a block: the FC's input)
};
constexpr int CH_FC_ITEMS = 32;                    // items of a chain's last layer (ChainL13 / ChainLx6): the partial sums per pair
struct ChainArgs {
    ChainLayer L[3];
    uint32_t* flag;          // the context's flag word (hnet_overflow_flag)
    // blocks 1 - 3 (nullptr for block 4): the block-tail Linear(5120, 8) as PARTIAL sums per item of the LAST layer - the item's 8 channels x HO WO pixels times their
    // slice of the FC weights - summed over the items by the next warp + pool launch (kernels.hip fc_part_dlt_block) instead of every workgroup of that launch
    // re-reading 184 KB of features and weights
    const float* fcw;        // [8][5120], NHWC-flatten order (hnet_create's fc_w)
    float* fc_part;          // [MB][CH_FC_ITEMS][8]
};

bool chain_layer(int layer);                                                       // a conv layer that belongs to a tail chain (kConvs index)
bool chain_pack_weights(int layer, const float* w, std::vector<uint16_t>& out);      // fp32 [Cout][Cin][KS][KS] -> the chain's MFMA fragments (two fp16 planes)
hipError_t chain_init_device();                                                    // dynamic-LDS limit of the chain kernels; once per device
// block 1 .. 4 -> its tail chain; args = the block's ChainArgs (L = its tail layers in order; block 1: two), batch <= CH_MAX_PAIRS
// sync: the launch's counter area (zero); next_sync: the area of the next chain launch of this stream (zeroed by this launch)
// grid: workgroups (256 = one per CU; the tests launch 8 and 3: fewer resident workgroups than items, XCDs without workgroups - any grid >= 1 computes the same bits)
hipError_t launch_tail_chain(int block, const ChainArgs& args, uint32_t* sync, uint32_t* next_sync, int batch, hipStream_t s, int grid = 256);

}  // namespace hnet
