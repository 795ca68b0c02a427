// hnet_capi.hip — context, weight packing, forward orchestration and the C ABI of include/hnet.h.
//
// Mirrors the behaviour of the reference runtime class pytorch::HomographyNet
// (cuahn_ros/homography_network/src/HomographyNet.cpp) without libtorch: weights come from an HNETW001
// blob, the forward is a fixed sequence of HIP kernel launches on one stream over persistent buffers.
#include "../../include/hnet.h"
#include "../../include/hnet_rng.h"
#include "geom.h"
#include "kernels.h"
#include "chain_args.h"
#include "s3_format.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// `make ROCTX=1` (-DHNET_ROCTX, links libroctx64): roctx ranges around the forward and around each block, visible in rocprofv3 --marker-trace and in
// the timeline tools - the counterpart of the stopwatches the reference brackets `forward` with (HomographyNet.cpp:178-188).  Off in the default build:
// the hot path makes no call into a tracing library.
#ifdef HNET_ROCTX
#include <roctracer/roctx.h>
struct HnetRange {
    explicit HnetRange(const char* name) { roctxRangePush(name); }
    ~HnetRange() { roctxRangePop(); }
    HnetRange(const HnetRange&) = delete;
};
#define HNET_RANGE(var, name) HnetRange var(name)
#else
#define HNET_RANGE(var, name) do { } while (0)
#endif

using namespace hnet;

namespace {

struct Tensor { std::vector<uint32_t> dims; const float* data; size_t count; };

struct Blob {
    std::vector<std::pair<std::string, Tensor>> t;
    // the tensor must have exactly the reference's shape (state_dict of model_to_trace.py:88-115, :210-235), not just its size
    const Tensor* find(const std::string& n, std::initializer_list<uint32_t> shape) const {
        for (auto& e : t)
            if (e.first == n) return e.second.dims == std::vector<uint32_t>(shape) ? &e.second : nullptr;
        return nullptr;
    }
};

bool parse_blob(const uint8_t* p, size_t len, Blob& out) {
    if (len < 12 || memcmp(p, "HNETW001", 8) != 0) return false;
    uint32_t n; memcpy(&n, p + 8, 4);
    if (n > 1024) return false;
    size_t pos = 12;
    struct Ent { std::string name; std::vector<uint32_t> dims; uint64_t off; size_t count; };
    std::vector<Ent> ents;
    for (uint32_t i = 0; i < n; i++) {
        if (pos + 4 > len) return false;
        uint32_t ln; memcpy(&ln, p + pos, 4); pos += 4;
        if (ln > 512 || pos + ln + 4 > len) return false;
        Ent e; e.name.assign((const char*)p + pos, ln); pos += ln;
        uint32_t nd; memcpy(&nd, p + pos, 4); pos += 4;
        if (nd > 8 || pos + 4 * nd + 8 > len) return false;
        e.count = 1;
        for (uint32_t d = 0; d < nd; d++) {
            uint32_t v; memcpy(&v, p + pos, 4); pos += 4;
            e.dims.push_back(v);
            if (v != 0 && e.count > (len / 4) / v) return false;     // the product cannot exceed the file (no 64-bit wrap)
            e.count *= v;
        }
        memcpy(&e.off, p + pos, 8); pos += 8;
        if (e.off % 4) return false;
        ents.push_back(e);
    }
    const size_t data0 = (pos + 63) / 64 * 64;
    if (data0 > len) return false;
    const size_t room = len - data0;                                  // bytes of the data section
    for (auto& e : ents) {                                            // offsets come from the file: every check without overflow
        if (e.off > room || e.count > (room - (size_t)e.off) / 4) return false;
        out.t.push_back({e.name, Tensor{e.dims, (const float*)(p + data0 + e.off), e.count}});
    }
    return true;
}

struct Stage { std::string name; double flops_per_pair; int kernels = 1; };      // kernels: what the launch of the last forward consisted of (hnet_stage_kernels)

}  // namespace

struct hnet_ctx {
    hnet_config cfg;
    hipStream_t stream = nullptr;
    std::string err;
    bool owns_stream = true;           // false: a member of an hnet_group (the group owns the streams)
    std::vector<uint8_t> blob_copy;    // HNET_PREC_F16X2 only: the weight blob, kept so that an activation overflow can demote the context to HNET_PREC_BF16X3
    // weights (device)
    float* conv_w[20] = {};
    float* conv_b[20] = {};
    float* fc_w[3] = {};
    float* fc_b[3] = {};
    float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
    // activations (device), sized for cfg.max_batch
    // matrix-core modes (every precision but HNET_PREC_FP32): activations of the layers feeding a conv are 16-bit planes (s3_format.h)
    bool s3 = false;
    uint16_t* conv_w16[20] = {};       // [3][Cout][Kp] 16-bit weight planes of the Cin >= 8 layers (fp16 in HNET_PREC_F16X2, bf16 in the bf16 modes; s3_format.h)
    uint16_t* conv_wfrag[20] = {};     // fp16-plane mode, igemm_region.h layers (block_1_2, block_1_3, block_2_4 / 3_5 / 4_6): the weights as MFMA fragments in consumption order
    uint16_t* act16[20] = {};          // [planes][max_batch][Ho][Wo][Cout] 16-bit activation planes: two fp16 planes in the default mode, three / one bf16 planes in HNET_PREC_BF16X3 / _BF16
    bool fuse_b4 = false;              // block_4_0 + block_4_1 in one kernel (conv_b4_fused.h), every matrix-core mode
    int b4_flags = 0;                  // bit 0: the fused kernel walks its tiles from the end of the batch (hnet_op_block4_fused `reverse`, tests)
    uint32_t* x16_b4 = nullptr;        // block-4 input as padded 16-bit planes (fp16 / bf16 by mode) [planes][max_batch][B4_HP][B4_WP] dwords (DMA-staged fused kernel, kernels.h)
    size_t x16_plane = 0;              // dwords per plane
    int n_planes = 3;                  // 16-bit planes the matrix-core layers read and write = their arithmetic mode: 3 = split-bf16 (fp32-grade), 1 = plain bf16 (HNET_PREC_BF16), 2 = fp16 planes (HNET_PREC_F16X2, fp32-grade)
    uint16_t* patch_frag[20] = {};     // conv_patch_s2.h weight fragments of block_3_1 / block_4_2: [2][NSTEP][3][64] x 16 B
    bool use_patch = false;
    int patch_rb5 = 5;                 // HNET_PATCH_RB5: region rows per batch of staging loads in the 5x5 patch kernel (1 / 2 / as many as fit: 5 in the fp16 mode, 3 in split-bf16)
    int s3_tile = 0;                   // HNET_S3_TILE: tile-shape experiments of the implicit-GEMM layers (s3_dispatch.h), 0 = measured defaults
    bool patch_b128 = true;            // block_3_1 / block_4_2 read their fragments with ds_read_b128 from the interleaved layout (HNET_PATCH_B128=0: two ds_read_b64, half-major layout)
    bool fuse_b3 = false;              // block_3_0 + block_3_1 in one kernel (conv_b3_fused.h): fp16-plane mode, HNET_FUSE_B3=0 switches back
    // its weights: block_3_0 as the three-plane fragments b30_frag of conv_first.h, block_3_1 as [2][13][2][64] x 16 B
    uint16_t* b3f_w1 = nullptr;
    bool a14_pad = false;              // block_4_1's output (act16[14]) in the bordered layout of kernels.h B42_* (fused block-4 kernel -> LDS-DMA of the fused block_4_2 + 4_3 kernel)
    bool fuse_b42 = false;             // block_4_2 + block_4_3 in one kernel (conv_b42_fused.h): fp16-plane mode, HNET_FUSE_B42=0 switches back
    uint16_t* b42_w2 = nullptr;        // its weights: [2][5][2][64] x 16 B and [4][9][2][64] x 16 B fragments
    uint16_t* b42_w3 = nullptr;
    bool fuse_small = true;            // batch <= 8 (latency path): block-tail FC + DLT inside the next block's prep kernel, heads_fc2 + mc_finish in one launch (HNET_FUSE_SMALL=0: the separate launches; bit-identical)
    float* Hm2 = nullptr;              // second homography buffer of that path (a prep workgroup stores H while others still read the previous one)
    const float* H_last = nullptr;     // where the last forward left H_part1 (Hm or Hm2)
    bool warp_exact = false;           // HNET_WARP_EXACT=1: the prep kernels keep grid_sample's sampling positions bit for bit (kernels.hip, A/B switch); default: the fast sampler
    bool use_patch32 = true;           // block_3_2 / block_4_3 through conv_patch32_s2_kernel (HNET_PATCH32=0: implicit GEMM)
    uint16_t* b30_frag = nullptr;      // block_3_0 weights as 32x32x16 fragments of the pixel-pair GEMM [7][3][64] x 16 B (conv_first.h)
    bool b30_s3 = true;
    uint16_t* s2_frag[4] = {};         // block_1_1 / block_2_1 (layers 0, 3) weights as 16x16x32 A-fragments [Cout/16][4][3][64] x 16 B (conv7_c2_s2_s3_kernel)
    bool first_s2 = true;              // HNET_FIRST_S2=0: the round-1 fp32-MFMA implicit GEMM for these two layers
    uint16_t* b40_frag = nullptr;      // block_4_0 weights as 16x16x32 B-fragments of the pixel-pair GEMM [4][3][64] x 16 B, + slot [4]: kernel row 6 as 16x16x16 fragments
    uint16_t* b41_frag = nullptr;      // block_4_1 weights as 16x16x32 B-fragments [7][3][64] x 16 B
    uint16_t* w1_16 = nullptr;         // heads Linear(5120,256) x2: [3][512][5120] 16-bit weight planes (fp16 / bf16 by mode)
    uint16_t* feat16 = nullptr;        // [planes][max_batch][5120] 16-bit planes: feat * 1/(1-p), split
    uint8_t* head_mask = nullptr;      // [max_batch][n_local][2][640] keep bits
    size_t act_count[20] = {};         // elements per pair of layer l's output
    float* x_in[4] = {};
    float* act[20] = {};
    int act_c[20], act_h[20], act_w[20];
    float* ws = nullptr;               // split-K partial sums (igemm.h), 64 MB, followed by the SPLITK_TICKETS tile counters of the latency path (kernels.h LatIO)
    size_t ws_floats = 0;
    // round 6: the tail of every block (its last 2 - 3 stride-2 layers) of a batch <= 8 as ONE launch on one XCD (chain_lat.h); fp16-plane mode, variant bit NO_CHAIN = off
    bool use_chain = false;
    float* fc_part = nullptr;          // [CH_MAX_PAIRS][32][8]: the block-tail FC as partial sums per item of a tail chain's last layer (chain_lat.h), read by the next warp + pool launch
    bool b4_in_stale = false;          // the last forward's block 4 sampled its input in-kernel: x16_b4 does not hold it (hnet_debug_layer_output(13) refuses)
    bool warp_in = false;              // batch > 8: block 4's warp + concat sampled inside the block_4_0 + block_4_1 kernel (conv_b4_fused.h WARPIN) - no prep_b4 launch
    int chain_grid = 256;              // workgroups of a chain launch (one per CU; HNET_VARIANT_CHAIN_GRID_8 / _3: the tests' small grids)
    uint16_t* chain_w[20] = {};        // the chain layers' weights as MFMA fragments (chain_pack_weights)
    ChainArgs chain_args[4] = {};      // one argument block per block's chain (passed by value)
    uint32_t* chain_sync = nullptr;    // CH_AREAS counter areas of CH_SYNC_WORDS words (claims, per-pair item / done counters), one per block's chain: the area of a launch is zero
                                       // when it starts - every chain launch zeroes the area of the NEXT chain launch of the forward sequence (stream ordered)
    bool lat_tail = true;              // round 5: split-K tiles of the 4 x 5 layers finished by their last-arriving workgroup, heads FC1 of small batches as one launch (heads_lat.h); variant 30 = off
    float *hidden = nullptr, *Hm = nullptr, *Htot = nullptr, *mean_s = nullptr, *logvar_s = nullptr;
    float *d_mean = nullptr, *d_cov = nullptr, *d_err = nullptr, *d_prior = nullptr;
    uint8_t* d_err_u8 = nullptr;
    void *stage_prev = nullptr, *stage_curr = nullptr;    // batch staging for host-buffer entry points (f32 sized)
    uint8_t* ring[2] = {};                                 // streaming prev / curr
    float* und_map[2] = {};                                // undistortion maps (x, y), 224x320 floats each (hnet_set_camera)
    uint8_t* raw_dev = nullptr;                            // staging of one raw frame
    int raw_rows = 0, raw_cols = 0;
    int curr_slot = 0;
    int img_counter = 0;
    double latest_t = -1.0;
    hnet_ctx* img_src = nullptr;       // hnet_attach_images: frames, counters and the mask sequence number are read from this context (the IEKF's iterative model)
    int n_local = 0, s_begin = 0;
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hnet_timing timing = {};
    std::vector<Stage> stages;
    std::vector<hipEvent_t> prof_ev;   // when non-empty: one event after every stage
    size_t prof_pos = 0;
    int last_batch = 0;
    // hipGraph replay of small-batch forwards (29-45 dependent launches: at batch 1 the host launch cost dominates).
    // The sequence number of the MC-dropout masks lives in device memory (d_seq) and is refreshed by a memcpy node
    // from a pinned host word, so one captured graph serves every call.
    bool graph_zero_copy = false;      // ... whose kernels read {sequence number, prior} from and write {mean, cov, error map, flag} to the pinned host block directly (no memcpy nodes)
    bool use_graph = false;            // hnet_infer replays the forward as one hipGraph (default on; HNET_GRAPH=0: eager launches)
    bool graph_timing = false;         // hnet_time_batch_device too (HNET_GRAPH=1 only: the bare device time is 3 % better eager)
    uint64_t* d_seq = nullptr;
    uint32_t* d_flag = nullptr;        // hnet_overflow_flag: bit 0 = a forward produced a non-finite output since the last poll
    struct Pinned { uint64_t seq; float prior[8]; float mean[8]; float cov[64]; uint32_t flag; uint8_t err[HNET_IMG_ROWS * HNET_IMG_COLS]; };
    Pinned* pinned = nullptr;
    uint8_t* pinned_img[2] = {nullptr, nullptr};         // host staging of the pushed frame, one per ring slot
    hipEvent_t ev_img[2] = {nullptr, nullptr};           // its upload has completed
    hipGraphExec_t g_infer[2] = {nullptr, nullptr};      // hnet_infer, one per ring orientation
    const float* g_infer_H[2] = {nullptr, nullptr};      // where that graph's forward leaves H_part1 (H_last is only written while a forward is ENQUEUED, i.e. at capture time)
    const float* g_batch_H = nullptr;
    struct GraphKey { const void *prev, *curr, *prior, *mean, *cov; int batch, fmt; bool operator==(const GraphKey& o) const {
        return prev == o.prev && curr == o.curr && prior == o.prior && mean == o.mean && cov == o.cov && batch == o.batch && fmt == o.fmt; } };
    GraphKey g_key = {};
    hipGraphExec_t g_batch = nullptr;                     // hnet_time_batch_device on resident buffers (last signature)
};

// N contexts of one configuration on one device for INDEPENDENT steps (a server's batches, a rank's share of a streamed sequence): step k runs on context k mod N,
// each context on its own HIP stream, so that the dependent launch chain of one step runs under the kernels of the others.
struct hnet_group {
    std::vector<hnet_ctx*> ctx;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> ev;        // hnet_group_join: one event per member stream
    int device_id = 0;
    uint64_t next = 0;                 // round-robin position
    std::string err;
};

namespace {

const char* kStatus[] = {"ok", "invalid argument", "bad weights", "device error", "not ready (need two images)",
                         "batch exceeds max_batch", "unsupported"};

int fail(hnet_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(c, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail((c), HNET_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

template <typename T>
hipError_t dalloc(T** p, size_t count) { return hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)); }

// Device temporaries of the operator-level entry points: freed on every return path (HIPCHK returns early).
struct DevTemps {
    std::vector<void*> ptrs;
    template <typename T>
    hipError_t alloc(T** p, size_t count) {
        const hipError_t e = dalloc(p, count);
        if (e == hipSuccess) ptrs.push_back((void*)*p);
        return e;
    }
    ~DevTemps() { for (void* q : ptrs) (void)hipFree(q); }
};

hipError_t upload(float** dst, const std::vector<float>& v) {
    hipError_t e = dalloc(dst, v.size());
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice);
}

// conv weight [Cout][Cin][KS][KS] -> [Cout][KS][SPR*SEG], inner index r = kw*Cin + ci, zero padded (igemm.h)
std::vector<float> pack_conv(const float* w, const ConvDesc& d, int kp) {
    const int rl = d.ks * d.cin, rlp = kp / d.ks;
    std::vector<float> out((size_t)d.cout * kp, 0.0f);
    for (int co = 0; co < d.cout; co++)
        for (int ci = 0; ci < d.cin; ci++)
            for (int kh = 0; kh < d.ks; kh++)
                for (int kw = 0; kw < d.ks; kw++) {
                    const int r = kw * d.cin + ci;
                    (void)rl;
                    out[(size_t)co * kp + kh * rlp + r] = w[(((size_t)co * d.cin + ci) * d.ks + kh) * d.ks + kw];
                }
    return out;
}

// 7x7 / Cin 2 / stride 1 first layers: B-operand fragments of the pixel-pair GEMM of conv_first.h.
// W'[kh][kk = 2*kw' + ci][(dx, co)] = W[co][ci][kh][kw' - dx]  (0 outside 0..6); fragment t of lane l:
//   Cout  8 (16x16x4): t = kh*4 + e,        n = l&15, g = l>>4, kk = 4g + e
//   Cout 16 (32x32x2): t = kh*8 + q*4 + e,  n = l&31, h = l>>5, kk = 8q + 4h + e
std::vector<float> pack_first_weights(const float* w, int cout) {
    const int nfrag = cout == 8 ? 28 : 56;
    std::vector<float> out((size_t)nfrag * 64, 0.0f);
    for (int t = 0; t < nfrag; t++)
        for (int l = 0; l < 64; l++) {
            int kh, kk, n;
            if (cout == 8) { kh = t / 4; kk = 4 * (l >> 4) + (t % 4); n = l & 15; }
            else { kh = t / 8; const int q = (t % 8) / 4, e = t % 4; kk = 8 * q + 4 * (l >> 5) + e; n = l & 31; }
            const int kwp = kk >> 1, ci = kk & 1, dx = n / cout, co = n % cout, kw = kwp - dx;
            if (kw >= 0 && kw < 7) out[(size_t)t * 64 + l] = w[(((size_t)co * 2 + ci) * 7 + kh) * 7 + kw];
        }
    return out;
}

// linear weight [out][5120] with NCHW-flatten input index c*20+pix -> NHWC-flatten index pix*256+c
std::vector<float> permute_fc(const float* w, int n_out) {
    std::vector<float> out((size_t)n_out * 5120);
    for (int o = 0; o < n_out; o++)
        for (int c = 0; c < 256; c++)
            for (int pix = 0; pix < 20; pix++) out[(size_t)o * 5120 + pix * 256 + c] = w[(size_t)o * 5120 + c * 20 + pix];
    return out;
}

// block 4 of a forward of `batch` pairs samples its own input (no prep_b4 launch); prev == nullptr: images unknown yet - the usual case (4-byte aligned u8) is assumed
static bool b4_warp_in(const hnet_ctx* c, int batch, const void* prev, const void* curr, int pix_fmt) {
    if (!c->warp_in || !c->fuse_b4 || !c->x16_b4 || c->n_planes != 2 || (c->fuse_small && batch <= 8)) return false;
    return prev ? block4_warp_in_supported(prev, curr, pix_fmt == HNET_PIX_U8, c->n_planes) : true;
}

// the launches of one forward of `batch` pairs, in order (what the STAGE macro of forward_chunk records events for): the latency path
// (batch <= 8) has fewer of them
void build_stages(hnet_ctx* c, int batch, const void* prev = nullptr, const void* curr = nullptr, int pix_fmt = HNET_PIX_U8) {      // (image pointers: forward_chunk fuses the block tail into a prep launch only for 16-byte-aligned images)
    c->stages.clear();
    const hnet_config& g = c->cfg;
    auto conv_flops = [&](int l, int h, int w) {
        const ConvDesc& d = kConvs[l];
        return 2.0 * d.cout * d.cin * d.ks * d.ks * conv_out_dim(h, d.ks, d.stride) * conv_out_dim(w, d.ks, d.stride);
    };
    static const int first[4] = {0, 3, 7, 13}, last[4] = {2, 6, 12, 19}, chain_first[4] = {1, 4, 10, 17};
    const bool small = c->fuse_small && batch <= 8;
    bool pend = false;
    if (g.use_prior) {
        if (small) pend = true;
        else c->stages.push_back({"prior_dlt", 0});
    }
    const int fb = g.use_prior ? 4 - g.blocks_to_run : 0;
    for (int blk = fb; blk < 4; blk++) {
        const bool fused_prep = pend && (prev ? prep_fc_supported(prev, curr, 8 >> blk, blk == 3 && c->x16_b4 != nullptr) : (blk < 3 || c->x16_b4 != nullptr));
        if (pend && !fused_prep) c->stages.push_back({blk == fb && g.use_prior ? "prior_dlt" : "fc_dlt_b" + std::to_string(blk), blk == fb && g.use_prior ? 0.0 : 2.0 * 8 * 5120});
        if (!(blk == 3 && !pend && b4_warp_in(c, batch, prev, curr, pix_fmt)))       // (block 4 of a large batch samples its input itself: conv_b4_fused.h WARPIN)
        c->stages.push_back({std::string(fused_prep ? (blk == fb && g.use_prior ? "prior_dlt+" : "fc_dlt+") : "") + "prep_b" + std::to_string(blk + 1),
                             fused_prep && !(blk == fb && g.use_prior) ? 2.0 * 8 * 5120 : 0.0});
        pend = false;
        int h = IMG_H >> (3 - blk), w = IMG_W >> (3 - blk);
        for (int l = first[blk]; l <= last[blk]; l++) {
            double fl = conv_flops(l, h, w);
            std::string nm = kConvs[l].name;
            h = conv_out_dim(h, kConvs[l].ks, kConvs[l].stride);
            w = conv_out_dim(w, kConvs[l].ks, kConvs[l].stride);
            if (c->fuse_b42 && l == 15) {      // one launch for block_4_2 + block_4_3
                fl += conv_flops(16, h, w);
                nm = "block_4_2+4_3";
                h = conv_out_dim(h, kConvs[16].ks, kConvs[16].stride);
                w = conv_out_dim(w, kConvs[16].ks, kConvs[16].stride);
                l = 16;
            }
            if (c->fuse_b3 && l == 7) {        // one launch for block_3_0 + block_3_1
                fl += conv_flops(8, h, w);
                nm = "block_3_0+3_1";
                h = conv_out_dim(h, kConvs[8].ks, kConvs[8].stride);
                w = conv_out_dim(w, kConvs[8].ks, kConvs[8].stride);
                l = 8;
            }
            if (c->use_chain && small && l == chain_first[blk]) {      // one launch for the block's tail (chain_lat.h)
                for (int l2 = l + 1; l2 <= last[blk]; l2++) {
                    fl += conv_flops(l2, h, w);
                    h = conv_out_dim(h, kConvs[l2].ks, kConvs[l2].stride);
                    w = conv_out_dim(w, kConvs[l2].ks, kConvs[l2].stride);
                    nm += std::string("+") + (kConvs[l2].name + 6);      // "block_4_4+4_5+4_6"
                }
                l = last[blk];
            }
            if (c->fuse_b4 && l == 13) {       // one launch for block_4_0 + block_4_1
                fl += conv_flops(14, h, w);
                nm = "block_4_0+4_1";
                h = conv_out_dim(h, kConvs[14].ks, kConvs[14].stride);
                w = conv_out_dim(w, kConvs[14].ks, kConvs[14].stride);
                l = 14;
            }
            c->stages.push_back({nm, fl});
        }
        if (blk < 3) {
            if (small) pend = true;
            else c->stages.push_back({"fc_dlt_b" + std::to_string(blk + 1), 2.0 * 8 * 5120});
        }
    }
    c->stages.push_back({"heads_fc1", 2.0 * 512 * 5120 * c->n_local});
    if (small && c->n_local <= HEADS_FC2_FINISH_MAX_N) c->stages.push_back({"heads_fc2+mc_finish", 2.0 * 16 * 256 * c->n_local});
    else {
        c->stages.push_back({"heads_fc2", 2.0 * 16 * 256 * c->n_local});
        c->stages.push_back({"mc_finish", 0});
    }
    if (g.emit_error_map) c->stages.push_back({"errmap", 0});
}

struct FwdArgs {
    const void *prev, *curr;
    int pix_fmt;
    const float* prior;
    int batch;
    uint64_t seq0;
    float *mean, *cov;        // device outputs (finish path)
    float* err;               // device error map (float) or null
    uint8_t* err_u8;
    float *mean_s, *logvar_s, *h_part1;   // partial path outputs (device) or null
    bool partial;
    int pair0 = 0;            // first pair of this chunk inside the persistent buffers (caller arrays are pre-offset)
    bool use_ws = true;       // may use the context's split-K workspace (false for concurrent chunks)
    const uint64_t* seq_dev = nullptr;   // device addend to seq0 (graph replays)
    int mean_stride = 8, cov_stride = 64;   // floats between consecutive pairs of `mean` / `cov` (72 / 72: the packed [B][72] record)
    uint32_t* flag = nullptr;               // where the kernels raise the overflow / timeout bits (nullptr: the context's device word; hnet_infer's graph: a word of its pinned block)
};

#define STAGE(call)                                                                                         \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        stage_i++;                                                                                          \
        if (e_ != hipSuccess) return fail(c, HNET_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
        if (!c->prof_ev.empty() && c->prof_pos < c->prof_ev.size()) {                                       \
            e_ = hipEventRecord(c->prof_ev[c->prof_pos++], s);                                              \
            if (e_ != hipSuccess) return fail(c, HNET_ERR_DEVICE, "hipEventRecord(stage)");                 \
        }                                                                                                   \
    } while (0)

// The forward of combined_stu_model (model_to_trace.py:299-330) for `a.batch` independent frame pairs that occupy
// slots [a.pair0, a.pair0 + a.batch) of the persistent buffers; everything is enqueued on stream `s`.
int forward_chunk(hnet_ctx* c, const FwdArgs& a, hipStream_t s) {
    const hnet_config& g = c->cfg;
    const int B = a.batch;
    const size_t P0 = (size_t)a.pair0;
    static const int first[4] = {0, 3, 7, 13}, last[4] = {2, 6, 12, 19}, chain_first[4] = {1, 4, 10, 17};
    float* Hm = c->Hm + P0 * 9;
    float* Htot = c->Htot + P0 * 9;
    // split-K workspace: only for a launch that covers the whole batch on one stream (small batches)
    float* ws = a.use_ws ? c->ws : nullptr;
    const size_t wsn = a.use_ws ? c->ws_floats : 0;
    // Latency path (batch <= 8): the homography of a block is not produced by a launch of its own (prior DLT / FC + DLT + composition) but
    // recomputed inside the next block's prep kernel by every workgroup (kernels.h FcArgs): 3-4 launches fewer in the dependent chain.
    // `pend` holds what the next prep has to evaluate; the homographies alternate between Hm and Hm2 (a workgroup stores the new one while
    // others still read the old one).
    const bool small = c->fuse_small && B <= 8;
    uint32_t* const flagp = a.flag ? a.flag : c->d_flag;
    size_t stage_i = 0;                            // launches so far (index into c->stages when that list describes this forward)
    auto set_kernels = [&](int k) { if (stage_i >= 1 && stage_i <= c->stages.size()) c->stages[stage_i - 1].kernels = k; };
    // the keep bits of the heads depend on the seeds only: on the latency path they are drawn by surplus workgroups of block 4's prep launch (FcArgs::mask)
    const bool mask_in_prep = small && c->lat_tail && c->s3 && heads_fc1_one_launch(B, c->n_local, c->n_planes);
    bool mask_ready = false;
    FcArgs pend = {};
    bool have_pend = false;
    float* Hcur = Hm;                              // buffer holding the homography so far
    float* Hnext = c->Hm2 + P0 * 9;
    if (g.use_prior) {
        if (small) { pend = FcArgs{nullptr, nullptr, nullptr, nullptr, a.prior, nullptr}; have_pend = true; }
        else STAGE(launch_prior_dlt(a.prior, Hm, B, s));                               // :129-130
    }
    const int fb = g.use_prior ? 4 - g.blocks_to_run : 0;
    HNET_RANGE(range_fwd, "hnet forward");
    for (int blk = fb; blk < 4; blk++) {
        static const char* const kBlockRange[4] = {"hnet block 1", "hnet block 2", "hnet block 3", "hnet block 4 trunk"};
        HNET_RANGE(range_blk, kBlockRange[blk]);
        (void)kBlockRange;
        const bool warp = g.use_prior || blk > 0;                                    // block 1 of the full model sees raw img2 (:138)
        int h = IMG_H >> (3 - blk), w = IMG_W >> (3 - blk);
        float* x = c->x_in[blk] + P0 * h * w * 2;
        const bool b4_dma = blk == 3 && c->x16_b4 != nullptr;      // block 4 always warps (:261): the prep kernel writes the padded planes
        uint32_t* x16 = b4_dma ? c->x16_b4 + P0 * B4_HP * B4_WP : nullptr;
        B4Warp b4w = {};
        bool b4w_on = false;
        if (blk == 3) c->b4_in_stale = false;
        if (have_pend) {
            if (prep_fc_supported(a.prev, a.curr, 8 >> blk, x16 != nullptr)) {
                pend.H_out = pend.feat ? Hnext : Hcur;                                // the prior's DLT has no input homography: it may land in Hcur
                if (blk == 3 && mask_in_prep) {
                    pend.mask = c->head_mask + P0 * c->n_local * 2 * 640;
                    pend.mask_blocks = (int)(((size_t)B * c->n_local * 2 * 160 + 255) / 256);
                    pend.n_local = c->n_local; pend.s_begin = c->s_begin; pend.thr = hnet_drop_threshold(g.dropout_p);
                    pend.mc_seed = g.mc_seed; pend.pair_seq0 = a.seq0; pend.seq_dev = a.seq_dev;
                    mask_ready = true;
                }
                STAGE(launch_prep_fc(a.prev, a.curr, a.pix_fmt, pend, 8 >> blk, x, B, s, x16, c->x16_plane, c->n_planes, c->warp_exact));
                if (pend.feat) std::swap(Hcur, Hnext);
            } else {                                                                  // (unaligned images / K = 8: the separate launches)
                if (pend.feat) STAGE(launch_block_fc_dlt(pend.feat, pend.wfc, pend.bfc, pend.H_in, Hcur, B, s));
                else STAGE(launch_prior_dlt(pend.prior, Hcur, B, s));
                STAGE(launch_prep(a.prev, a.curr, a.pix_fmt, Hcur, 8 >> blk, x, B, s, x16, c->x16_plane, c->n_planes, c->warp_exact));
            }
            have_pend = false;
        } else if (blk == 3 && b4_warp_in(c, B, a.prev, a.curr, a.pix_fmt)) {
            b4w = B4Warp{(const uint8_t*)a.prev, (const uint8_t*)a.curr, Hcur};          // no launch: block_4_0 + block_4_1 samples cat(img1, warp(img2, H)) itself
            b4w_on = true;
            c->b4_in_stale = true;
        } else {
            STAGE(launch_prep(a.prev, a.curr, a.pix_fmt, warp ? Hcur : nullptr, 8 >> blk, x, B, s, x16, c->x16_plane, c->n_planes, c->warp_exact));
        }
        const float* in = x;
        const uint16_t* in16 = nullptr;
        size_t in_plane = 0;
        const size_t MB = (size_t)g.max_batch;
        bool chain_fc_done = false;                // this block's tail chain left the FC's partial sums in c->fc_part
        for (int l = first[blk]; l <= last[blk]; l++) {
            if (c->use_chain && small && P0 == 0 && l == chain_first[blk] && in16) {      // the block's tail in one launch on one XCD (chain_lat.h)
                int nxt = blk < 3 ? blk + 1 : fb;                                         // the chain launch that follows this one on the stream: the next block's, or the next forward's first
                if (nxt == blk) {                                                         // prior-1: ONE chain per forward - nobody else zeroes its area: a memset node in front of it
                    if (hipMemsetAsync(c->chain_sync + blk * CH_SYNC_WORDS, 0, CH_SYNC_WORDS * sizeof(uint32_t), s) != hipSuccess) return fail(c, HNET_ERR_DEVICE, "chain area memset");
                    nxt = 0;
                }
                ChainArgs cargs = c->chain_args[blk];
                cargs.flag = flagp;
                chain_fc_done = blk < 3 && cargs.fcw != nullptr;
                STAGE(launch_tail_chain(blk + 1, cargs, c->chain_sync + blk * CH_SYNC_WORDS, c->chain_sync + nxt * CH_SYNC_WORDS, B, s, c->chain_grid));
                l = last[blk];
                in = c->act[l];
                in16 = nullptr;
                in_plane = 0;
                h = c->act_h[l]; w = c->act_w[l];
                continue;
            }
            if (c->fuse_b4 && l == 13) {       // block_4_0 + block_4_1 in one launch; the 8-channel map stays in LDS
                const size_t cnt1 = c->a14_pad ? B42_IMG * 16 : c->act_count[14];
                uint16_t* o16 = c->act16[14] + P0 * cnt1;
                STAGE(launch_block4_fused(b4_dma ? (const void*)x16 : (const void*)in, c->x16_plane, c->b40_frag, c->conv_b[13], c->b41_frag, c->conv_b[14], o16,
                                          MB * cnt1, B, s, c->b4_flags | (c->a14_pad ? 64 : 0), c->n_planes, b4w_on ? &b4w : nullptr));
                in = nullptr; in16 = o16; in_plane = MB * cnt1;
                h = c->act_h[14]; w = c->act_w[14];
                l = 14;
                continue;
            }
            if (c->fuse_b42 && c->a14_pad && l == 15 && c->n_planes == 2 && c->b42_w2 && c->b42_w3 && in16 && h == 112 && w == 160) {   // block_4_2 + block_4_3 in one launch
                const size_t cnt1 = c->act_count[16];
                uint16_t* o16b = c->act16[16] + P0 * cnt1;
                STAGE(launch_block42_fused(in16, in_plane, c->b42_w2, c->conv_b[15], c->b42_w3, c->conv_b[16], o16b, MB * cnt1, B, s, c->n_planes));
                in = nullptr; in16 = o16b; in_plane = MB * cnt1;
                h = c->act_h[16]; w = c->act_w[16];
                l = 16;
                continue;
            }
            if (c->fuse_b3 && l == 7 && c->n_planes == 2 && c->b30_frag && c->b3f_w1 && h == 112 && w == 160) {   // block_3_0 + block_3_1 in one launch
                const size_t cnt1 = c->act_count[8];
                uint16_t* o16b = c->act16[8] + P0 * cnt1;
                STAGE(launch_block3_fused(in, c->b30_frag, c->conv_b[7], c->b3f_w1, c->conv_b[8], o16b, MB * cnt1, B, s, c->n_planes));
                in = nullptr; in16 = o16b; in_plane = MB * cnt1;
                h = c->act_h[8]; w = c->act_w[8];
                l = 8;
                continue;
            }
            const size_t cnt = c->act_count[l];
            float* o = c->act[l] ? c->act[l] + P0 * cnt : nullptr;
            uint16_t* o16 = c->act16[l] ? c->act16[l] + P0 * cnt : nullptr;
            if (c->s3 && l == 7 && c->b30_s3 && c->b30_frag && o16)
                STAGE(launch_conv_first_s3(in, c->b30_frag, c->conv_b[l], o16, MB * cnt, B, h, w, s, c->n_planes));
            else if (c->s3 && conv_is_first_s2(l) && c->first_s2 && c->s2_frag[l] && o16)
                STAGE(launch_conv_first_s2(l, in, c->s2_frag[l], c->conv_b[l], o16, MB * cnt, B, s, c->n_planes));
            else if (c->use_patch && (conv_is_patch_layer(l) || (c->use_patch32 && conv_is_patch32_layer(l) && h == 56 && w == 80)))
                STAGE(launch_conv_patch(l, in16, in_plane, B, h, w, c->patch_frag[l], c->conv_b[l], o16, MB * cnt, s, c->n_planes, c->patch_b128, c->patch_rb5));
            else if (c->s3 && conv_is_s3_layer(l)) {
                LatIO lat = {c->lat_tail && ws ? reinterpret_cast<uint32_t*>(c->ws + c->ws_floats) : nullptr, 1, false, false};
                STAGE(launch_conv_s3(l, in16, in_plane, B, h, w, c->conv_w16[l], (size_t)kConvs[l].cout * conv_padded_k(l),
                                     c->conv_b[l], o16, MB * cnt, o16 ? nullptr : o, s, ws, wsn, c->conv_wfrag[l], c->n_planes, c->s3_tile, &lat));
                set_kernels(lat.kernels);
            } else
                STAGE(launch_conv(l, in, B, h, w, c->conv_w[l], c->conv_b[l], o, s, ws, wsn, o16, MB * cnt));
            in = o;
            in16 = o16;
            in_plane = MB * cnt;
            h = c->act_h[l];
            w = c->act_w[l];
        }
        if (blk < 3) {                                                                // :143-150, :163-168, :183-188
            if (small) {
                pend = FcArgs{in, c->fc_w[blk], c->fc_b[blk], warp ? Hcur : nullptr, nullptr, nullptr};
                if (chain_fc_done) pend.fc_part = c->fc_part;      // (the unaligned-image fallback below still computes the FC from `in`)
                have_pend = true;
            }
            else STAGE(launch_block_fc_dlt(in, c->fc_w[blk], c->fc_b[blk], warp ? Hcur : nullptr, Hcur, B, s));
        }
    }
    Hm = Hcur;                                     // H_part1 of this forward
    c->H_last = Hcur - P0 * 9;
    // block 4 heads (:272-282) and output assembly (:310-317)
    HNET_RANGE(range_heads, "hnet heads + ensemble");
    const float* feat = c->act[19] + P0 * 5120;
    float* hidden = c->hidden + P0 * c->n_local * 512;
    if (c->s3) {
        LatIO lat_h = {nullptr, 1, small && c->lat_tail && c->n_planes == 2, mask_ready};
        STAGE(launch_heads_fc1_s3(feat, B, c->n_local, c->s_begin, g.dropout_p, g.mc_seed, a.seq0, c->w1_16, c->b1, hidden,
                                  c->feat16 + P0 * 5120, (size_t)g.max_batch * 5120, c->head_mask + P0 * c->n_local * 2 * 640, s, ws, wsn, a.seq_dev, c->n_planes,
                                  c->s3_tile, &lat_h));
        set_kernels(lat_h.kernels);
    }
    else
        STAGE(launch_heads_fc1(feat, B, c->n_local, c->s_begin, g.dropout_p, g.mc_seed, a.seq0, c->w1, c->b1, hidden, s, ws, wsn, a.seq_dev));
    if (a.partial) {
        STAGE(launch_heads_fc2(hidden, B, c->n_local, c->s_begin, g.dropout_p, g.mc_seed, a.seq0, c->w2, c->b2,
                               a.mean_s, a.logvar_s, s, a.seq_dev, flagp));
        if (a.h_part1) {
            hipError_t e = hipMemcpyAsync(a.h_part1, Hm, (size_t)B * 9 * sizeof(float), hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return fail(c, HNET_ERR_DEVICE, "copy H_part1");
        }
        return HNET_OK;
    }
    float* ms = c->mean_s + P0 * c->n_local * 8;
    float* lv = c->logvar_s + P0 * c->n_local * 8;
    if (small && c->n_local <= HEADS_FC2_FINISH_MAX_N) {
        STAGE(launch_heads_fc2_finish(hidden, B, c->n_local, c->s_begin, g.dropout_p, g.mc_seed, a.seq0, c->w2, c->b2, Hm, a.mean, a.cov, Htot, s,
                                      a.seq_dev, flagp, a.mean_stride, a.cov_stride));
    } else {
        STAGE(launch_heads_fc2(hidden, B, c->n_local, c->s_begin, g.dropout_p, g.mc_seed, a.seq0, c->w2, c->b2, ms, lv, s, a.seq_dev));
        STAGE(launch_mc_finish(ms, lv, c->n_local, Hm, B, a.mean, a.cov, Htot, s, flagp, a.mean_stride, a.cov_stride));
    }
    if (g.emit_error_map && (a.err || a.err_u8))                                     // :319-327
        STAGE(launch_errmap(a.prev, a.curr, a.pix_fmt, Htot, a.err, a.err_u8, B, s));
    return HNET_OK;
}

// Validates and enqueues the forward of the whole batch on stream `s`.
// (Round 1 had an HNET_STREAMS switch that cut the batch into chunks on separate HIP streams.  It never gave a speed-up and
// the round-2 determinism test showed run-to-run differences of ~1e-3 px between concurrent chunks on the split-bf16 path
// (tools/dbg_streams.py; single-stream runs are bit-reproducible), so the chunked mode was removed rather than shipped.)
int forward(hnet_ctx* c, const FwdArgs& a, hipStream_t s) {
    const hnet_config& g = c->cfg;
    if (a.batch < 1) return fail(c, HNET_ERR_INVALID_ARG, "batch < 1");
    if (a.batch > g.max_batch) return fail(c, HNET_ERR_CAPACITY, "batch exceeds max_batch");
    if (g.use_prior && !a.prior) return fail(c, HNET_ERR_INVALID_ARG, "context uses a prior but none was given");
    c->last_batch = a.batch;
    const int rc = forward_chunk(c, a, s);
    // a forward that stopped part-way may leave split-K tile counters of the latency path non-zero (a launch that failed after its predecessors ran):
    // they are zeroed again behind whatever was enqueued, so the next forward starts from the state it assumes (kernels.h SPLITK_TICKETS)
    if (rc != HNET_OK && c->ws) (void)hipMemsetAsync(c->ws + c->ws_floats, 0, SPLITK_TICKETS * sizeof(uint32_t), s);
    if (rc != HNET_OK && c->chain_sync) (void)hipMemsetAsync(c->chain_sync, 0, CH_AREAS * CH_SYNC_WORDS * sizeof(uint32_t), s);      // (likewise the chains' counter areas)
    return rc;
}

// Captures `body` (work enqueued on c->stream) into an executable graph.  Returns nullptr when capture is not possible;
// callers then fall back to eager launches of the same kernels.
template <class F>
hipGraphExec_t capture_graph(hnet_ctx* c, F&& body) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return nullptr;
    const int rc = body();
    const hipError_t e = hipStreamEndCapture(c->stream, &graph);
    if (rc != HNET_OK || e != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); (void)hipGetLastError(); return nullptr; }
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
    (void)hipGraphDestroy(graph);
    return exec;
}

// Weights -> device, in the layouts of the kernels of the context's arithmetic mode (c->s3, c->n_planes).  Called by hnet_create and again by
// demote_to_bf16x3 (buffers of an earlier call are released first).  On failure the caller destroys the context.
// weight planes of the implicit-GEMM layers (igemm_s3.h): the fp16 mode uses the two-plane activation split there (s3_wplanes_gemm)
static inline void wsplit_gemm(float w, int np, uint16_t& a, uint16_t& b, uint16_t& c3) {
    if (np == 2) { split2h(w, a, b); c3 = 0; }
    else split3(w, a, b, c3);
}
int upload_weights(hnet_ctx* c, const Blob& b) {
#define CK(expr)                                                                    \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "hnet weights: %s: %s\n", #expr, hipGetErrorString(e_)); \
            return HNET_ERR_DEVICE;                                                 \
        }                                                                           \
    } while (0)
    {
        auto fr = [](auto*& p) { if (p) (void)hipFree(p); p = nullptr; };
        for (int l = 0; l < 20; l++) { fr(c->patch_frag[l]); fr(c->conv_w[l]); fr(c->conv_b[l]); fr(c->conv_w16[l]); fr(c->conv_wfrag[l]); fr(c->chain_w[l]); }
        for (int k = 0; k < 3; k++) { fr(c->fc_w[k]); fr(c->fc_b[k]); }
        fr(c->s2_frag[0]); fr(c->s2_frag[3]); fr(c->b30_frag); fr(c->b40_frag); fr(c->b41_frag); fr(c->w1_16); fr(c->b3f_w1); fr(c->b42_w2); fr(c->b42_w3);
        fr(c->w1); fr(c->b1); fr(c->w2); fr(c->b2);
    }
    // ---- weights: names are the reference state_dict keys (model_to_trace.py:88-115, :210-235)
    for (int l = 0; l < 20; l++) {
        const ConvDesc& d = kConvs[l];
        const std::string pre = std::string(d.block == 4 ? "model_last_block_list.0." : "model_part1.") + d.name + ".0.";
        const Tensor* w = b.find(pre + "weight", {(uint32_t)d.cout, (uint32_t)d.cin, (uint32_t)d.ks, (uint32_t)d.ks});
        const Tensor* bi = b.find(pre + "bias", {(uint32_t)d.cout});
        if (!w || !bi) return HNET_ERR_BAD_WEIGHTS;
        if (c->s3 && c->n_planes == 2 && chain_layer(l)) {      // latency path: the layer's weights as the fragments of its one-XCD tail chain (chain_lat.h)
            std::vector<uint16_t> fr;
            if (!chain_pack_weights(l, w->data, fr)) return HNET_ERR_BAD_WEIGHTS;
            CK(hipMalloc((void**)&c->chain_w[l], fr.size() * 2));
            CK(hipMemcpy(c->chain_w[l], fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
        }
        if (c->s3 && l == 13) {     // block_4_0 for the fused kernel: K index 8g+j of step st = (kh = 2st + (g>>1), kk = 8(g&1) + j)
            std::vector<uint16_t> fr((size_t)5 * 3 * 64 * 8, 0);     // slot 4: kernel row 6 alone as 16x16x16 fragments (K = 4 gg + e = tap 2 gg + (e >> 1), ci = e & 1), low 8 bytes
            for (int ln = 0; ln < 64; ln++) {
                const int n = ln & 15, gg = ln >> 4, dx = n >> 3, co = n & 7;
                for (int e = 0; e < 4; e++) {
                    const int kk = 4 * gg + e, kw = (kk >> 1) - dx, ci = kk & 1;
                    if (kw < 0 || kw >= 7) continue;
                    uint16_t sp[3];
                    wsplit_np(w->data[(((size_t)co * 2 + ci) * 7 + 6) * 7 + kw], c->n_planes, sp[0], sp[1], sp[2]);
                    for (int pl = 0; pl < 3; pl++) fr[(((size_t)4 * 3 + pl) * 64 + ln) * 8 + e] = sp[pl];
                }
            }
            for (int st = 0; st < 4; st++)
                for (int ln = 0; ln < 64; ln++) {
                    const int n = ln & 15, gg = ln >> 4, kh = 2 * st + (gg >> 1);
                    if (kh >= 7) continue;
                    const int dx = n >> 3, co = n & 7;
                    for (int j = 0; j < 8; j++) {
                        const int kk = 8 * (gg & 1) + j, kw = (kk >> 1) - dx, ci = kk & 1;
                        if (kw < 0 || kw >= 7) continue;
                        uint16_t sp[3];
                        wsplit_np(w->data[(((size_t)co * 2 + ci) * 7 + kh) * 7 + kw], c->n_planes, sp[0], sp[1], sp[2]);
                        for (int pl = 0; pl < 3; pl++) fr[(((size_t)st * 3 + pl) * 64 + ln) * 8 + j] = sp[pl];
                    }
                }
            CK(hipMalloc((void**)&c->b40_frag, fr.size() * 2));
            CK(hipMemcpy(c->b40_frag, fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
        }
        if (c->s3 && l == 7) {      // block_3_0 for conv7_c2_s1_s3_kernel: lane (n = l&31 = (dx, co), hh = l>>5), kk = 8hh + j of kernel row kh
            std::vector<uint16_t> fr((size_t)7 * 3 * 64 * 8, 0);
            for (int kh = 0; kh < 7; kh++)
                for (int ln = 0; ln < 64; ln++) {
                    const int n = ln & 31, hh = ln >> 5, dx = n >> 4, co = n & 15;
                    for (int j = 0; j < 8; j++) {
                        const int kk = 8 * hh + j, kw = (kk >> 1) - dx, ci = kk & 1;
                        if (kw < 0 || kw >= 7) continue;
                        uint16_t sp[3];
                        wsplit_np(w->data[(((size_t)co * 2 + ci) * 7 + kh) * 7 + kw], c->n_planes, sp[0], sp[1], sp[2]);
                        for (int pl = 0; pl < 3; pl++) fr[(((size_t)kh * 3 + pl) * 64 + ln) * 8 + j] = sp[pl];
                    }
                }
            CK(hipMalloc((void**)&c->b30_frag, fr.size() * 2));
            CK(hipMemcpy(c->b30_frag, fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
            c->b30_s3 = true;
        }
        if (c->s3 && l == 8 && c->n_planes == 2) {   // block_3_1 for the fused kernel: lane (i, g) of n-tile nt, step st: channel 16 nt + i, tap 2 st + (g >> 1), ci 8 (g & 1) + j
            std::vector<uint16_t> f2((size_t)2 * 13 * 2 * 64 * 8, 0);
            for (int nt = 0; nt < 2; nt++)
                for (int st = 0; st < 13; st++)
                    for (int ln = 0; ln < 64; ln++) {
                        const int co = 16 * nt + (ln & 15), gg = ln >> 4, t = 2 * st + (gg >> 1);
                        if (t >= 25) continue;
                        const int kh = t / 5, kw = t % 5;
                        for (int j = 0; j < 8; j++) {
                            const int ci = 8 * (gg & 1) + j;
                            uint16_t a0, a1;
                            split2h(w->data[(((size_t)co * 16 + ci) * 5 + kh) * 5 + kw], a0, a1);
                            f2[((((size_t)nt * 13 + st) * 2 + 0) * 64 + ln) * 8 + j] = a0;
                            f2[((((size_t)nt * 13 + st) * 2 + 1) * 64 + ln) * 8 + j] = a1;
                        }
                    }
            CK(hipMalloc((void**)&c->b3f_w1, f2.size() * 2));
            CK(hipMemcpy(c->b3f_w1, f2.data(), f2.size() * 2, hipMemcpyHostToDevice));
        }
        if (c->s3 && (l == 15 || l == 16) && c->n_planes == 2) {   // block_4_2 / block_4_3 for the fused kernel (conv_b42_fused.h), two weight planes
            const int nnt = d.cout / 16, nst = l == 15 ? 5 : 9;
            std::vector<uint16_t> f2((size_t)nnt * nst * 2 * 64 * 8, 0);
            for (int nt = 0; nt < nnt; nt++)
                for (int st = 0; st < nst; st++)
                    for (int ln = 0; ln < 64; ln++) {
                        const int co = 16 * nt + (ln & 15), gg = ln >> 4;
                        const int t = l == 15 ? 2 * st + (gg >> 1) : st;           // 16 -> 32: two taps per 32-deep step; 32 -> 64: one
                        if (t >= 9) continue;
                        const int kh = t / 3, kw = t % 3;
                        for (int j = 0; j < 8; j++) {
                            const int ci = l == 15 ? 8 * (gg & 1) + j : 8 * gg + j;
                            uint16_t a0, a1;
                            split2h(w->data[(((size_t)co * d.cin + ci) * 3 + kh) * 3 + kw], a0, a1);
                            f2[((((size_t)nt * nst + st) * 2 + 0) * 64 + ln) * 8 + j] = a0;
                            f2[((((size_t)nt * nst + st) * 2 + 1) * 64 + ln) * 8 + j] = a1;
                        }
                    }
            uint16_t*& dstp = l == 15 ? c->b42_w2 : c->b42_w3;
            CK(hipMalloc((void**)&dstp, f2.size() * 2));
            CK(hipMemcpy(dstp, f2.data(), f2.size() * 2, hipMemcpyHostToDevice));
        }
        if (c->s3 && conv_is_first_s2(l)) {   // lane (i = channel of the n-tile, g): kernel row 2 st + (g>>1), taps 4 (g&1) + (j>>1), ci = j&1
            const int nt_n = d.cout / 16;
            std::vector<uint16_t> fr((size_t)nt_n * 4 * 3 * 64 * 8, 0);
            for (int nt = 0; nt < nt_n; nt++)
                for (int st = 0; st < 4; st++)
                    for (int ln = 0; ln < 64; ln++) {
                        const int co = nt * 16 + (ln & 15), gg = ln >> 4, kh = 2 * st + (gg >> 1);
                        if (kh >= 7) continue;
                        for (int j = 0; j < 8; j++) {
                            const int kw = 4 * (gg & 1) + (j >> 1), ci = j & 1;
                            if (kw >= 7) continue;
                            uint16_t sp[3];
                            wsplit_np(w->data[(((size_t)co * 2 + ci) * 7 + kh) * 7 + kw], c->n_planes, sp[0], sp[1], sp[2]);
                            for (int pl = 0; pl < 3; pl++) fr[((((size_t)nt * 4 + st) * 3 + pl) * 64 + ln) * 8 + j] = sp[pl];
                        }
                    }
            CK(hipMalloc((void**)&c->s2_frag[l], fr.size() * 2));
            CK(hipMemcpy(c->s2_frag[l], fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
            c->first_s2 = true;
        }
        if (c->s3 && conv_is_patch32_layer(l)) {   // 32 -> 64, 3x3: step st = tap st; lane group g -> channels 8g .. 8g+7 (odd groups rotated by 4)
            std::vector<uint16_t> fr((size_t)4 * 9 * 3 * 64 * 8, 0);
            for (int nt = 0; nt < 4; nt++)
                for (int st = 0; st < 9; st++)
                    for (int ln = 0; ln < 64; ln++) {
                        const int n = nt * 16 + (ln & 15), gg = ln >> 4, kh = st / 3, kw = st % 3;
                        for (int j = 0; j < 8; j++) {
                            const int ci = 8 * gg + j;
                            uint16_t sp[3];
                            wsplit_np(w->data[(((size_t)n * 32 + ci) * 3 + kh) * 3 + kw], c->n_planes, sp[0], sp[1], sp[2]);
                            for (int pl = 0; pl < 3; pl++) fr[((((size_t)nt * 9 + st) * 3 + pl) * 64 + ln) * 8 + j] = sp[pl];
                        }
                    }
            CK(hipMalloc((void**)&c->patch_frag[l], fr.size() * 2));
            CK(hipMemcpy(c->patch_frag[l], fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
        }
        if (c->s3 && conv_is_patch_layer(l)) {   // 16 -> 32, KSxKS: step st = taps 2st, 2st+1; lane group g -> tap 2st + (g>>1), ci 8(g&1)+j
            const int ks = d.ks, nstep = (ks * ks + 1) / 2;
            std::vector<uint16_t> fr((size_t)2 * nstep * 3 * 64 * 8, 0);
            for (int nt = 0; nt < 2; nt++)
                for (int st = 0; st < nstep; st++)
                    for (int ln = 0; ln < 64; ln++) {
                        const int n = nt * 16 + (ln & 15), gg = ln >> 4, t = 2 * st + (gg >> 1);
                        if (t >= ks * ks) continue;
                        const int kh = t / ks, kw = t % ks;
                        for (int j = 0; j < 8; j++) {
                            // odd lane groups read their 16-byte chunk high half first (conv_patch_s2.h): element j = channel (j + 4) % 8 of the half
                            const int ci = 8 * (gg & 1) + (((gg & 1) && !c->patch_b128) ? (j + 4) % 8 : j);
                            uint16_t sp[3];
                            wsplit_np(w->data[(((size_t)n * 16 + ci) * ks + kh) * ks + kw], c->n_planes, sp[0], sp[1], sp[2]);
                            for (int pl = 0; pl < 3; pl++) fr[((((size_t)nt * nstep + st) * 3 + pl) * 64 + ln) * 8 + j] = sp[pl];
                        }
                    }
            CK(hipMalloc((void**)&c->patch_frag[l], fr.size() * 2));
            CK(hipMemcpy(c->patch_frag[l], fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
        }
        if (conv_is_first_direct(l)) CK(upload(&c->conv_w[l], pack_first_weights(w->data, d.cout)));
        else {
            const std::vector<float> packed = pack_conv(w->data, d, conv_padded_k(l));
            CK(upload(&c->conv_w[l], packed));
            if (c->s3 && l == 14) {                 // block_4_1 B-fragments for the fused kernel: tap t = 4*st + g, 8 channels
                std::vector<uint16_t> fr((size_t)7 * 3 * 64 * 8, 0);
                for (int st = 0; st < 7; st++)
                    for (int ln = 0; ln < 64; ln++) {
                        // fp16-plane mode: the tap table of kernels.h (lane-group pairs share one ds_read_b128), channels in order
                        const int n = ln & 15, gg = ln >> 4, t = c->n_planes == 2 ? b41_tap(st, gg) : 4 * st + gg;
                        if (t < 0 || t >= 25) continue;
                        const int kh = t / 5, kw = t % 5;
                        for (int j = 0; j < 8; j++) {
                            // three-plane / bf16 modes: odd lane groups read their 16-byte chunk high half first (conv_b4_fused.h): element j = channel (j + 4) % 8
                            const int ci = (c->n_planes != 2 && (gg & 1)) ? (j + 4) % 8 : j;
                            uint16_t sp[3];
                            wsplit_np(w->data[(((size_t)n * 8 + ci) * 5 + kh) * 5 + kw], c->n_planes, sp[0], sp[1], sp[2]);
                            for (int pl = 0; pl < 3; pl++) fr[(((size_t)st * 3 + pl) * 64 + ln) * 8 + j] = sp[pl];
                        }
                    }
                CK(hipMalloc((void**)&c->b41_frag, fr.size() * 2));
                CK(hipMemcpy(c->b41_frag, fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
            }
            if (c->s3 && conv_is_s3_layer(l)) {     // exact 3-way bf16 split of every weight: planes [3][Cout][Kp]
                std::vector<uint16_t> pl(packed.size() * 3);
                for (size_t i = 0; i < packed.size(); i++)
                    wsplit_gemm(packed[i], c->n_planes, pl[i], pl[packed.size() + i], pl[2 * packed.size() + i]);
                CK(hipMalloc((void**)&c->conv_w16[l], pl.size() * 2));
                CK(hipMemcpy(c->conv_w16[l], pl.data(), pl.size() * 2, hipMemcpyHostToDevice));
            }
            if (c->n_planes == 2 && conv_region_layer(l)) {
                // igemm_region.h: the two weight planes as MFMA fragments in the order the kernel consumes them:
                // [Cout / 16][Cin / 64 chunks][taps, padded][2 steps][2 planes][64 lanes][8 halves]; lane (r = lane & 15, g = lane >> 4) holds
                // output channel 16 nt + r, input channels 64 c + 32 st + 8 g .. + 7 of tap t (taps >= KS x KS: the zero-weight padding tap of the K-split form)
                const int ntap = d.ks * d.ks, ntap_pad = conv_region_taps_padded(l), nchunk = d.cin / 64;
                std::vector<uint16_t> fr((size_t)(d.cout / 16) * nchunk * ntap_pad * 2 * 2 * 64 * 8, 0);
                for (int nt = 0; nt < d.cout / 16; nt++)
                    for (int cc = 0; cc < nchunk; cc++)
                        for (int t = 0; t < ntap; t++)
                            for (int st = 0; st < 2; st++)
                                for (int ln = 0; ln < 64; ln++)
                                    for (int e = 0; e < 8; e++) {
                                        const int n = nt * 16 + (ln & 15), ci = 64 * cc + 32 * st + 8 * (ln >> 4) + e;
                                        uint16_t sp[3];
                                        wsplit_gemm(w->data[(((size_t)n * d.cin + ci) * d.ks + t / d.ks) * d.ks + t % d.ks], 2, sp[0], sp[1], sp[2]);
                                        const size_t base = ((((size_t)(nt * nchunk + cc) * ntap_pad + t) * 2 + st) * 2) * 64 * 8;
                                        fr[base + (size_t)ln * 8 + e] = sp[0];
                                        fr[base + 64 * 8 + (size_t)ln * 8 + e] = sp[1];
                                    }
                CK(hipMalloc((void**)&c->conv_wfrag[l], fr.size() * 2));
                CK(hipMemcpy(c->conv_wfrag[l], fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
            }
        }
        CK(upload(&c->conv_b[l], std::vector<float>(bi->data, bi->data + d.cout)));
    }
    for (int k = 0; k < 3; k++) {
        const std::string pre = "model_part1.fc_block_" + std::to_string(k + 1) + ".";
        const Tensor* w = b.find(pre + "weight", {8, 5120});
        const Tensor* bi = b.find(pre + "bias", {8});
        if (!w || !bi) return HNET_ERR_BAD_WEIGHTS;
        CK(upload(&c->fc_w[k], permute_fc(w->data, 8)));
        CK(upload(&c->fc_b[k], std::vector<float>(bi->data, bi->data + 8)));
    }
    {
        static const char* heads[2] = {"fc_block_4_mean", "fc_block_4_uncertainty"};
        std::vector<float> w1, b1, w2, b2;
        for (int h = 0; h < 2; h++) {
            const std::string pre = std::string("model_last_block_list.0.") + heads[h] + ".";
            const Tensor* tw1 = b.find(pre + "1.weight", {256, 5120});
            const Tensor* tb1 = b.find(pre + "1.bias", {256});
            const Tensor* tw2 = b.find(pre + "4.weight", {8, 256});
            const Tensor* tb2 = b.find(pre + "4.bias", {8});
            if (!tw1 || !tb1 || !tw2 || !tb2) return HNET_ERR_BAD_WEIGHTS;
            std::vector<float> p = permute_fc(tw1->data, 256);
            w1.insert(w1.end(), p.begin(), p.end());
            b1.insert(b1.end(), tb1->data, tb1->data + 256);
            w2.insert(w2.end(), tw2->data, tw2->data + 8 * 256);
            b2.insert(b2.end(), tb2->data, tb2->data + 8);
        }
        CK(upload(&c->w1, w1)); CK(upload(&c->b1, b1)); CK(upload(&c->w2, w2)); CK(upload(&c->b2, b2));
        if (c->s3) {
            std::vector<uint16_t> pl(w1.size() * 3);
            for (size_t i = 0; i < w1.size(); i++) wsplit_gemm(w1[i], c->n_planes, pl[i], pl[w1.size() + i], pl[2 * w1.size() + i]);
            CK(hipMalloc((void**)&c->w1_16, pl.size() * 2));
            CK(hipMemcpy(c->w1_16, pl.data(), pl.size() * 2, hipMemcpyHostToDevice));
        }
    }

    return HNET_OK;
#undef CK
}

int create_impl(const hnet_config* cfg_in, const uint8_t* blob, size_t len, hnet_ctx** out, hipStream_t preset_stream = nullptr) {
    if (!cfg_in || !out) return HNET_ERR_INVALID_ARG;
    hnet_config g;
    hnet_default_config(&g);
    memcpy(&g, cfg_in, std::min<size_t>(cfg_in->struct_size ? cfg_in->struct_size : sizeof(g), sizeof(g)));
    g.struct_size = sizeof(g);
    if (g.max_batch < 1) return HNET_ERR_INVALID_ARG;
    if (g.graph < HNET_GRAPH_DEFAULT || g.graph > HNET_GRAPH_TIMING) return HNET_ERR_INVALID_ARG;      // (unknown values no longer select the defaults silently)
    if (g.variant & ~(uint32_t)(HNET_VARIANT_GEMM_MASK | HNET_VARIANT_NO_LATENCY_PATH | HNET_VARIANT_UNFUSED_B3 | HNET_VARIANT_UNFUSED_B42 | HNET_VARIANT_NO_CHAIN | HNET_VARIANT_CHAIN_GRID_8 | HNET_VARIANT_CHAIN_GRID_3 | HNET_VARIANT_WARP_FUSE | HNET_VARIANT_GRAPH_COPIES | HNET_VARIANT_CHAIN_NO_FC)) return HNET_ERR_INVALID_ARG;
    {
        const uint32_t code = g.variant & HNET_VARIANT_GEMM_MASK;
        static const uint32_t known[] = {0, 13, 20, 21, 22, 25, 30};
        if (std::find(std::begin(known), std::end(known), code) == std::end(known)) return HNET_ERR_INVALID_ARG;
    }
    // the LDS-DMA / buffer-load kernels address an activation with 31-bit byte offsets (a buffer descriptor covers 2 GiB; an offset beyond it reads zeros,
    // silently).  The largest array is block_4_1's bordered map, B42_IMG x 16 channels x 2 bytes = 603 520 bytes per pair and plane, and block42_fused_kernel
    // reaches BOTH fp16 planes through one descriptor (lane offset = plane + patch chunk, scalar offset = tile origin): the sum of the two stays inside the
    // descriptor - whichever of them the hardware range-checks - for 2 x max_batch x 603 520 < 2^31 -> 1 779 pairs (round 4 bounded one plane: 3 558).
    // Larger batches: several calls.
    if ((size_t)2 * g.max_batch * B42_IMG * 32 >= ((size_t)1 << 31)) return HNET_ERR_CAPACITY;
    if (g.precision != HNET_PREC_FP32 && g.precision != HNET_PREC_BF16X3 && g.precision != HNET_PREC_BF16 && g.precision != HNET_PREC_F16X2)
        return HNET_ERR_UNSUPPORTED;
    Blob b;
    if (!parse_blob(blob, len, b)) return HNET_ERR_BAD_WEIGHTS;
    // the model variant: what the reference bakes into the traced .pt it loads (trace_model.py:16,36-46; HomographyNet.cpp:81-124 only names the file) comes from
    // the blob's `hnet.variant` record wherever the caller left the field at HNET_FROM_FILE; without a record, the reference's launch values
    {
        const Tensor* v = b.find("hnet.variant", {8});
        const bool rec = v && v->data[0] == 1.0f;
        if (v && !rec) return HNET_ERR_BAD_WEIGHTS;                       // a record version this library does not know
        // the record's numbers come from a file: finite, integral and in range BEFORE any float -> int conversion (a NaN or 1e30 there is undefined behaviour)
        int r_prior = 1, r_blocks = 3, r_mc = 16, r_err = 0;
        float r_p = 0.05f;
        if (rec) {
            auto rec_int = [&](int i, int lo, int hi, int& out) {
                const float f = v->data[i];
                if (!(f >= (float)lo && f <= (float)hi) || f != std::floor(f)) return false;
                out = (int)f;
                return true;
            };
            r_p = v->data[4];
            if (!rec_int(1, 0, 1, r_prior) || !rec_int(2, 1, 3, r_blocks) || !rec_int(3, 1, 256, r_mc) || !rec_int(5, 0, 1, r_err) || !(r_p >= 0.f && r_p < 1.f))
                return HNET_ERR_BAD_WEIGHTS;
        }
        if (g.use_prior == HNET_FROM_FILE) g.use_prior = r_prior;
        if (g.blocks_to_run == HNET_FROM_FILE) g.blocks_to_run = r_blocks;
        if (g.mc_samples == HNET_FROM_FILE) g.mc_samples = r_mc;
        if (g.dropout_p < 0.f) g.dropout_p = r_p;
        if (g.emit_error_map == HNET_FROM_FILE) g.emit_error_map = r_err;
    }
    if (g.mc_samples < 1 || g.mc_samples > 256 || !(g.dropout_p >= 0.f) || g.dropout_p >= 1.f || (g.use_prior != 0 && g.use_prior != 1) ||
        (g.emit_error_map != 0 && g.emit_error_map != 1))
        return HNET_ERR_INVALID_ARG;
    if (g.use_prior && (g.blocks_to_run < 1 || g.blocks_to_run > 3)) return HNET_ERR_INVALID_ARG;
    if ((size_t)g.max_batch * std::max((size_t)g.mc_samples * 1280, (size_t)4 * 5120) + 1024 >= ((size_t)1 << 32)) return HNET_ERR_CAPACITY;   // 32-bit indices of the keep-bit kernel (s3_dispatch.h)
    if (g.mc_sample_begin == 0 && g.mc_sample_end == 0) g.mc_sample_end = g.mc_samples;
    if (g.mc_sample_begin < 0 || g.mc_sample_end > g.mc_samples || g.mc_sample_begin >= g.mc_sample_end)
        return HNET_ERR_INVALID_ARG;

    if (g.precision == HNET_PREC_F16X2) {     // fp16 planes carry 4096 w: every matrix-core weight must stay below 16 (s3_format.h)
        float wmax = 0.f;
        for (int l = 0; l < 20; l++) {
            const ConvDesc& d = kConvs[l];
            const Tensor* w = b.find(std::string(d.block == 4 ? "model_last_block_list.0." : "model_part1.") + d.name + ".0.weight",
                                     {(uint32_t)d.cout, (uint32_t)d.cin, (uint32_t)d.ks, (uint32_t)d.ks});
            if (!w) return HNET_ERR_BAD_WEIGHTS;
            for (size_t i = 0; i < (size_t)d.cout * d.cin * d.ks * d.ks; i++) wmax = std::max(wmax, std::fabs(w->data[i]));
        }
        for (const char* head : {"fc_block_4_mean", "fc_block_4_uncertainty"}) {
            const Tensor* w = b.find(std::string("model_last_block_list.0.") + head + ".1.weight", {256, 5120});
            if (!w) return HNET_ERR_BAD_WEIGHTS;
            for (size_t i = 0; i < (size_t)256 * 5120; i++) wmax = std::max(wmax, std::fabs(w->data[i]));
        }
        if (!(wmax < 15.99f)) {    // same results, twice the matrix-core work: not an error
            fprintf(stderr, "hnet_create: HNET_PREC_F16X2 needs |weight| < 16 (largest here: %g): using HNET_PREC_BF16X3\n", wmax);
            g.precision = HNET_PREC_BF16X3;
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || g.device_id < 0 || g.device_id >= ndev) return HNET_ERR_DEVICE;
    hnet_ctx* c = new hnet_ctx();
    c->cfg = g;
    c->s3 = g.precision != HNET_PREC_FP32;   // the 16-bit matrix-core kernels; their arithmetic mode = the number of activation planes (s3_format.h)
    c->n_planes = g.precision == HNET_PREC_BF16 ? 1 : g.precision == HNET_PREC_F16X2 ? 2 : 3;
    // Kernel selection.  The library reads NO environment variable: what used to be HNET_* switches read at hnet_create (rounds 1 - 3) is either
    // gone with the kernels that lost ("Removed in round 4") or a documented field of hnet_config (warp_exact, graph, variant) that
    // the tests and tools/ab_bench.py set explicitly.
    c->fuse_b4 = c->s3;
    c->use_patch = c->s3;
    c->use_patch32 = true;
    c->warp_exact = g.warp_exact != 0;
    c->fuse_small = !(g.variant & HNET_VARIANT_NO_LATENCY_PATH);
    c->fuse_b3 = c->n_planes == 2 && !(g.variant & HNET_VARIANT_UNFUSED_B3);
    c->fuse_b42 = c->n_planes == 2 && !(g.variant & HNET_VARIANT_UNFUSED_B42);
    c->s3_tile = (int)(g.variant & HNET_VARIANT_GEMM_MASK);
    c->lat_tail = c->s3_tile != 30;                // (30: the splitk_reduce* launches and the split-K heads of rounds 1 - 4, A/B and bitwise tests)
    c->a14_pad = c->fuse_b4 && c->fuse_b42;
    c->patch_rb5 = 5;
    c->patch_b128 = true;
    c->b4_flags = 0;
    c->n_local = g.mc_sample_end - g.mc_sample_begin;
    c->s_begin = g.mc_sample_begin;
#define CK(expr)                                                                    \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "hnet_create: %s: %s\n", #expr, hipGetErrorString(e_)); \
            hnet_destroy(c);                                                        \
            return HNET_ERR_DEVICE;                                                 \
        }                                                                           \
    } while (0)
    CK(hipSetDevice(g.device_id));
    CK(conv_kernels_init_device());      // dynamic-LDS limits of the patch / fused kernels: per device, so set at every create
    CK(chain_init_device());
    if (preset_stream) { c->stream = preset_stream; c->owns_stream = false; }      // a group member: the group created its streams first, each on a priority level / hardware queue of its own
    else CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CK(hipEventCreate(&c->ev0));
    CK(hipEventCreate(&c->ev1));
    { const int rc_w = upload_weights(c, b); if (rc_w != HNET_OK) { hnet_destroy(c); return rc_w; } }
    if (c->n_planes == 2) c->blob_copy.assign(blob, blob + len);

    // ---- persistent activation buffers (NHWC fp32), one per layer so every intermediate can be read back
    const size_t MB = (size_t)g.max_batch;
    static const int first[4] = {0, 3, 7, 13}, last[4] = {2, 6, 12, 19};
    for (int blk = 0; blk < 4; blk++) {
        int h = IMG_H >> (3 - blk), w = IMG_W >> (3 - blk);
        CK(dalloc(&c->x_in[blk], MB * h * w * 2));
        for (int l = first[blk]; l <= last[blk]; l++) {
            h = conv_out_dim(h, kConvs[l].ks, kConvs[l].stride);
            w = conv_out_dim(w, kConvs[l].ks, kConvs[l].stride);
            c->act_c[l] = kConvs[l].cout; c->act_h[l] = h; c->act_w[l] = w;
            c->act_count[l] = (size_t)h * w * kConvs[l].cout;
            if (c->fuse_b4 && l == 13) continue;                   // block_4_0's output lives in LDS only (conv_b4_fused.h): 880 MB at batch 256 saved
            if (c->s3 && l == 14 && c->a14_pad) {                  // two bordered fp16 planes, or (after a demotion to split-bf16) three plain ones; the border stays zero
                const size_t bytes = std::max((size_t)3 * MB * c->act_count[l], (size_t)2 * MB * B42_IMG * 16) * 2;
                CK(hipMalloc((void**)&c->act16[l], bytes));
                CK(hipMemset(c->act16[l], 0, bytes));
            } else
            if (c->s3 && l != last[blk]) CK(hipMalloc((void**)&c->act16[l], 3 * MB * c->act_count[l] * 2));   // feeds a conv: S3 planes
            else CK(dalloc(&c->act[l], MB * c->act_count[l]));
        }
    }
    // measured on MI355X: graph replay 0.302 ms vs eager 0.292 ms per batch-1 forward - the batch-1 latency is device side
    // (45 short dependent kernels), not host launch cost, so replay is opt-in (HNET_GRAPH=1)
    // streaming entry point: one graph launch instead of ~43 kernel launches per frame: end-to-end p50 0.380 -> 0.334 ms
    // (device time of the forward 0.289 -> 0.300 ms)
    c->use_graph = g.graph != HNET_GRAPH_OFF;
    c->graph_timing = g.graph == HNET_GRAPH_TIMING;
    c->graph_zero_copy = !(g.variant & HNET_VARIANT_GRAPH_COPIES);
    CK(hipMalloc((void**)&c->d_seq, 8));
    CK(hipMemset(c->d_seq, 0, 8));
    CK(hipMalloc((void**)&c->d_flag, 4));
    CK(hipMemset(c->d_flag, 0, 4));
    CK(hipHostMalloc((void**)&c->pinned, sizeof(hnet_ctx::Pinned), hipHostMallocDefault));
    for (int i = 0; i < 2; i++) {
        CK(hipHostMalloc((void**)&c->pinned_img[i], NPIX, hipHostMallocDefault));
        CK(hipEventCreateWithFlags(&c->ev_img[i], hipEventDisableTiming));
    }
    if (c->fuse_b4) {                                        // zeroed once: the border is block_4_0's zero padding and is never written again
        c->x16_plane = MB * B4_HP * B4_WP;
        CK(hipMalloc((void**)&c->x16_b4, 3 * c->x16_plane * 4));
        CK(hipMemset(c->x16_b4, 0, 3 * c->x16_plane * 4));
    }
    // the one-XCD tail chains of the latency path (chain_lat.h): default mode only; variant 30 (the round-4 latency path) and NO_CHAIN keep the launches
    c->use_chain = c->n_planes == 2 && c->fuse_small && c->lat_tail && !(g.variant & HNET_VARIANT_NO_CHAIN);
    c->chain_grid = (g.variant & HNET_VARIANT_CHAIN_GRID_3) ? 3 : (g.variant & HNET_VARIANT_CHAIN_GRID_8) ? 8 : 256;
    // opt-in (it measured slower): block 4's warp + concat inside the block_4_0 + block_4_1 kernel (batch > 8): fp16-plane mode with the fast sampler
    c->warp_in = c->n_planes == 2 && c->fuse_b4 && c->x16_b4 && !c->warp_exact && (g.variant & HNET_VARIANT_WARP_FUSE);
    if (c->use_chain) {
        CK(hipMalloc((void**)&c->chain_sync, CH_AREAS * CH_SYNC_WORDS * sizeof(uint32_t)));
        CK(hipMemset(c->chain_sync, 0, CH_AREAS * CH_SYNC_WORDS * sizeof(uint32_t)));
        static const int chain_first[4] = {1, 4, 10, 17};
        ChainArgs* ca = c->chain_args;
        const bool chain_fc = !(g.variant & HNET_VARIANT_CHAIN_NO_FC);
        if (chain_fc) CK(dalloc(&c->fc_part, (size_t)CH_MAX_PAIRS * CH_FC_ITEMS * 8));
        for (int blk = 0; blk < 4; blk++) {
            for (int l = chain_first[blk], j = 0; l <= last[blk]; l++, j++) {
                ChainLayer& L = ca[blk].L[j];
                L.in = c->act16[l - 1]; L.in_plane = MB * c->act_count[l - 1];
                L.wfrag = c->chain_w[l]; L.bias = c->conv_b[l];
                L.out16 = c->act16[l]; L.out_plane = MB * c->act_count[l];
                L.out32 = c->act[l];
                if (!L.in || !L.wfrag || !(L.out16 || L.out32)) c->use_chain = false;      // (a layout this build does not expect: the launches)
            }
            ca[blk].flag = c->d_flag;
            ca[blk].fcw = nullptr; ca[blk].fc_part = nullptr;
            if (chain_fc && blk < 3 && c->fc_w[blk]) { ca[blk].fcw = c->fc_w[blk]; ca[blk].fc_part = c->fc_part; }      // the block-tail FC as partial sums of the last layer (chain_lat.h)
        }
    }
    c->ws_floats = (size_t)16 << 20;
    CK(dalloc(&c->ws, c->ws_floats + SPLITK_TICKETS));      // + the tile counters of the split-K launches (kernels.h): zero between launches
    CK(hipMemset(c->ws + c->ws_floats, 0, SPLITK_TICKETS * sizeof(uint32_t)));
    CK(dalloc(&c->hidden, MB * c->n_local * 512));
    if (c->s3) {
        CK(hipMalloc((void**)&c->feat16, 3 * MB * 5120 * 2));
        CK(hipMalloc((void**)&c->head_mask, MB * c->n_local * 2 * 640));
    }
    CK(dalloc(&c->Hm, MB * 9));
    CK(dalloc(&c->Hm2, MB * 9));
    CK(dalloc(&c->Htot, MB * 9));
    CK(dalloc(&c->mean_s, MB * c->n_local * 8));
    CK(dalloc(&c->logvar_s, MB * c->n_local * 8));
    CK(dalloc(&c->d_mean, MB * 8));
    CK(dalloc(&c->d_cov, MB * 64));
    CK(dalloc(&c->d_prior, MB * 8));
    CK(hipMalloc(&c->stage_prev, MB * NPIX * sizeof(float)));
    CK(hipMalloc(&c->stage_curr, MB * NPIX * sizeof(float)));
    if (g.emit_error_map) {
        CK(dalloc(&c->d_err, MB * NPIX));
        CK(dalloc(&c->d_err_u8, MB * NPIX));
    }
    CK(dalloc(&c->ring[0], (size_t)NPIX));
    CK(dalloc(&c->ring[1], (size_t)NPIX));
    build_stages(c, g.max_batch);

    // ---- warm-up forward on the reference's constant inputs (HomographyNet.cpp:28-45): 0.2 / 0.5 / prior 1.0
    {
        std::vector<float> i1(NPIX, 0.2f), i2(NPIX, 0.5f), pr(8, 1.0f);
        CK(hipMemcpy(c->stage_prev, i1.data(), NPIX * sizeof(float), hipMemcpyHostToDevice));
        CK(hipMemcpy(c->stage_curr, i2.data(), NPIX * sizeof(float), hipMemcpyHostToDevice));
        CK(hipMemcpy(c->d_prior, pr.data(), 8 * sizeof(float), hipMemcpyHostToDevice));
        FwdArgs a = {c->stage_prev, c->stage_curr, HNET_PIX_F32, g.use_prior ? c->d_prior : nullptr, 1, 0,
                     c->d_mean, c->d_cov, c->d_err, nullptr, nullptr, nullptr, nullptr, false};
        auto t0 = std::chrono::steady_clock::now();
        int rc = forward(c, a, c->stream);
        if (rc == HNET_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = HNET_ERR_DEVICE;
        if (rc != HNET_OK) {
            fprintf(stderr, "hnet_create: warm-up forward failed: %s\n", c->err.c_str());
            hnet_destroy(c);
            return rc;
        }
        c->timing.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
#undef CK
    *out = c;
    return HNET_OK;
}

}  // namespace

extern "C" {

void hnet_default_config(hnet_config* cfg) {
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = sizeof(*cfg);
    cfg->blocks_to_run = 3;
    cfg->mc_samples = 16;        // model_to_trace.py:202
    cfg->dropout_p = 0.05f;      // trace_model.py:16
    cfg->precision = HNET_PREC_F16X2;    // fp32-grade results on the fp16 matrix cores, three MFMAs per product (same parity tests as HNET_PREC_BF16X3 / FP32)
    cfg->max_batch = 1;
}

int hnet_create_from_memory(const hnet_config* cfg, const void* blob, size_t len, hnet_ctx** out) {
    if (!blob) return HNET_ERR_INVALID_ARG;
    return create_impl(cfg, (const uint8_t*)blob, len, out);
}

int hnet_create(const hnet_config* cfg, const char* weights_path, hnet_ctx** out) {
    if (!weights_path) return HNET_ERR_INVALID_ARG;
    FILE* f = fopen(weights_path, "rb");
    if (!f) {   // the reference only prints on a load failure (HomographyNet.cpp:91-93); here it is an error code
        fprintf(stderr, "hnet_create: cannot open weights file %s\n", weights_path);
        return HNET_ERR_BAD_WEIGHTS;
    }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(n > 0 ? n : 0);
    size_t got = n > 0 ? fread(buf.data(), 1, n, f) : 0;
    fclose(f);
    if ((long)got != n) return HNET_ERR_BAD_WEIGHTS;
    return create_impl(cfg, buf.data(), buf.size(), out);
}

// ---- context groups
static int create_group_impl(const hnet_config* cfg, const uint8_t* blob, size_t len, int n_ctx, hnet_group** out) {
    if (!cfg || !out || n_ctx < 1 || n_ctx > HNET_GROUP_MAX) return HNET_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device_id < 0 || cfg->device_id >= ndev) return HNET_ERR_DEVICE;
    if (hipSetDevice(cfg->device_id) != hipSuccess) return HNET_ERR_DEVICE;
    hnet_group* g = new hnet_group();
    g->device_id = cfg->device_id;
    // The streams FIRST, in a fixed order, before any other stream of the group exists, and each on its own priority level as far as the device has levels: the
    // HIP runtime multiplexes the streams of ONE priority over its hardware queues (which queue a stream gets depends on every stream the process created before),
    // and two member streams that share a queue serialise their steps; queues of different priority are never shared.
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);          // numerically: greatest <= least (lower = more urgent)
    const int levels = least - greatest + 1;
    for (int i = 0; i < n_ctx; i++) {
        hipStream_t st = nullptr;
        const int prio = levels > 1 ? greatest + (i % levels) : 0;
        if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio) != hipSuccess) { hnet_destroy_group(g); return HNET_ERR_DEVICE; }
        g->streams.push_back(st);
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { hnet_destroy_group(g); return HNET_ERR_DEVICE; }
        g->ev.push_back(e);
    }
    for (int i = 0; i < n_ctx; i++) {
        hnet_ctx* c = nullptr;
        const int rc = create_impl(cfg, blob, len, &c, g->streams[i]);
        if (rc != HNET_OK) { hnet_destroy_group(g); return rc; }
        g->ctx.push_back(c);
    }
    *out = g;
    return HNET_OK;
}

int hnet_create_group_from_memory(const hnet_config* cfg, const void* blob, size_t len, int n_ctx, hnet_group** out) {
    if (!blob) return HNET_ERR_INVALID_ARG;
    return create_group_impl(cfg, (const uint8_t*)blob, len, n_ctx, out);
}

int hnet_create_group(const hnet_config* cfg, const char* weights_path, int n_ctx, hnet_group** out) {
    if (!weights_path) return HNET_ERR_INVALID_ARG;
    FILE* f = fopen(weights_path, "rb");
    if (!f) { fprintf(stderr, "hnet_create_group: cannot open weights file %s\n", weights_path); return HNET_ERR_BAD_WEIGHTS; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(n > 0 ? n : 0);
    size_t got = n > 0 ? fread(buf.data(), 1, n, f) : 0;
    fclose(f);
    if ((long)got != n) return HNET_ERR_BAD_WEIGHTS;
    return create_group_impl(cfg, buf.data(), buf.size(), n_ctx, out);
}

void hnet_destroy_group(hnet_group* g) {
    if (!g) return;
    (void)hipSetDevice(g->device_id);
    for (hnet_ctx* c : g->ctx) hnet_destroy(c);                          // (synchronises the member's stream; the streams are the group's)
    for (hipEvent_t e : g->ev) if (e) (void)hipEventDestroy(e);
    for (hipStream_t s : g->streams) if (s) (void)hipStreamDestroy(s);
    delete g;
}

int hnet_group_size(const hnet_group* g) { return g ? (int)g->ctx.size() : 0; }
hnet_ctx* hnet_group_context(hnet_group* g, int i) { return (g && i >= 0 && i < (int)g->ctx.size()) ? g->ctx[i] : nullptr; }
void* hnet_group_stream(hnet_group* g, int i) { return (g && i >= 0 && i < (int)g->streams.size()) ? (void*)g->streams[i] : nullptr; }
const char* hnet_group_last_error(const hnet_group* g) { return g ? g->err.c_str() : ""; }

int hnet_group_infer_batch_packed_device(hnet_group* g, const void* d_prev, const void* d_curr, int pix_fmt, const float* d_prior, int batch, uint64_t pair_seq0,
                                         float* d_out72, float* d_err_map, int* member) {
    if (!g || g->ctx.empty()) return HNET_ERR_INVALID_ARG;
    const int i = (int)(g->next % g->ctx.size());
    const int rc = hnet_infer_batch_packed_device(g->ctx[i], d_prev, d_curr, pix_fmt, d_prior, batch, pair_seq0, d_out72, d_err_map, nullptr);
    if (rc != HNET_OK) { g->err = g->ctx[i]->err; return rc; }
    g->next++;
    if (member) *member = i;
    return HNET_OK;
}

int hnet_group_join(hnet_group* g, void* stream) {
    if (!g || !stream) return HNET_ERR_INVALID_ARG;
    if (hipSetDevice(g->device_id) != hipSuccess) return HNET_ERR_DEVICE;
    for (size_t i = 0; i < g->streams.size(); i++) {
        if (hipEventRecord(g->ev[i], g->streams[i]) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, g->ev[i], 0) != hipSuccess) {
            g->err = "hnet_group_join: event";
            return HNET_ERR_DEVICE;
        }
    }
    return HNET_OK;
}

int hnet_group_synchronize(hnet_group* g) {
    if (!g) return HNET_ERR_INVALID_ARG;
    if (hipSetDevice(g->device_id) != hipSuccess) return HNET_ERR_DEVICE;
    for (hipStream_t s : g->streams)
        if (hipStreamSynchronize(s) != hipSuccess) { g->err = "hipStreamSynchronize"; return HNET_ERR_DEVICE; }
    return HNET_OK;
}

int hnet_group_overflow_flag(hnet_group* g, int* flags) {
    if (!g || !flags) return HNET_ERR_INVALID_ARG;
    int all = 0;
    for (hnet_ctx* c : g->ctx) {
        int f = 0;
        const int rc = hnet_overflow_flag(c, nullptr, &f);
        if (rc != HNET_OK) { g->err = c->err; return rc; }
        all |= f;
    }
    *flags = all;
    return HNET_OK;
}

void hnet_destroy(hnet_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device_id);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    auto fr = [](void* p) { if (p) (void)hipFree(p); };
    for (int l = 0; l < 20; l++) { fr(c->patch_frag[l]); fr(c->conv_w[l]); fr(c->conv_b[l]); fr(c->act[l]); fr(c->conv_w16[l]); fr(c->conv_wfrag[l]); fr(c->act16[l]); }
    for (int k = 0; k < 3; k++) { fr(c->fc_w[k]); fr(c->fc_b[k]); }
    for (int k = 0; k < 4; k++) fr(c->x_in[k]);
    for (int i = 0; i < 2; i++) if (c->g_infer[i]) (void)hipGraphExecDestroy(c->g_infer[i]);
    if (c->g_batch) (void)hipGraphExecDestroy(c->g_batch);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (int i = 0; i < 2; i++) {
        if (c->pinned_img[i]) (void)hipHostFree(c->pinned_img[i]);
        if (c->ev_img[i]) (void)hipEventDestroy(c->ev_img[i]);
    }
    fr(c->d_seq); fr(c->d_flag); fr(c->chain_sync); fr(c->fc_part);
    for (int l = 0; l < 20; l++) fr(c->chain_w[l]);
    fr(c->und_map[0]); fr(c->und_map[1]); fr(c->raw_dev);
    fr(c->s2_frag[0]); fr(c->s2_frag[3]); fr(c->x16_b4); fr(c->b30_frag); fr(c->b40_frag); fr(c->b41_frag); fr(c->w1_16); fr(c->b3f_w1); fr(c->b42_w2); fr(c->b42_w3); fr(c->feat16); fr(c->head_mask);
    fr(c->ws); fr(c->w1); fr(c->b1); fr(c->w2); fr(c->b2); fr(c->hidden); fr(c->Hm); fr(c->Hm2); fr(c->Htot); fr(c->mean_s); fr(c->logvar_s);
    fr(c->d_mean); fr(c->d_cov); fr(c->d_err); fr(c->d_err_u8); fr(c->d_prior); fr(c->stage_prev); fr(c->stage_curr);
    fr(c->ring[0]); fr(c->ring[1]);
    for (auto e : c->prof_ev) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream && c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* hnet_status_string(int s) { return (s >= 0 && s <= 6) ? kStatus[s] : "unknown status"; }
const char* hnet_last_error(const hnet_ctx* c) { return c ? c->err.c_str() : ""; }
const char* hnet_version(void) { return "hnet-hip 0.1.0 (gfx950)"; }

int hnet_push_image(hnet_ctx* c, const uint8_t* data, int rows, int cols, int row_stride, double t) {
    if (!c || !data) return HNET_ERR_INVALID_ARG;
    if (rows != IMG_H || cols != IMG_W || row_stride < cols) return fail(c, HNET_ERR_INVALID_ARG, "image must be 224x320 8-bit");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    c->img_counter++;                                               // HomographyNet.cpp:134
    const int slot = c->img_counter == 1 ? 0 : (c->curr_slot ^ 1);  // prev <- curr: flip the ring instead of cloning (:143)
    // `data` is not retained (the cv::Mat may be reused): copy it into a pinned staging frame and upload from there without
    // waiting; network_inference follows on the same stream.  The staging frame of this slot was last used two pushes ago.
    HIPCHK(c, hipEventSynchronize(c->ev_img[slot]));
    for (int r = 0; r < IMG_H; r++) memcpy(c->pinned_img[slot] + (size_t)r * IMG_W, data + (size_t)r * row_stride, IMG_W);
    HIPCHK(c, hipMemcpyAsync(c->ring[slot], c->pinned_img[slot], NPIX, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_img[slot], c->stream));
    c->curr_slot = slot;
    if (c->img_counter >= 2) c->latest_t = t;                       // :148
    return HNET_OK;
}

// ---- image pre-processing: CamBase.h:165-186
int hnet_set_undistort_maps(hnet_ctx* c, const float* map_x, const float* map_y, int raw_rows, int raw_cols) {
    if (!c || !map_x || !map_y || raw_rows < 1 || raw_cols < 1 || raw_rows > 16384 || raw_cols > 16384) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    for (int i = 0; i < 2; i++)
        if (!c->und_map[i]) HIPCHK(c, dalloc(&c->und_map[i], (size_t)NPIX));
    HIPCHK(c, hipMemcpy(c->und_map[0], map_x, NPIX * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->und_map[1], map_y, NPIX * 4, hipMemcpyHostToDevice));
    if (c->raw_dev && (size_t)raw_rows * raw_cols > (size_t)c->raw_rows * c->raw_cols) { (void)hipFree(c->raw_dev); c->raw_dev = nullptr; }
    if (!c->raw_dev) HIPCHK(c, hipMalloc((void**)&c->raw_dev, (size_t)raw_rows * raw_cols));
    c->raw_rows = raw_rows;
    c->raw_cols = raw_cols;
    return HNET_OK;
}

int hnet_set_camera(hnet_ctx* c, const hnet_camera* cam) {
    if (!c || !cam) return HNET_ERR_INVALID_ARG;
    // the virtual camera every frame is resampled to: 90 deg horizontal field of view on 320 px (CamBase.h:166-169)
    const double f = (IMG_W - 1.0) / 2.0 / std::tan(45.0 / 180.0 * (2.0 * std::acos(0.0)));
    const double cx = (IMG_W - 1.0) / 2.0, cy = (IMG_H - 1.0) / 2.0;
    std::vector<float> mx(NPIX), my(NPIX);
    for (int v = 0; v < IMG_H; v++)
        for (int u = 0; u < IMG_W; u++) {
            const double x = (u - cx) / f, y = (v - cy) / f;          // R = I: the pixel's ray in the virtual camera
            double xd, yd;
            if (cam->fisheye) {                                      // cv::fisheye::initUndistortRectifyMap (equidistant)
                const double r = std::sqrt(x * x + y * y), th = std::atan(r), t2 = th * th;
                const double thd = th * (1.0 + t2 * (cam->d[0] + t2 * (cam->d[1] + t2 * (cam->d[2] + t2 * cam->d[3]))));
                const double sc = r == 0.0 ? 1.0 : thd / r;
                xd = x * sc; yd = y * sc;
            } else {                                                 // cv::initUndistortRectifyMap, D = (k1, k2, p1, p2)
                const double r2 = x * x + y * y, kr = 1.0 + r2 * (cam->d[0] + r2 * cam->d[1]);
                xd = x * kr + 2.0 * cam->d[2] * x * y + cam->d[3] * (r2 + 2.0 * x * x);
                yd = y * kr + cam->d[2] * (r2 + 2.0 * y * y) + 2.0 * cam->d[3] * x * y;
            }
            mx[v * IMG_W + u] = (float)(cam->k[0] * xd + cam->k[2]);
            my[v * IMG_W + u] = (float)(cam->k[1] * yd + cam->k[3]);
        }
    return hnet_set_undistort_maps(c, mx.data(), my.data(), cam->raw_rows, cam->raw_cols);
}

int hnet_get_undistort_maps(hnet_ctx* c, float* map_x, float* map_y) {
    if (!c || !map_x || !map_y) return HNET_ERR_INVALID_ARG;
    if (!c->und_map[0]) return fail(c, HNET_ERR_NOT_READY, "no camera set");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    HIPCHK(c, hipMemcpy(map_x, c->und_map[0], NPIX * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(map_y, c->und_map[1], NPIX * 4, hipMemcpyDeviceToHost));
    return HNET_OK;
}

static int remap_raw(hnet_ctx* c, const uint8_t* raw, int rows, int cols, int row_stride, uint8_t* d_out) {
    if (!c->und_map[0]) return fail(c, HNET_ERR_NOT_READY, "hnet_set_camera / hnet_set_undistort_maps first");
    if (rows != c->raw_rows || cols != c->raw_cols || row_stride < cols) return fail(c, HNET_ERR_INVALID_ARG, "raw image size differs from the camera's");
    HIPCHK(c, hipMemcpy2DAsync(c->raw_dev, cols, raw, row_stride, cols, rows, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_undistort(c->raw_dev, rows, cols, cols, c->und_map[0], c->und_map[1], d_out, c->stream));
    return HNET_OK;
}

int hnet_push_raw_image(hnet_ctx* c, const uint8_t* raw, int rows, int cols, int row_stride, double t) {
    if (!c || !raw) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const int slot = c->img_counter == 0 ? 0 : (c->curr_slot ^ 1);
    const int rc = remap_raw(c, raw, rows, cols, row_stride, c->ring[slot]);
    if (rc != HNET_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));                     // `raw` is not retained
    c->img_counter++;
    c->curr_slot = slot;
    if (c->img_counter >= 2) c->latest_t = t;
    return HNET_OK;
}

int hnet_op_undistort(hnet_ctx* c, const uint8_t* raw, int rows, int cols, int row_stride, uint8_t* out) {
    if (!c || !raw || !out) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    DevTemps t;
    uint8_t* d_o = nullptr;
    HIPCHK(c, t.alloc(&d_o, (size_t)NPIX));
    const int rc = remap_raw(c, raw, rows, cols, row_stride, d_o);
    if (rc == HNET_OK) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(out, d_o, NPIX, hipMemcpyDeviceToHost));
    }
    return rc;
}

int hnet_image_count(const hnet_ctx* c) { return c ? (c->img_src ? c->img_src : c)->img_counter : 0; }
double hnet_latest_time(const hnet_ctx* c) { return c ? (c->img_src ? c->img_src : c)->latest_t : -1.0; }

// `main_model`: the call the reference times (num_of_inference == 0, HomographyNet.cpp:174-189); IEKF re-runs (iteration > 0)
// advance the mask sequence number (n_inferences) but not the timing statistics (:245-251 sit under `num_of_inference == 0`)
static void note_timing(hnet_ctx* c, float dev_ms, double host_ms, bool main_model = true) {
    c->timing.device_ms = dev_ms;
    c->timing.host_ms = host_ms;
    c->timing.n_inferences++;
    if (main_model) {
        c->timing.n_main_inferences++;                                                    // inference_counting, :189
        if (c->timing.n_main_inferences > 100) c->timing.sum_device_ms_after_100 += dev_ms;   // :245-251
    }
}

// HNET_PREC_F16X2 carries activations in fp16 planes (|a| < 32768 guaranteed, s3_format.h).  An overflow turns into infinities / NaNs that reach the outputs; the
// host-result entry points then re-pack the weights for HNET_PREC_BF16X3 (fp32 range, same kernels in their six-product form), run the call
// again and stay in that mode: a finite answer of the reference is never lost to the faster arithmetic.
static bool all_finite(const float* v, size_t n) {
    for (size_t i = 0; i < n; i++) if (!std::isfinite(v[i])) return false;
    return true;
}
static int demote_to_bf16x3(hnet_ctx* c) {
    if (c->n_planes != 2 || c->blob_copy.empty()) return HNET_ERR_UNSUPPORTED;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->ws) HIPCHK(c, hipMemset(c->ws + c->ws_floats, 0, SPLITK_TICKETS * sizeof(uint32_t)));      // the overflowed forward may have ended anywhere: counters back to zero
    Blob b;
    if (!parse_blob(c->blob_copy.data(), c->blob_copy.size(), b)) return fail(c, HNET_ERR_BAD_WEIGHTS, "weight blob");
    c->n_planes = 3;
    c->use_chain = false;
    c->warp_in = false;
    c->fuse_b3 = c->fuse_b42 = c->a14_pad = false;    // the fused block-3 / block_4_2+4_3 kernels exist for the fp16 planes only (their layers' buffers stay allocated; act16[14] goes back to the plain layout)
    c->cfg.precision = HNET_PREC_BF16X3;
    const int rc = upload_weights(c, b);
    if (rc != HNET_OK) return fail(c, rc, "re-packing the weights for HNET_PREC_BF16X3");
    for (int i = 0; i < 2; i++) if (c->g_infer[i]) { (void)hipGraphExecDestroy(c->g_infer[i]); c->g_infer[i] = nullptr; }   // captured with the old kernels
    if (c->g_batch) { (void)hipGraphExecDestroy(c->g_batch); c->g_batch = nullptr; }
    c->blob_copy.clear();
    c->blob_copy.shrink_to_fit();
    build_stages(c, c->cfg.max_batch);     // the fused fp16-plane kernels are gone from the launch list
    fprintf(stderr, "hnet: activation beyond the fp16 range in HNET_PREC_F16X2: context demoted to HNET_PREC_BF16X3\n");
    return HNET_OK;
}
// A bounded spin of a one-XCD tail chain gave up (chain_lat.h CH_FLAG_TIMEOUT in the flag word: a claimed item never completed - not observed; the bound exists so
// that a scheduling surprise ends in a flagged forward instead of a hung device).  The context goes back to the launches for good; the host entry points repeat the call.
static int chain_gave_up(hnet_ctx* c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->use_chain = false;
    if (c->chain_sync) HIPCHK(c, hipMemset(c->chain_sync, 0, CH_AREAS * CH_SYNC_WORDS * sizeof(uint32_t)));
    for (int i = 0; i < 2; i++) if (c->g_infer[i]) { (void)hipGraphExecDestroy(c->g_infer[i]); c->g_infer[i] = nullptr; }
    if (c->g_batch) { (void)hipGraphExecDestroy(c->g_batch); c->g_batch = nullptr; }
    build_stages(c, c->cfg.max_batch);
    fprintf(stderr, "hnet: a tail-chain launch timed out on its in-launch hand-off: this context uses the per-layer launches from now on\n");
    return HNET_OK;
}
int hnet_precision(const hnet_ctx* c) { return c ? c->cfg.precision : -1; }
int hnet_get_config(const hnet_ctx* c, hnet_config* out) {
    if (!c || !out) return HNET_ERR_INVALID_ARG;
    *out = c->cfg;
    return HNET_OK;
}

int hnet_overflow_flag(hnet_ctx* c, void* stream, int* flags) {
    if (!c || !flags) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    uint32_t v = 0;
    HIPCHK(c, hipMemcpyAsync(&v, c->d_flag, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemsetAsync(c->d_flag, 0, 4, s));
    HIPCHK(c, hipStreamSynchronize(s));
    *flags = (int)v;
    return HNET_OK;
}
// A non-finite output only means "activation beyond the fp16-plane range" when the inputs were finite: a NaN prior of a diverged filter or a
// NaN float image gives NaN outputs in every arithmetic (the reference's too) and must not cost the context its mode.
static bool prior_finite(const double* p, size_t n) {
    for (size_t i = 0; p && i < n; i++) if (!std::isfinite(p[i])) return false;
    return true;
}

int hnet_infer(hnet_ctx* c, const double* prior_px, int iteration, float mean_out[8], float cov_out[64], uint8_t* err_map_out) {
    if (!c || !mean_out || !cov_out) return HNET_ERR_INVALID_ARG;
    hnet_ctx* src = c->img_src ? c->img_src : c;                 // the context that owns the frames and the sequence count (hnet_attach_images)
    if (src->img_counter < 2) return fail(c, HNET_ERR_NOT_READY, "HNet cannot inference! Only has one image!");   // :155-158
    if (err_map_out && !c->cfg.emit_error_map) return fail(c, HNET_ERR_INVALID_ARG, "context was created without emit_error_map");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    auto t0 = std::chrono::steady_clock::now();
    if (c->cfg.use_prior && !prior_px) return fail(c, HNET_ERR_INVALID_ARG, "prior required");
    if (src != c) HIPCHK(c, hipStreamWaitEvent(c->stream, src->ev_img[src->curr_slot], 0));      // the frame's upload runs on the source's stream
    if (c->use_graph) {
        // one graph per ring orientation.  Round 6: no memcpy nodes - the pinned host block is mapped into the device's address space and the kernels read the
        // sequence number and the prior from it and write mean, covariance, error map and the flag word to it (six copy / memset nodes of 3 - 8 us each in a chain of
        // 16 launches; HNET_VARIANT_GRAPH_COPIES keeps them: H2D {seq, prior} -> forward -> D2H {mean, cov, err, flag})
        const int slot = src->curr_slot;
        hnet_ctx::Pinned* pin = c->pinned;
        if (!c->g_infer[slot]) {
            hnet_ctx::Pinned* dpin = nullptr;
            if (c->graph_zero_copy && hipHostGetDevicePointer((void**)&dpin, pin, 0) != hipSuccess) { (void)hipGetLastError(); c->graph_zero_copy = false; }
            c->g_infer[slot] = capture_graph(c, [&]() -> int {
                if (c->graph_zero_copy) {
                    FwdArgs ga = {src->ring[slot ^ 1], src->ring[slot], HNET_PIX_U8, c->cfg.use_prior ? dpin->prior : nullptr, 1, 0, dpin->mean, dpin->cov,
                                  nullptr, c->cfg.emit_error_map ? dpin->err : nullptr, nullptr, nullptr, nullptr, false};
                    ga.seq_dev = &dpin->seq;
                    ga.flag = &dpin->flag;
                    return forward(c, ga, c->stream);
                }
                if (hipMemcpyAsync(c->d_seq, &pin->seq, 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) return HNET_ERR_DEVICE;
                if (c->cfg.use_prior && hipMemcpyAsync(c->d_prior, pin->prior, 32, hipMemcpyHostToDevice, c->stream) != hipSuccess) return HNET_ERR_DEVICE;
                FwdArgs ga = {src->ring[slot ^ 1], src->ring[slot], HNET_PIX_U8, c->cfg.use_prior ? c->d_prior : nullptr, 1, 0, c->d_mean, c->d_cov,
                              nullptr, c->cfg.emit_error_map ? c->d_err_u8 : nullptr, nullptr, nullptr, nullptr, false};
                ga.seq_dev = c->d_seq;
                const int rc = forward(c, ga, c->stream);
                if (rc != HNET_OK) return rc;
                if (hipMemcpyAsync(pin->mean, c->d_mean, 32, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return HNET_ERR_DEVICE;
                if (hipMemcpyAsync(pin->cov, c->d_cov, 256, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return HNET_ERR_DEVICE;
                if (c->cfg.emit_error_map && hipMemcpyAsync(pin->err, c->d_err_u8, NPIX, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return HNET_ERR_DEVICE;
                if (hipMemcpyAsync(&pin->flag, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return HNET_ERR_DEVICE;
                // the host entry points inspect (and, for an overflow, repair) their own results: the device flag is for the device-resident entry points only
                if (hipMemsetAsync(c->d_flag, 0, 4, c->stream) != hipSuccess) return HNET_ERR_DEVICE;
                return HNET_OK;
            });
            if (!c->g_infer[slot]) c->use_graph = false;      // capture unavailable: eager path below, same kernels
            c->g_infer_H[slot] = c->H_last;
        }
        if (c->g_infer[slot]) {
            c->H_last = c->g_infer_H[slot];
            pin->seq = (uint64_t)src->timing.n_inferences;
            if (c->cfg.use_prior) for (int i = 0; i < 8; i++) pin->prior[i] = (float)prior_px[i];    // :160-165 toType(kFloat)
            if (c->graph_zero_copy) pin->flag = 0;                 // (raised by the kernels directly; the memcpy form overwrites it)
            HIPCHK(c, hipEventRecord(c->ev0, c->stream));
            HIPCHK(c, hipGraphLaunch(c->g_infer[slot], c->stream));
            HIPCHK(c, hipEventRecord(c->ev1, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (pin->flag & CH_FLAG_TIMEOUT) {
                const int rd = chain_gave_up(c);
                return rd != HNET_OK ? rd : hnet_infer(c, prior_px, iteration, mean_out, cov_out, err_map_out);
            }
            if (c->n_planes == 2 && !(all_finite(pin->mean, 8) && all_finite(pin->cov, 64)) && prior_finite(c->cfg.use_prior ? prior_px : nullptr, 8)) {
                const int rd = demote_to_bf16x3(c);
                return rd != HNET_OK ? rd : hnet_infer(c, prior_px, iteration, mean_out, cov_out, err_map_out);
            }
            memcpy(mean_out, pin->mean, 32);
            memcpy(cov_out, pin->cov, 256);
            if (err_map_out) memcpy(err_map_out, pin->err, NPIX);
            float gms = 0;
            HIPCHK(c, hipEventElapsedTime(&gms, c->ev0, c->ev1));
            note_timing(c, gms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), iteration == 0);
            if (src != c) src->timing.n_inferences++;             // one shared sequence count
            return HNET_OK;
        }
    }
    if (c->cfg.use_prior) {
        float pf[8];
        for (int i = 0; i < 8; i++) pf[i] = (float)prior_px[i];    // :160-165 toType(kFloat)
        HIPCHK(c, hipMemcpyAsync(c->d_prior, pf, sizeof pf, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    FwdArgs a = {src->ring[src->curr_slot ^ 1], src->ring[src->curr_slot], HNET_PIX_U8, c->cfg.use_prior ? c->d_prior : nullptr, 1,
                 (uint64_t)src->timing.n_inferences, c->d_mean, c->d_cov, nullptr, err_map_out ? c->d_err_u8 : nullptr,
                 nullptr, nullptr, nullptr, false};
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    int rc = forward(c, a, c->stream);
    if (rc != HNET_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipMemcpyAsync(mean_out, c->d_mean, 8 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cov_out, c->d_cov, 64 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (err_map_out) HIPCHK(c, hipMemcpyAsync(err_map_out, c->d_err_u8, NPIX, hipMemcpyDeviceToHost, c->stream));
    uint32_t flag_now = 0;
    HIPCHK(c, hipMemcpyAsync(&flag_now, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_flag, 0, 4, c->stream));      // (see the graph path)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (flag_now & CH_FLAG_TIMEOUT) {
        const int rd = chain_gave_up(c);
        return rd != HNET_OK ? rd : hnet_infer(c, prior_px, iteration, mean_out, cov_out, err_map_out);
    }
    if (c->n_planes == 2 && !(all_finite(mean_out, 8) && all_finite(cov_out, 64)) && prior_finite(c->cfg.use_prior ? prior_px : nullptr, 8)) {
        const int rd = demote_to_bf16x3(c);
        return rd != HNET_OK ? rd : hnet_infer(c, prior_px, iteration, mean_out, cov_out, err_map_out);
    }
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    note_timing(c, ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), iteration == 0);
    if (src != c) src->timing.n_inferences++;
    return HNET_OK;
}

int hnet_attach_images(hnet_ctx* c, hnet_ctx* source) {
    if (!c || !source || c == source || source->img_src) return HNET_ERR_INVALID_ARG;
    if (c->cfg.device_id != source->cfg.device_id) return fail(c, HNET_ERR_INVALID_ARG, "hnet_attach_images: both contexts must live on one device");
    c->img_src = source;
    return HNET_OK;
}

int hnet_infer_batch_device(hnet_ctx* c, const void* d_prev, const void* d_curr, int pix_fmt, const float* d_prior, int batch,
                            uint64_t pair_seq0, float* d_mean, float* d_cov, float* d_err_map, void* stream) {
    if (!c || !d_prev || !d_curr || !d_mean || !d_cov) return HNET_ERR_INVALID_ARG;
    if (pix_fmt != HNET_PIX_U8 && pix_fmt != HNET_PIX_F32) return fail(c, HNET_ERR_INVALID_ARG, "pix_fmt");
    if (d_err_map && !c->cfg.emit_error_map) return fail(c, HNET_ERR_INVALID_ARG, "context was created without emit_error_map");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    FwdArgs a = {d_prev, d_curr, pix_fmt, d_prior, batch, pair_seq0, d_mean, d_cov, d_err_map, nullptr, nullptr, nullptr, nullptr, false};
    return forward(c, a, stream ? (hipStream_t)stream : c->stream);
}

int hnet_infer_batch_packed_device(hnet_ctx* c, const void* d_prev, const void* d_curr, int pix_fmt, const float* d_prior, int batch,
                                   uint64_t pair_seq0, float* d_out72, float* d_err_map, void* stream) {
    if (!c || !d_prev || !d_curr || !d_out72) return HNET_ERR_INVALID_ARG;
    if (pix_fmt != HNET_PIX_U8 && pix_fmt != HNET_PIX_F32) return fail(c, HNET_ERR_INVALID_ARG, "pix_fmt");
    if (d_err_map && !c->cfg.emit_error_map) return fail(c, HNET_ERR_INVALID_ARG, "context was created without emit_error_map");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    FwdArgs a = {d_prev, d_curr, pix_fmt, d_prior, batch, pair_seq0, d_out72, d_out72 + 8, d_err_map, nullptr, nullptr, nullptr, nullptr, false};
    a.mean_stride = a.cov_stride = HNET_PACKED_FLOATS;
    return forward(c, a, stream ? (hipStream_t)stream : c->stream);
}

int hnet_infer_batch(hnet_ctx* c, const void* prev, const void* curr, int pix_fmt, const float* prior, int batch,
                     uint64_t pair_seq0, float* mean, float* cov, float* err_map) {
    if (!c || !prev || !curr || !mean || !cov) return HNET_ERR_INVALID_ARG;
    if (pix_fmt != HNET_PIX_U8 && pix_fmt != HNET_PIX_F32) return fail(c, HNET_ERR_INVALID_ARG, "pix_fmt");
    if (batch < 1) return fail(c, HNET_ERR_INVALID_ARG, "batch < 1");
    if (batch > c->cfg.max_batch) return fail(c, HNET_ERR_CAPACITY, "batch exceeds max_batch");
    if (err_map && !c->cfg.emit_error_map) return fail(c, HNET_ERR_INVALID_ARG, "context was created without emit_error_map");
    if (c->cfg.use_prior && !prior) return fail(c, HNET_ERR_INVALID_ARG, "prior required");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    auto t0 = std::chrono::steady_clock::now();
    const size_t px = pix_fmt == HNET_PIX_U8 ? 1 : 4;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->stage_prev, prev, (size_t)batch * NPIX * px, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->stage_curr, curr, (size_t)batch * NPIX * px, hipMemcpyHostToDevice, s));
    if (prior) HIPCHK(c, hipMemcpyAsync(c->d_prior, prior, (size_t)batch * 8 * sizeof(float), hipMemcpyHostToDevice, s));
    FwdArgs a = {c->stage_prev, c->stage_curr, pix_fmt, prior ? c->d_prior : nullptr, batch, pair_seq0, c->d_mean, c->d_cov,
                 err_map ? c->d_err : nullptr, nullptr, nullptr, nullptr, nullptr, false};
    HIPCHK(c, hipEventRecord(c->ev0, s));
    int rc = forward(c, a, s);
    if (rc != HNET_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, s));
    HIPCHK(c, hipMemcpyAsync(mean, c->d_mean, (size_t)batch * 8 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(cov, c->d_cov, (size_t)batch * 64 * sizeof(float), hipMemcpyDeviceToHost, s));
    if (err_map) HIPCHK(c, hipMemcpyAsync(err_map, c->d_err, (size_t)batch * NPIX * sizeof(float), hipMemcpyDeviceToHost, s));
    uint32_t flag_now = 0;
    HIPCHK(c, hipMemcpyAsync(&flag_now, c->d_flag, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemsetAsync(c->d_flag, 0, 4, s));              // host results are inspected below: hnet_overflow_flag reports device-resident batches only
    HIPCHK(c, hipStreamSynchronize(s));
    if (flag_now & CH_FLAG_TIMEOUT) {
        const int rd = chain_gave_up(c);
        return rd != HNET_OK ? rd : hnet_infer_batch(c, prev, curr, pix_fmt, prior, batch, pair_seq0, mean, cov, err_map);
    }
    if (c->n_planes == 2 && !(all_finite(mean, (size_t)batch * 8) && all_finite(cov, (size_t)batch * 64)) &&
        (!prior || all_finite(prior, (size_t)batch * 8)) &&
        (pix_fmt == HNET_PIX_U8 || (all_finite((const float*)prev, (size_t)batch * NPIX) && all_finite((const float*)curr, (size_t)batch * NPIX)))) {
        const int rd = demote_to_bf16x3(c);
        return rd != HNET_OK ? rd : hnet_infer_batch(c, prev, curr, pix_fmt, prior, batch, pair_seq0, mean, cov, err_map);
    }
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    note_timing(c, ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return HNET_OK;
}

int hnet_infer_mc_partial_device(hnet_ctx* c, const void* d_prev, const void* d_curr, int pix_fmt, const float* d_prior,
                                 int batch, uint64_t pair_seq0, float* d_mean_s, float* d_logvar_s, float* d_h_part1, void* stream) {
    if (!c || !d_prev || !d_curr || !d_mean_s || !d_logvar_s) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    FwdArgs a = {d_prev, d_curr, pix_fmt, d_prior, batch, pair_seq0, nullptr, nullptr, nullptr, nullptr, d_mean_s, d_logvar_s, d_h_part1, true};
    return forward(c, a, stream ? (hipStream_t)stream : c->stream);
}

int hnet_mc_finish_device(hnet_ctx* c, const float* d_mean_s, const float* d_logvar_s, int n_total, const float* d_h_part1,
                          int batch, float* d_mean, float* d_cov, void* stream) {
    if (!c || !d_mean_s || !d_logvar_s || !d_h_part1 || !d_mean || !d_cov || n_total < 1 || batch < 1) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    HIPCHK(c, launch_mc_finish(d_mean_s, d_logvar_s, n_total, d_h_part1, batch, d_mean, d_cov, nullptr,
                               stream ? (hipStream_t)stream : c->stream, c->d_flag));
    return HNET_OK;
}

int hnet_mc_finish_packed_device(hnet_ctx* c, const float* d_mean_s, const float* d_logvar_s, int n_total, const float* d_h_part1,
                                 int batch, float* d_out72, void* stream) {
    if (!c || !d_mean_s || !d_logvar_s || !d_h_part1 || !d_out72 || n_total < 1 || batch < 1) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    HIPCHK(c, launch_mc_finish(d_mean_s, d_logvar_s, n_total, d_h_part1, batch, d_out72, d_out72 + 8, nullptr,
                               stream ? (hipStream_t)stream : c->stream, c->d_flag, HNET_PACKED_FLOATS, HNET_PACKED_FLOATS));
    return HNET_OK;
}

int hnet_mc_finish_gathered_device(hnet_ctx* c, const float* d_gathered, int world, int n_local, const float* d_h_part1, int batch, float* d_out72, void* stream) {
    if (!c || !d_gathered || !d_h_part1 || !d_out72 || world < 1 || n_local < 1 || batch < 1) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const size_t block = (size_t)batch * n_local * 8;        // floats of one [B][n_local][8] array; a rank's message is [mean block | log-variance block]
    HIPCHK(c, launch_mc_finish(d_gathered, d_gathered + block, world * n_local, d_h_part1, batch, d_out72, d_out72 + 8, nullptr,
                               stream ? (hipStream_t)stream : c->stream, c->d_flag, HNET_PACKED_FLOATS, HNET_PACKED_FLOATS, n_local, 2 * block));
    return HNET_OK;
}

int hnet_synchronize(hnet_ctx* c, void* stream) {
    if (!c) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    HIPCHK(c, hipStreamSynchronize(stream ? (hipStream_t)stream : c->stream));
    return HNET_OK;
}

int hnet_last_timing(const hnet_ctx* c, hnet_timing* out) {
    if (!c || !out) return HNET_ERR_INVALID_ARG;
    *out = c->timing;
    return HNET_OK;
}

int hnet_time_batch_device(hnet_ctx* c, const void* d_prev, const void* d_curr, int pix_fmt, const float* d_prior, int batch,
                           uint64_t pair_seq0, float* d_mean, float* d_cov, int iters, float* per_iter_ms, float* total_ms) {
    if (!c || iters < 1) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    std::vector<hipEvent_t> ev(iters + 1);
    for (auto& e : ev) HIPCHK(c, hipEventCreate(&e));
    int rc = HNET_OK;
    // small batches: replay the forward as one hipGraph (what hnet_infer does); the mask sequence number is refreshed
    // from pinned memory once before the timed loop (every replay then uses the same masks - timing only)
    hipGraphExec_t gx = nullptr;
    if (c->graph_timing && c->use_graph && batch <= 8) {
        hnet_ctx::GraphKey key = {d_prev, d_curr, d_prior, d_mean, d_cov, batch, pix_fmt};
        if (!(c->g_batch && key == c->g_key)) {
            if (c->g_batch) { (void)hipGraphExecDestroy(c->g_batch); c->g_batch = nullptr; }
            c->g_batch = capture_graph(c, [&]() -> int {
                if (hipMemcpyAsync(c->d_seq, &c->pinned->seq, 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) return HNET_ERR_DEVICE;
                FwdArgs ga = {d_prev, d_curr, pix_fmt, d_prior, batch, 0, d_mean, d_cov, nullptr, nullptr, nullptr, nullptr, nullptr, false};
                ga.seq_dev = c->d_seq;
                return forward(c, ga, c->stream);
            });
            c->g_key = key;
            c->g_batch_H = c->H_last;
        }
        gx = c->g_batch;
        if (gx) { c->pinned->seq = pair_seq0; c->H_last = c->g_batch_H; }
    }
    HIPCHK(c, hipEventRecord(ev[0], c->stream));
    for (int i = 0; i < iters && rc == HNET_OK; i++) {
        if (gx) { if (hipGraphLaunch(gx, c->stream) != hipSuccess) rc = fail(c, HNET_ERR_DEVICE, "hipGraphLaunch"); }
        else rc = hnet_infer_batch_device(c, d_prev, d_curr, pix_fmt, d_prior, batch, pair_seq0 + (uint64_t)i * batch, d_mean, d_cov, nullptr, nullptr);
        if (rc == HNET_OK && hipEventRecord(ev[i + 1], c->stream) != hipSuccess) rc = fail(c, HNET_ERR_DEVICE, "hipEventRecord");
    }
    if (rc == HNET_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, HNET_ERR_DEVICE, "hipStreamSynchronize");
    if (rc == HNET_OK) {
        float tot = 0;
        for (int i = 0; i < iters; i++) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            if (per_iter_ms) per_iter_ms[i] = ms;
            tot += ms;
        }
        if (total_ms) (void)hipEventElapsedTime(total_ms, ev[0], ev[iters]);
        (void)tot;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

/* ---- per-stage (per kernel launch) device timing with HIP events on the context stream ---- */
int hnet_stage_count(const hnet_ctx* c) { return c ? (int)c->stages.size() : 0; }
const char* hnet_stage_name(const hnet_ctx* c, int i) { return (c && i >= 0 && i < (int)c->stages.size()) ? c->stages[i].name.c_str() : ""; }
int hnet_stage_kernels(const hnet_ctx* c, int i) { return (c && i >= 0 && i < (int)c->stages.size()) ? c->stages[i].kernels : 0; }
double hnet_stage_flops_per_pair(const hnet_ctx* c, int i) { return (c && i >= 0 && i < (int)c->stages.size()) ? c->stages[i].flops_per_pair : 0.0; }

int hnet_profile_batch_device(hnet_ctx* c, const void* d_prev, const void* d_curr, int pix_fmt, const float* d_prior, int batch,
                              uint64_t pair_seq0, float* d_mean, float* d_cov, int iters, float* stage_ms_avg) {
    if (!c || iters < 1 || !stage_ms_avg) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    build_stages(c, batch, d_prev, d_curr, pix_fmt);   // the launches of a forward of THIS batch and THESE images (the latency path has fewer): hnet_stage_name follows
    const size_t ns = c->stages.size();
    std::vector<double> acc(ns, 0.0);
    // the per-stage events live in the context only for the duration of this call: whatever happens, they are destroyed
    // and the list is emptied again (a non-empty list makes every later forward single-stream and event-recording)
    struct Cleanup {
        hnet_ctx* c;
        ~Cleanup() { for (auto e : c->prof_ev) if (e) (void)hipEventDestroy(e); c->prof_ev.clear(); c->prof_pos = 0; }
    } cleanup{c};
    c->prof_ev.assign(ns, nullptr);
    for (auto& e : c->prof_ev) HIPCHK(c, hipEventCreate(&e));
    for (int it = 0; it < iters; it++) {
        c->prof_pos = 0;
        HIPCHK(c, hipEventRecord(c->ev0, c->stream));
        const int rc = hnet_infer_batch_device(c, d_prev, d_curr, pix_fmt, d_prior, batch, pair_seq0, d_mean, d_cov,
                                               c->cfg.emit_error_map ? c->d_err : nullptr, nullptr);
        if (rc != HNET_OK) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipEvent_t prev = c->ev0;
        for (size_t i = 0; i < c->prof_pos; i++) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, prev, c->prof_ev[i]);
            acc[i] += ms;
            prev = c->prof_ev[i];
        }
    }
    for (size_t i = 0; i < ns; i++) stage_ms_avg[i] = (float)(acc[i] / iters);
    return HNET_OK;
}

/* ---- operator-level entry points (host buffers) ---- */
int hnet_op_warp(hnet_ctx* c, const float* img, const float* H, float* out) {
    if (!c || !img || !H || !out) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    DevTemps t;
    float *d_i = nullptr, *d_o = nullptr, *d_h = nullptr;
    HIPCHK(c, t.alloc(&d_i, (size_t)NPIX)); HIPCHK(c, t.alloc(&d_o, (size_t)NPIX)); HIPCHK(c, t.alloc(&d_h, (size_t)9));
    HIPCHK(c, hipMemcpy(d_i, img, NPIX * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_h, H, 36, hipMemcpyHostToDevice));
    HIPCHK(c, launch_warp_f32(d_i, d_h, d_o, 1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, d_o, NPIX * 4, hipMemcpyDeviceToHost));
    return HNET_OK;
}

int hnet_op_dlt(hnet_ctx* c, const float* dst, int n, float* H) {
    if (!c || !dst || !H || n < 1) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    DevTemps t;
    float *d_d = nullptr, *d_h = nullptr;
    HIPCHK(c, t.alloc(&d_d, (size_t)n * 8)); HIPCHK(c, t.alloc(&d_h, (size_t)n * 9));
    HIPCHK(c, hipMemcpy(d_d, dst, (size_t)n * 32, hipMemcpyHostToDevice));
    HIPCHK(c, launch_dlt(d_d, d_h, n, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(H, d_h, (size_t)n * 36, hipMemcpyDeviceToHost));
    return HNET_OK;
}

int hnet_op_conv(hnet_ctx* c, int layer, const float* in, int batch, int h, int w, float* out) {
    if (!c || !in || !out || layer < 0 || layer >= 20 || batch < 1 || h < 1 || w < 1) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const ConvDesc& d = kConvs[layer];
    const int ho = conv_out_dim(h, d.ks, d.stride), wo = conv_out_dim(w, d.ks, d.stride);
    const size_t n_in = (size_t)batch * d.cin * h * w, n_out = (size_t)batch * d.cout * ho * wo;
    DevTemps t;
    float *d_a = nullptr, *d_b = nullptr, *d_c = nullptr, *d_d = nullptr;
    HIPCHK(c, t.alloc(&d_a, n_in)); HIPCHK(c, t.alloc(&d_b, n_in)); HIPCHK(c, t.alloc(&d_c, n_out)); HIPCHK(c, t.alloc(&d_d, n_out));
    HIPCHK(c, hipMemcpy(d_a, in, n_in * 4, hipMemcpyHostToDevice));
    if (!c->s3) {
        HIPCHK(c, launch_nchw_to_nhwc(d_a, d_b, batch, d.cin, h, w, c->stream));
        HIPCHK(c, launch_conv(layer, d_b, batch, h, w, c->conv_w[layer], c->conv_b[layer], d_c, c->stream));
        HIPCHK(c, launch_nhwc_to_nchw(d_c, d_d, batch, d.cout, ho, wo, c->stream));
    } else {   // split-bf16 mode: the layer reads / writes three bf16 planes, exactly as inside the forward
        uint16_t *p_in = nullptr, *p_out = nullptr;
        int out_np = c->n_planes;            // plane format p_out is written in
        HIPCHK(c, t.alloc(&p_in, 3 * n_in + 32));
        HIPCHK(c, t.alloc(&p_out, 3 * n_out + 32));
        if (c->use_patch && (conv_is_patch_layer(layer) || (c->use_patch32 && conv_is_patch32_layer(layer) && h == 56 && w == 80))) {
            HIPCHK(c, launch_nchw_f32_to_nhwc_s3(d_a, p_in, n_in, batch, d.cin, h, w, c->stream, c->n_planes));
            HIPCHK(c, launch_conv_patch(layer, p_in, n_in, batch, h, w, c->patch_frag[layer], c->conv_b[layer], p_out, n_out, c->stream, c->n_planes, c->patch_b128, c->patch_rb5));
        } else if (conv_is_s3_layer(layer)) {
            HIPCHK(c, launch_nchw_f32_to_nhwc_s3(d_a, p_in, n_in, batch, d.cin, h, w, c->stream, c->n_planes));
            HIPCHK(c, launch_conv_s3(layer, p_in, n_in, batch, h, w, c->conv_w16[layer], (size_t)d.cout * conv_padded_k(layer),
                                     c->conv_b[layer], p_out, n_out, nullptr, c->stream, nullptr, 0, c->conv_wfrag[layer], c->n_planes, c->s3_tile));
        } else {
            HIPCHK(c, launch_nchw_to_nhwc(d_a, d_b, batch, d.cin, h, w, c->stream));
            if (conv_is_first_s2(layer) && c->first_s2 && c->s2_frag[layer] && h == (layer == 0 ? 28 : 56) && w == (layer == 0 ? 40 : 80))
                HIPCHK(c, launch_conv_first_s2(layer, d_b, c->s2_frag[layer], c->conv_b[layer], p_out, n_out, batch, c->stream, c->n_planes));
            else if (layer == 7 && c->b30_s3 && c->b30_frag)     // the kernel the forward uses for block_3_0
                HIPCHK(c, launch_conv_first_s3(d_b, c->b30_frag, c->conv_b[layer], p_out, n_out, batch, h, w, c->stream, c->n_planes));
            else {   // Cin = 2 layers at other geometries: the fp32-MFMA kernel, which writes three bf16 planes
                HIPCHK(c, launch_conv(layer, d_b, batch, h, w, c->conv_w[layer], c->conv_b[layer], nullptr, c->stream, nullptr, 0, p_out, n_out));
                if (out_np == 2) out_np = 3;
            }
        }
        HIPCHK(c, launch_nhwc_s3_to_nchw_f32(p_out, n_out, d_d, batch, d.cout, ho, wo, c->stream, out_np));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, d_d, n_out * 4, hipMemcpyDeviceToHost));
    return HNET_OK;
}

int hnet_op_block4_fused(hnet_ctx* c, const float* in, int batch, int reverse, float* out) {
    if (!c || !in || !out || batch < 1) return HNET_ERR_INVALID_ARG;
    if (!c->fuse_b4 || !c->b40_frag || !c->b41_frag) return fail(c, HNET_ERR_UNSUPPORTED, "the fused block_4_0 + block_4_1 kernel exists in the split-bf16 mode only");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const size_t n_in = (size_t)batch * 2 * NPIX, n_out = (size_t)batch * 16 * (IMG_H / 2) * (IMG_W / 2);
    DevTemps t;
    float *d_a = nullptr, *d_b = nullptr, *d_d = nullptr;
    uint16_t* p_out = nullptr;
    HIPCHK(c, t.alloc(&d_a, n_in)); HIPCHK(c, t.alloc(&d_b, n_in)); HIPCHK(c, t.alloc(&d_d, n_out)); HIPCHK(c, t.alloc(&p_out, 3 * n_out + 32));
    HIPCHK(c, hipMemcpy(d_a, in, n_in * 4, hipMemcpyHostToDevice));
    HIPCHK(c, launch_nchw_to_nhwc(d_a, d_b, batch, 2, IMG_H, IMG_W, c->stream));
    const void* x_in = d_b;
    size_t x_plane = 0;
    {                                    // the kernel reads padded 16-bit planes with a zero border
        uint32_t* d_p = nullptr;
        x_plane = (size_t)batch * B4_HP * B4_WP;
        HIPCHK(c, t.alloc(&d_p, 3 * x_plane));
        HIPCHK(c, hipMemsetAsync(d_p, 0, 3 * x_plane * 4, c->stream));
        HIPCHK(c, launch_f32_nhwc_to_s3pad(d_b, d_p, x_plane, batch, c->n_planes, c->stream));
        x_in = d_p;
    }
    HIPCHK(c, launch_block4_fused(x_in, x_plane, c->b40_frag, c->conv_b[13], c->b41_frag, c->conv_b[14], p_out, n_out, batch, c->stream, (reverse ? 1 : 0) | (c->b4_flags & 32),
                                  c->n_planes));
    HIPCHK(c, launch_nhwc_s3_to_nchw_f32(p_out, n_out, d_d, batch, 16, IMG_H / 2, IMG_W / 2, c->stream, c->n_planes));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, d_d, n_out * 4, hipMemcpyDeviceToHost));
    return HNET_OK;
}

int hnet_op_block42_fused(hnet_ctx* c, const float* in, int batch, float* out) {
    if (!c || !in || !out || batch < 1) return HNET_ERR_INVALID_ARG;
    if (!c->fuse_b42 || !c->b42_w2 || !c->b42_w3) return fail(c, HNET_ERR_UNSUPPORTED, "the fused block_4_2 + block_4_3 kernel exists in the fp16-plane mode only");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const int h1 = IMG_H / 2, w1 = IMG_W / 2;
    const size_t n_in = (size_t)batch * 16 * h1 * w1, n_out = (size_t)batch * 64 * (h1 / 4) * (w1 / 4);
    DevTemps t;
    float *d_a = nullptr, *d_d = nullptr;
    uint16_t *p_in = nullptr, *p_out = nullptr;
    HIPCHK(c, t.alloc(&d_a, n_in)); HIPCHK(c, t.alloc(&d_d, n_out)); HIPCHK(c, t.alloc(&p_in, 3 * n_in + 32)); HIPCHK(c, t.alloc(&p_out, 3 * n_out + 32));
    HIPCHK(c, hipMemcpy(d_a, in, n_in * 4, hipMemcpyHostToDevice));
    HIPCHK(c, launch_nchw_f32_to_nhwc_s3(d_a, p_in, n_in, batch, 16, h1, w1, c->stream, c->n_planes));
    uint16_t* p_pad = nullptr;                     // the kernel's input layout: planes with a zero border (kernels.h B42_*)
    const size_t n_pad = (size_t)batch * B42_IMG * 16;
    HIPCHK(c, t.alloc(&p_pad, 2 * n_pad));
    HIPCHK(c, hipMemsetAsync(p_pad, 0, 2 * n_pad * 2, c->stream));
    HIPCHK(c, launch_s3_repitch(p_in, n_in, p_pad, n_pad, batch, h1, w1, 16, B42_HP, B42_WP, B42_PADY, B42_PADX, true, c->stream, 2));
    HIPCHK(c, launch_block42_fused(p_pad, n_pad, c->b42_w2, c->conv_b[15], c->b42_w3, c->conv_b[16], p_out, n_out, batch, c->stream, c->n_planes));
    HIPCHK(c, launch_nhwc_s3_to_nchw_f32(p_out, n_out, d_d, batch, 64, h1 / 4, w1 / 4, c->stream, c->n_planes));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, d_d, n_out * 4, hipMemcpyDeviceToHost));
    return HNET_OK;
}

int hnet_op_block3_fused(hnet_ctx* c, const float* in, int batch, float* out) {
    if (!c || !in || !out || batch < 1) return HNET_ERR_INVALID_ARG;
    if (!c->fuse_b3 || !c->b30_frag || !c->b3f_w1) return fail(c, HNET_ERR_UNSUPPORTED, "the fused block_3_0 + block_3_1 kernel exists in the fp16-plane mode only");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const int h0 = IMG_H / 2, w0 = IMG_W / 2;
    const size_t n_in = (size_t)batch * 2 * h0 * w0, n_out = (size_t)batch * 32 * (h0 / 2) * (w0 / 2);
    DevTemps t;
    float *d_a = nullptr, *d_b = nullptr, *d_d = nullptr;
    uint16_t* p_out = nullptr;
    HIPCHK(c, t.alloc(&d_a, n_in)); HIPCHK(c, t.alloc(&d_b, n_in)); HIPCHK(c, t.alloc(&d_d, n_out)); HIPCHK(c, t.alloc(&p_out, 3 * n_out + 32));
    HIPCHK(c, hipMemcpy(d_a, in, n_in * 4, hipMemcpyHostToDevice));
    HIPCHK(c, launch_nchw_to_nhwc(d_a, d_b, batch, 2, h0, w0, c->stream));
    HIPCHK(c, launch_block3_fused(d_b, c->b30_frag, c->conv_b[7], c->b3f_w1, c->conv_b[8], p_out, n_out, batch, c->stream, c->n_planes));
    HIPCHK(c, launch_nhwc_s3_to_nchw_f32(p_out, n_out, d_d, batch, 32, h0 / 2, w0 / 2, c->stream, c->n_planes));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, d_d, n_out * 4, hipMemcpyDeviceToHost));
    return HNET_OK;
}

static int op_prep_impl(hnet_ctx* c, const void* img1, const void* img2, int pix_fmt, const float* H, int k, float* out) {
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const int ho = IMG_H / k, wo = IMG_W / k;
    const size_t px = pix_fmt == HNET_PIX_U8 ? 1 : 4;
    DevTemps t;
    uint8_t *d_1 = nullptr, *d_2 = nullptr;
    float *d_h = nullptr, *d_o = nullptr, *d_t = nullptr;
    HIPCHK(c, t.alloc(&d_1, NPIX * px)); HIPCHK(c, t.alloc(&d_2, NPIX * px)); HIPCHK(c, t.alloc(&d_h, (size_t)9));
    HIPCHK(c, t.alloc(&d_o, (size_t)2 * ho * wo)); HIPCHK(c, t.alloc(&d_t, (size_t)2 * ho * wo));
    HIPCHK(c, hipMemcpy(d_1, img1, NPIX * px, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_2, img2, NPIX * px, hipMemcpyHostToDevice));
    if (H) HIPCHK(c, hipMemcpy(d_h, H, 36, hipMemcpyHostToDevice));
    HIPCHK(c, launch_prep(d_1, d_2, pix_fmt, H ? d_h : nullptr, k, d_o, 1, c->stream, nullptr, 0, 3, c->warp_exact));
    HIPCHK(c, launch_nhwc_to_nchw(d_o, d_t, 1, 2, ho, wo, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, d_t, (size_t)2 * ho * wo * 4, hipMemcpyDeviceToHost));
    return HNET_OK;
}

int hnet_op_prep(hnet_ctx* c, const float* img1, const float* img2, const float* H, int k, float* out) {
    if (!c || !img1 || !img2 || !out || (k != 1 && k != 2 && k != 4 && k != 8)) return HNET_ERR_INVALID_ARG;
    return op_prep_impl(c, img1, img2, HNET_PIX_F32, H, k, out);
}

int hnet_op_prep_u8(hnet_ctx* c, const uint8_t* img1, const uint8_t* img2, const float* H, int k, float* out) {
    if (!c || !img1 || !img2 || !out || (k != 1 && k != 2 && k != 4 && k != 8)) return HNET_ERR_INVALID_ARG;
    return op_prep_impl(c, img1, img2, HNET_PIX_U8, H, k, out);
}

int hnet_debug_layer_output(hnet_ctx* c, int layer, int pair, float* out, size_t cap) {
    if (!c || !out || layer < 0 || layer >= 20 || pair < 0 || pair >= c->cfg.max_batch) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const size_t n = (size_t)c->act_c[layer] * c->act_h[layer] * c->act_w[layer];
    if (cap < n) return fail(c, HNET_ERR_INVALID_ARG, "buffer too small");
    DevTemps t;
    float* d_t = nullptr;
    HIPCHK(c, t.alloc(&d_t, n));
    if (c->fuse_b4 && layer == 13) {   // the fused kernel keeps block_4_0's output in LDS: recompute it unfused for inspection
        if (c->b4_in_stale) return fail(c, HNET_ERR_UNSUPPORTED, "layer 13 of a batch > 8: block 4 sampled its input in-kernel (inspect it on a context without HNET_VARIANT_WARP_FUSE)");
        uint16_t* tmp = nullptr;
        HIPCHK(c, t.alloc(&tmp, 3 * n));
        const float* xin = c->x_in[3] + (size_t)pair * NPIX * 2;
        if (c->x16_b4) {                 // the block-4 input only exists as padded bf16 planes: their sum is the exact fp32 value
            float* xf = nullptr;
            HIPCHK(c, t.alloc(&xf, (size_t)NPIX * 2));
            HIPCHK(c, launch_s3pad_to_f32_nhwc(c->x16_b4 + (size_t)pair * B4_HP * B4_WP, c->x16_plane, xf, 1, c->n_planes, c->stream));
            xin = xf;
        }
        HIPCHK(c, launch_conv(13, xin, 1, IMG_H, IMG_W, c->conv_w[13], c->conv_b[13], nullptr, c->stream,
                              nullptr, 0, tmp, n));
        HIPCHK(c, launch_nhwc_s3_to_nchw_f32(tmp, n, d_t, 1, c->act_c[13], c->act_h[13], c->act_w[13], c->stream));   // three planes: written by the fp32-MFMA kernel
    } else if (c->fuse_b42 && layer == 15) {   // block_4_2's output lives in LDS only: recompute it with the stand-alone patch kernel for inspection
        uint16_t* tmp = nullptr;
        HIPCHK(c, t.alloc(&tmp, 3 * n));
        const size_t n14 = c->act_count[14];
        const uint16_t* a14 = c->act16[14] + (size_t)pair * n14;
        size_t a14_plane = (size_t)c->cfg.max_batch * n14;
        if (c->a14_pad) {                          // bordered layout -> a plain copy of this pair
            uint16_t* plain = nullptr;
            HIPCHK(c, t.alloc(&plain, 2 * n14));
            HIPCHK(c, launch_s3_repitch(c->act16[14] + (size_t)pair * B42_IMG * 16, (size_t)c->cfg.max_batch * B42_IMG * 16, plain, n14, 1, c->act_h[14], c->act_w[14], 16,
                                        B42_HP, B42_WP, B42_PADY, B42_PADX, false, c->stream, 2));
            a14 = plain; a14_plane = n14;
        }
        HIPCHK(c, launch_conv_patch(15, a14, a14_plane, 1, c->act_h[14], c->act_w[14], c->patch_frag[15],
                                    c->conv_b[15], tmp, n, c->stream, c->n_planes, c->patch_b128, c->patch_rb5));
        HIPCHK(c, launch_nhwc_s3_to_nchw_f32(tmp, n, d_t, 1, c->act_c[15], c->act_h[15], c->act_w[15], c->stream, c->n_planes));
    } else if (c->fuse_b3 && layer == 7) {   // the fused block-3 kernel keeps block_3_0's output in LDS: recompute it unfused for inspection
        uint16_t* tmp = nullptr;
        HIPCHK(c, t.alloc(&tmp, 3 * n));
        HIPCHK(c, launch_conv_first_s3(c->x_in[2] + (size_t)pair * (NPIX / 4) * 2, c->b30_frag, c->conv_b[7], tmp, n, 1, IMG_H / 2, IMG_W / 2, c->stream, c->n_planes));
        HIPCHK(c, launch_nhwc_s3_to_nchw_f32(tmp, n, d_t, 1, c->act_c[7], c->act_h[7], c->act_w[7], c->stream, c->n_planes));
    } else if (layer == 14 && c->a14_pad) {        // bordered layout -> a plain copy of this pair
        uint16_t* plain = nullptr;
        HIPCHK(c, t.alloc(&plain, 2 * n));
        HIPCHK(c, launch_s3_repitch(c->act16[14] + (size_t)pair * B42_IMG * 16, (size_t)c->cfg.max_batch * B42_IMG * 16, plain, n, 1, c->act_h[14], c->act_w[14], 16,
                                    B42_HP, B42_WP, B42_PADY, B42_PADX, false, c->stream, 2));
        HIPCHK(c, launch_nhwc_s3_to_nchw_f32(plain, n, d_t, 1, c->act_c[layer], c->act_h[layer], c->act_w[layer], c->stream, 2));
    } else if (c->act16[layer])
        HIPCHK(c, launch_nhwc_s3_to_nchw_f32(c->act16[layer] + (size_t)pair * n, (size_t)c->cfg.max_batch * n, d_t, 1, c->act_c[layer],
                                             c->act_h[layer], c->act_w[layer], c->stream,
                                             (conv_is_first_s2(layer) && !c->first_s2) ? 3 : c->n_planes));   // HNET_FIRST_S2=0: block_1_1 / block_2_1 from the fp32-MFMA kernel, three planes
    else
        HIPCHK(c, launch_nhwc_to_nchw(c->act[layer] + (size_t)pair * n, d_t, 1, c->act_c[layer], c->act_h[layer], c->act_w[layer], c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, d_t, n * 4, hipMemcpyDeviceToHost));
    return HNET_OK;
}

int hnet_debug_h_part1(hnet_ctx* c, int pair, float* out9) {
    if (!c || !out9 || pair < 0 || pair >= c->cfg.max_batch) return HNET_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out9, (c->H_last ? c->H_last : c->Hm) + (size_t)pair * 9, 36, hipMemcpyDeviceToHost));
    return HNET_OK;
}

}  // extern "C"
