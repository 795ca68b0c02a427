// chain_lat.h — the TAIL of a block (its last two or three stride-2 convolutions) for SMALL batches (latency path, round 6; fp16-plane mode) as ONE
// launch whose working workgroups all sit on ONE XCD, the layers separated by XCD-local counter barriers instead of kernel boundaries.
// Reference: the conv + bias + LeakyReLU(0.1) stacks of model_to_trace.py:88-115 (blocks 1 - 3) and :210-216 (block 4), layers
//   block 1: block_1_2 (128 -> 128, 5 x 5), block_1_3 (128 -> 256, 3 x 3)                      14 x 20 -> 7 x 10 -> 4 x 5
//   block 2: block_2_2 (64 -> 128, 5 x 5), block_2_3 (128 -> 256, 3 x 3), block_2_4 (256 -> 256) 28 x 40 -> 14 x 20 -> 7 x 10 -> 4 x 5
//   blocks 3, 4: block_x_3 / _4 (64 -> 128, 3 x 3), _4 / _5 (128 -> 256), _5 / _6 (256 -> 256)   the same sizes
//
// Why.  At batch 1 these layers are 0.01 - 0.06 GFLOP each behind 0.3 - 2.4 MB of weights.  As launches they are 18 kernels of 4 - 9 us (split-K GEMM + reduce /
// last-arriver tickets; rocprofv3: profiles/r06_experiments_not_shipped.log item 10), none of which does more than a microsecond of work: a dependent kernel costs
// its boundary and three or four dependent round trips whatever it does.  A hand-off INSIDE a launch is cheap only between workgroups that share an L2
// (MI355X_MICROARCH.md, price list: chip-wide barriers 4 - 7 us, same-XCD hand-offs ~ 1 us).  So one XCD (32 CUs, one L2) runs the whole tail of a frame pair:
//   * plain stores of a layer's output stay in that XCD's L2; the next layer reads them with sc1 loads (L1 bypassed, L2 served): no write-through, no fence;
//   * a layer is cut into 32 ITEMS (a tile of 8 / 16 output channels x a slice of output rows, the whole K): one per CU, no split-K partials in memory - the
//     eight waves of a workgroup take equal shares of K and add their partial tiles through LDS in wave order (deterministic);
//   * the weights come as pre-packed MFMA fragments straight into registers (hnet_create packs them in consumption order: a wave instruction is contiguous), every
//     load of an item in flight at once - and ONE LAYER AHEAD: a workgroup draws its first item of every layer when the launch starts, and issues the fragments
//     of its next layer's item behind the stores of the current one (its registers are free there), so that they travel during the signalling, the wait for the
//     layer and the region copy (counted vmcnt waits: loads and stores complete in issue order);
//   * the item's input region is staged once into LDS by LDS-DMA as four parity images (a stride-2 tap reads ONE of them: consecutive output pixels are
//     consecutive slots; slot pitch = CIN x 2 + 16 bytes: the 16 pixels of a B fragment fall into 16 different bank groups), zero padding included; the slot ->
//     pixel table is built before the wait;
//   * workgroup barriers wait for LDS traffic only (ch_bar): __syncthreads() would wait for the fragments in flight and for every counter atomic.
// What a layer costs here (tools/trace_chain.hip, in-kernel stamps): wait for the previous layer 0.7 us, region copy 1.9 (~ 100 KB per CU at ~ 50 GB/s), MFMAs 1.3
// (two waves per SIMD share the matrix pipe), reduction + stores + next fragments 1.5, signal 0.4, ~ 0.8 of transitions: 6.5 us per layer, 22.5 us for a
// three-layer chain against 28 us + gaps for the five kernels it replaces.
//
// Placement independence.  HIP promises nothing about workgroup -> XCD placement, so nothing here assumes it: every workgroup reads its XCD from HW_REG_XCC_ID;
// the XCD whose workgroup wins an agent-scope compare-and-swap on the pair's claim word OWNS that pair for this launch (pair p is first tried by the XCDs
// x = p mod batch; a pair nobody claimed - an XCD without workgroups - is picked up by any workgroup that is done), workgroups of other XCDs leave.  Items are
// CLAIMED from an XCD-local counter (any number >= 1 of resident owner workgroups finishes the chain: no co-residency assumption, no deadlock: a workgroup only
// ever waits for items that running workgroups have claimed), every spin is bounded (a timeout raises bit 1 of the context's flag word).  The tests launch the
// chain with 8 and with 3 workgroups: same bits.  With the observed round-robin placement a 256-workgroup launch gives every XCD 32 workgroups and up to eight
// pairs run on eight XCDs side by side.
//
// The claim words are the only words XCDs share (agent-scope compare-and-swap); the item counter and the done counters of a pair are touched by its owner XCD
// only (L2-local atomics, sc1 polls).  Every chain launch of a forward has its own counter AREA, all zero when the launch starts: workgroup 0 of the PREVIOUS
// chain launch of the stream zeroes it (write-through stores; stream order: nobody uses it then) - no word is ever reset while somebody may still use it, and
// consecutive launches may be owned by different XCDs (kernel boundaries write the L2 back).
//
// Arithmetic: the two-plane fp16 form of igemm_s3.h (hi += W0 A0, lo += W0 A1 + W1 A0, result = hi + lo / 4096) with K summed in another order than the
// split-K kernels: results agree to fp32 rounding (tests/test_gpu_latency_path.py gates 5e-5 px; hnet_config.variant HNET_VARIANT_NO_CHAIN keeps the launches).
#pragma once
#include "igemm_pipe.h"
#include "chain_args.h"

namespace hnet {

constexpr int CH_NT = 512, CH_NW = 8;

// phase time stamps for tools/trace_chain.hip (compiled out of the library): [workgroup][32] x 100-MHz ticks
#ifdef HNET_CHAIN_TRACE
__device__ unsigned long long* g_chain_trace;
#define CHT(slot)                                                                                               \
    do {                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if (threadIdx.x == 0 && (slot) < 32) g_chain_trace[blockIdx.x * 32 + (slot)] = wall_clock64();          \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } while (0)
#else
#define CHT(slot) do {} while (0)
#endif

// NCH: output channels per item (8: half of the MFMA's rows are idle - these layers are bound by operand delivery, not by the matrix pipe); MS: slices of output rows
template <int CIN_, int KS_, int HI_, int WI_, int COUT_, int NCH_, int MS_, bool OUT32_>
struct ChainCfg {
    static constexpr int CIN = CIN_, KS = KS_, HI = HI_, WI = WI_, COUT = COUT_, NCH = NCH_, MS = MS_, PAD = (KS_ - 1) / 2;
    static constexpr bool OUT32 = OUT32_;
    static constexpr int HO = (HI_ + 2 * PAD - KS_) / 2 + 1, WO = (WI_ + 2 * PAD - KS_) / 2 + 1;
    static constexpr int ROWS = (HO + MS_ - 1) / MS_;                   // output rows of a slice (the last one may hold fewer)
    static constexpr int MI = ROWS * WO, TM = (MI + 15) / 16;           // GEMM rows of an item, 16-row M-tiles
    static constexpr int RI = 2 * (ROWS - 1) + KS_;                     // input rows of a slice's region
    static constexpr int PR = (RI + 1) / 2, PC = WO + PAD;              // rows / columns of a parity image
    static constexpr int SLOTS = 4 * PR * PC, PITCH = CIN_ * 2 + 16, PLANE = SLOTS * PITCH, REGION = 2 * PLANE;
    static constexpr int CPS = CIN_ / 32, NSTEP = KS_ * KS_ * CPS;      // 32-deep MFMA steps: tap major, 32-channel chunk minor
    static constexpr int SPW = (NSTEP + CH_NW - 1) / CH_NW;             // most steps of a wave
    static constexpr int NTILES = COUT_ / NCH_, ITEMS = NTILES * MS_;
    static constexpr int PPS = CIN_ / 8;                                // 16-byte pieces of a slot
    static constexpr int NPIECE = 2 * SLOTS * PPS, STG = (NPIECE + CH_NT - 1) / CH_NT;
    static constexpr int FRAG_BYTES = NCH_ * 64;                        // one (step, plane): 4 lane groups x NCH rows x 16 B
    static constexpr size_t WFRAG_HALVES = (size_t)NTILES * NSTEP * 2 * NCH_ * 4 * 8;
    static_assert(CIN_ % 32 == 0 && COUT_ % NCH_ == 0 && (NCH_ == 8 || NCH_ == 16) && WI_ % 2 == 0 && TM <= 5 && (MS_ - 1) * ROWS < HO, "chain layer");
    static_assert(REGION <= 153 * 1024 && CH_NW * TM * 1024 <= REGION && SLOTS * 4 <= 3 * 1024, "LDS");
};

// dynamic LDS: the region (the largest, block_1_2's: 156 672 B; its first bytes double as the reduction buffer), the slot table (<= 528 words), the broadcast words
constexpr int CH_TABLE_OFF = 153 * 1024, CH_LDS_BYTES = 160 * 1024;
// last layers (OUT32): the item's slice of the block-tail FC's weights ([8 outputs][HO WO pixels][NCH channels] floats, 5 KB) behind the region, then the two waves' partial sums
constexpr int CH_FC_OFF = 128 * 1024, CH_FC_RED_OFF = CH_FC_OFF + 8 * 1024;

__device__ __forceinline__ uint32_t ch_load_sc1(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // global_load_dword sc1: L1 bypassed
__device__ __forceinline__ uint32_t ch_local_add(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // executed in this XCD's L2
template <int N> __device__ __forceinline__ void ch_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier that waits for this wave's LDS traffic ONLY.  __syncthreads() is a workgroup-scope fence + barrier: hipcc puts s_waitcnt vmcnt(0) in front of it,
// i.e. every barrier of a layer would wait for the weight fragments in flight (issued exactly so that they travel DURING the wait for the previous layer and the
// region copy) and for the acknowledgement of every counter atomic (~1 us each).  Global data is ordered here by explicit vmcnt waits where it matters.
__device__ __forceinline__ void ch_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// sum over the 16 lanes of a DPP row, in every lane of the row (four cross-lane adds in the vector pipe: quad swaps, then the row's half mirror and mirror - a fixed pairing)
template <int CTRL> __device__ __forceinline__ float ch_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float ch_row_sum(float v) {
    v += ch_dpp<0xB1>(v);      // quad_perm [1, 0, 3, 2]
    v += ch_dpp<0x4E>(v);      // quad_perm [2, 3, 0, 1]
    v += ch_dpp<0x141>(v);     // row_half_mirror
    v += ch_dpp<0x140>(v);     // row_mirror
    return v;
}
__device__ __forceinline__ float ch_lane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

// waits until *p == target (thread 0 polls, the workgroup follows through the barrier); false = gave up
__device__ __forceinline__ bool ch_wait_eq(const uint32_t* p, uint32_t target, uint32_t* lds_word) {
    if (threadIdx.x == 0) {
        uint32_t ok = 0;
        for (uint32_t n = 0; n < CH_SPIN_LIMIT; n++) {
            if (ch_load_sc1(p) == target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        *lds_word = ok;
    }
    ch_bar();
    const bool r = *lds_word != 0;
    ch_bar();
    return r;
}

// the pieces of one item of layer C
template <class C>
struct ChainOps {
    static constexpr int TM = C::TM, SPW = C::SPW, NCH = C::NCH;
    // every wave takes SPW consecutive steps; steps beyond NSTEP (K padded to eight equal shares) load zeros (out-of-range offsets: no traffic) against any valid
    // fragment of the region - no control flow in the unrolled loops, one constant in the counted waits
    typedef bf16x8 W[SPW][2];
    struct Item { int tile, oy0, mi; };
    __device__ static __forceinline__ Item item(int it) {
        Item r;
        r.tile = it / C::MS;
        const int ms = it - r.tile * C::MS;
        r.oy0 = ms * C::ROWS;
        r.mi = min(C::ROWS, C::HO - r.oy0) * C::WO;
        return r;
    }
    // (1) this wave's weight fragments: steps [wave SPW, (wave + 1) SPW) of the item's K, every load in flight at once
    __device__ static __forceinline__ void load_w(const ChainLayer& L, const Item& it, W& fw, int wave, int lane) {
        const int li = lane & 15, lg = lane >> 4;
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)L.wfrag, 0, 0x7FFFFFF0, 0x00020000);
        const uint32_t wlane = li < NCH ? (uint32_t)((lg * NCH + li) * 16) : S3_OOB;      // rows beyond the item's channels: zeros, no traffic
        const int base = (it.tile * C::NSTEP + wave * SPW) * 2 * C::FRAG_BYTES;
#pragma unroll
        for (int k = 0; k < SPW; k++) {
            const uint32_t vo = wave * SPW + k < C::NSTEP ? wlane : S3_OOB;
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
                fw[k][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, vo, base + (k * 2 + pl) * C::FRAG_BYTES, 0));
        }
    }
    // waits until at most this wave's load_w of layer C is in flight (loads and stores return in issue order: everything issued BEFORE it has completed)
    __device__ static __forceinline__ void wait_all_but_w(int) { ch_wait_vm<2 * SPW>(); }
    // (2a) slot -> byte offset of its pixel in a plane of the input (S3_OOB: padding): a table in LDS, once per item (it depends on the item only: built before the
    // wait for the previous layer) - the region copy below then costs three instructions per piece instead of thirty
    __device__ static __forceinline__ void slot_table(int pair, const Item& it, uint32_t* table, int tid) {
        const int iy_base = 2 * it.oy0 - C::PAD;
        for (int slot = tid; slot < C::SLOTS; slot += CH_NT) {
            const int img = slot / (C::PR * C::PC), q = slot - img * (C::PR * C::PC), rr = q / C::PC, cc = q - rr * C::PC;
            const int ry = 2 * rr + (img >> 1), rx = 2 * cc + (img & 1);
            const int iy = iy_base + ry, ix = rx - C::PAD;
            const bool ok = ry < C::RI && (unsigned)iy < (unsigned)C::HI && (unsigned)ix < (unsigned)C::WI;
            table[slot] = ok ? (uint32_t)((((pair * C::HI + iy) * C::WI + ix) * C::CIN) * 2) : S3_OOB;
        }
    }
    // (2b) region: input rows 2 oy0 - PAD ... as four parity images, zero padding and the slots' pad pieces included, by LDS-DMA (no registers): a wave
    // instruction fills 64 consecutive 16-byte pieces; sc1: never this CU's L1 (another CU of the XCD wrote the bytes), served by the L2
    __device__ static __forceinline__ void stage(const ChainLayer& L, const uint32_t* table, uint8_t* smem, int wave, int lane) {
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)L.in, 0, 0x7FFFFFF0, 0x00020000);
        constexpr int PPP = C::PPS + 1, PPL = C::SLOTS * PPP;                // pieces of a slot (the last one is the pad), of a plane
        constexpr int NCHUNK = (2 * PPL + 63) / 64, NK = (NCHUNK + CH_NW - 1) / CH_NW;
        constexpr int DQ = (CH_NW * 64) / PPP, DR = (CH_NW * 64) % PPP;      // a step of eight chunks = DQ slots + DR pieces
        const uint32_t pl_bytes = (uint32_t)(L.in_plane * 2);
        int g = wave * 64 + lane;
        int slot = g / PPP, piece = g - slot * PPP;                          // (slot counts through both planes: plane 1 starts at SLOTS)
#ifdef HNET_CHAIN_UNROLL_STAGE
#pragma unroll
#else
#pragma unroll 4      // (four table reads in flight per pass; fully unrolled the copy was 2 000 instructions per layer for the same time)
#endif
        for (int k = 0; k < NK; k++) {
            const int chunk = wave + CH_NW * k;
            if (chunk < NCHUNK) {
                const int pl = slot >= C::SLOTS ? 1 : 0, sl = slot - pl * C::SLOTS;
                const uint32_t base = sl < C::SLOTS ? table[sl] : S3_OOB;
                const uint32_t vo = (piece < C::PPS && base != S3_OOB) ? base + pl * pl_bytes + (uint32_t)(piece * 16) : S3_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(smem + chunk * 1024), 16, vo, 0, 0, 16 /* sc1 */);
            }
            slot += DQ; piece += DR;
            if (piece >= PPP) { piece -= PPP; slot += 1; }
        }
    }
    // (2c) last layer of blocks 1 - 3: the item's slice of the FC weights, [o][pixel][channel group of 4] x 16 bytes, by LDS-DMA next to the region (no registers)
    __device__ static __forceinline__ void stage_fc(const float* fcw, const Item& it, uint8_t* smem, int wave, int lane) {
        constexpr int NPX = C::HO * C::WO, G = NCH / 4, PIECES = 8 * NPX * G, NI = (PIECES + 63) / 64;
        static_assert(C::MS == 1 && PIECES * 16 <= 8 * 1024 && C::REGION <= CH_FC_OFF && C::ITEMS == CH_FC_ITEMS, "FC slice of a last layer");
        const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void*)fcw, 0, 8 * 5120 * 4, 0x00020000);
        if (wave < NI) {
            const int q = wave * 64 + lane, o = q / (NPX * G), r = q - o * (NPX * G), px = r / G, h = r - px * G;
            const uint32_t vo = q < PIECES ? (uint32_t)(((o * 5120 + px * C::COUT + it.tile * NCH + h * 4)) * 4) : S3_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rF, (lds_ptr_t)(smem + CH_FC_OFF + wave * 1024), 16, vo, 0, 0, 0);
        }
    }
    // (3) this wave's K steps over every M-tile of the item, the eight K slices added through LDS in wave order, bias + LeakyReLU, store.
    // Entry: the region has landed and the workgroup has met at a barrier.  Exit: the item's stores have been ISSUED.
    __device__ static __forceinline__ void compute_store(const ChainLayer& L, int pair, const Item& it, W& fw, uint8_t* smem, int tid, int wave, int lane, int tb, const float* fcw = nullptr, float* fc_part = nullptr) {
        const int li = lane & 15, lg = lane >> 4;
        int lb[TM];
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int m = min(i * 16 + li, it.mi - 1);
            const int oyl = m / C::WO, ox = m - oyl * C::WO;
            lb[i] = (oyl * C::PC + ox) * C::PITCH + lg * 16;
        }
        f32x4_m16 hi[TM], lo[TM];
#pragma unroll
        for (int i = 0; i < TM; i++) { hi[i] = f32x4_m16{0.f, 0.f, 0.f, 0.f}; lo[i] = f32x4_m16{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int k = 0; k < SPW; k++) {
            const int s = min(wave * SPW + k, C::NSTEP - 1);                    // (a padding step multiplies zero weights with a valid fragment)
            const int t = s / C::CPS, c32 = s - t * C::CPS, kh = t / C::KS, kw = t - kh * C::KS;      // (wave-uniform)
            const int so = ((((kh & 1) * 2 + (kw & 1)) * C::PR + (kh >> 1)) * C::PC + (kw >> 1)) * C::PITCH + c32 * 64;
#pragma unroll
            for (int i = 0; i < TM; i++) {
                bf16x8 a[3], w[3];
                a[0] = *reinterpret_cast<const bf16x8*>(smem + lb[i] + so);
                a[1] = *reinterpret_cast<const bf16x8*>(smem + C::PLANE + lb[i] + so);
                a[2] = a[0];
                w[0] = fw[k][0]; w[1] = fw[k][1]; w[2] = fw[k][0];
                s3_mfma16_2acc(hi[i], lo[i], w, a);
            }
            __builtin_amdgcn_sched_barrier(0);                                  // (a step's fragment reads stay with its MFMAs: hoisting every read of the item spills)
        }
        ch_bar();                                                        // the region is dead: its first bytes become the reduction buffer
        CHT(tb + 3);
        float* red = reinterpret_cast<float*>(smem);                          // [8 waves][TM][64 lanes] x 16 B
#pragma unroll
        for (int i = 0; i < TM; i++) *reinterpret_cast<f32x4_m16*>(red + ((wave * TM + i) * 64 + lane) * 4) = hi[i] + lo[i] * S3_F16_INV;
        ch_bar();
        [[maybe_unused]] float fcp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (tid < TM * 64) {
            const int i = tid >> 6;                                            // M-tile (wave-uniform), lane = (pixel li, channel group lg)
            f32x4_m16 s = *reinterpret_cast<const f32x4_m16*>(red + ((0 * TM + i) * 64 + lane) * 4);
#pragma unroll
            for (int wv = 1; wv < CH_NW; wv++) s += *reinterpret_cast<const f32x4_m16*>(red + ((wv * TM + i) * 64 + lane) * 4);
            const int m = i * 16 + li;
            if (m < it.mi && 4 * lg < NCH) {
                const int n = it.tile * NCH + 4 * lg;
                const f32x4_m16 bv = *reinterpret_cast<const f32x4_m16*>(L.bias + n);
                f32x4_m16 x = s + bv;
#pragma unroll
                for (int e = 0; e < 4; e++) x[e] = s3p::lrelu(x[e]);
                const size_t pix = (size_t)pair * (C::HO * C::WO) + it.oy0 * C::WO + m;
                if constexpr (C::OUT32) {
                    *reinterpret_cast<f32x4_m16*>(L.out32 + pix * C::COUT + n) = x;
                    if (fcw) {                                                 // (kernel-uniform) this thread's four activations times their FC weights, eight outputs
                        const float* fw = reinterpret_cast<const float*>(smem + CH_FC_OFF);
#pragma unroll
                        for (int o = 0; o < 8; o++) {
                            const f32x4_m16 w4 = *reinterpret_cast<const f32x4_m16*>(fw + ((o * (C::HO * C::WO) + m) * (NCH / 4) + lg) * 4);
                            fcp[o] = fmaf(x[3], w4[3], fmaf(x[2], w4[2], fmaf(x[1], w4[1], x[0] * w4[0])));
                        }
                    }
                } else {
                    uint32_t pa[3], pb[3];
                    s3p::split_pair<2>(x[0], x[1], pa);
                    s3p::split_pair<2>(x[2], x[3], pb);
#pragma unroll
                    for (int pl = 0; pl < 2; pl++) *reinterpret_cast<uint2*>(L.out16 + pl * L.out_plane + pix * C::COUT + n) = make_uint2(pa[pl], pb[pl]);
                }
            }
        }
        if constexpr (C::OUT32) {
            if (fcw) {                                                         // (kernel-uniform) the item's partial sums: per wave (= M-tile) by DPP, then wave 0 + wave 1 + ...: a fixed order
                float* fred = reinterpret_cast<float*>(smem + CH_FC_RED_OFF);
                if (tid < TM * 64) {
                    // lanes (pixel li, channel group lg < NCH / 4) hold products, the others zeros: rows 0 .. NCH / 4 - 1 of the wave, each summed by DPP, then added in row order
                    static_assert(NCH / 4 <= 4, "one DPP row per channel group");
#pragma unroll
                    for (int o = 0; o < 8; o++) {
                        const float r = ch_row_sum(fcp[o]);
                        float v = ch_lane(r, 0);
#pragma unroll
                        for (int g2 = 1; g2 < NCH / 4; g2++) v += ch_lane(r, 16 * g2);
                        if (lane == 0) fred[wave * 8 + o] = v;
                    }
                }
                ch_bar();
                if (tid < 8) {
                    float v = fred[tid];
#pragma unroll
                    for (int wv = 1; wv < TM; wv++) v += fred[wv * 8 + tid];
                    fc_part[((size_t)pair * C::ITEMS + (size_t)(it.tile * C::MS)) * 8 + tid] = v;
                }
            }
        }
        // (the caller waits for the acknowledgement of these stores - counted, when it issues the next layer's weights behind them - before anybody is told)
    }
};
struct ChainNone {
    static constexpr int ITEMS = 0, SPW = 1;
};
template <> struct ChainOps<ChainNone> {
    typedef bf16x8 W[1][2];
    struct Item { int tile, oy0, mi; };
    __device__ static __forceinline__ Item item(int) { return Item{0, 0, 0}; }
    __device__ static __forceinline__ void load_w(const ChainLayer&, const Item&, W&, int, int) {}
    __device__ static __forceinline__ void wait_all_but_w(int) { ch_wait_vm<0>(); }
};

// One item of layer C up to its stores (issued, not yet acknowledged).  fw: its weight fragments, in flight.  wait_prev: the previous layer's counter to wait for
// (nullptr: none).  Returns the next item of this layer that the workgroup drew (>= ITEMS: none), or 0xFFFFFFFF when a bounded spin gave up.
template <class C>
__device__ __forceinline__ uint32_t chain_one_item(const ChainLayer& L, int pair, int phase, uint32_t item, typename ChainOps<C>::W& fw, const uint32_t* wait_prev,
                                                   uint32_t wait_target, uint32_t* slot, uint8_t* smem, uint32_t* table, uint32_t* lds_words, int tb,
                                                   const float* fcw, float* fc_part) {
    typedef ChainOps<C> O;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const typename O::Item it = O::item((int)item);
    O::slot_table(pair, it, table, tid);
    // the item's slice of the block-tail FC's weights (last layer of blocks 1 - 3): requested HERE, before the wait for the previous layer - they come from memory
    // (only the owner XCD of this launch reads them; its L2 need not hold them), the region comes from the L2: behind the region copy they delayed the MFMAs by ~ 1.5 us
    if constexpr (C::OUT32) { if (fcw) O::stage_fc(fcw, it, smem, wave, lane); }
    // the next unclaimed item of this layer, if any: drawn now, looked at after the item
    uint32_t drawn = 0;
    if (tid == 0) drawn = ch_local_add(slot, 1u << (10 * phase));
    CHT(tb + 0);
    // the previous layer of this pair is complete (its stores are in this XCD's L2); also the barrier behind the table
    if (wait_prev) { if (!ch_wait_eq(wait_prev, wait_target, lds_words)) return 0xFFFFFFFFu; }
    else ch_bar();
    CHT(tb + 1);
    O::stage(L, table, smem, wave, lane);
    ch_wait_vm<0>();
    ch_bar();
    CHT(tb + 2);
    if constexpr (C::OUT32) O::compute_store(L, pair, it, fw, smem, tid, wave, lane, tb, fcw, fc_part);
    else O::compute_store(L, pair, it, fw, smem, tid, wave, lane, tb);
    if (tid == 0) lds_words[1] = drawn;
    ch_bar();
    return (lds_words[1] >> (10 * phase)) & 1023u;
}

// One layer (phase) of the chain for this workgroup: every item of layer C it claims, `first` = the one it drew at the start of the launch (>= ITEMS: none).
// fw: the weight fragments of that first item: when PRE, the previous layer issued them behind its own stores - they travel during its signalling, the wait for it
// and this layer's region copy.  On exit the fragments of this workgroup's item of the NEXT layer (CN, item fn) are in flight in fwn, if it has one: issued at ONE
// place, behind the last item's stores, into registers that are free there (this layer's fragments and accumulators are dead) - one definition per register array,
// so no copies (a copy waits for the load), and two layers' fragments are never live together (a build that loaded them in front of the arithmetic spilled).
template <class C, class CPREV, class CN, bool PRE>
__device__ __forceinline__ bool chain_phase(const ChainLayer& L, const ChainLayer& Ln, int pair, int phase, uint32_t first, typename ChainOps<C>::W& fw, uint32_t fn,
                                            typename ChainOps<CN>::W& fwn, uint32_t* local, uint32_t* slot, uint8_t* smem, uint32_t* table, uint32_t* lds_words,
                                            const float* fcw = nullptr, float* fc_part = nullptr) {
    typedef ChainOps<C> O;
    typedef ChainOps<CN> ON;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tb = 3 + 6 * phase;
    const bool have = first < (uint32_t)C::ITEMS, have_n = CN::ITEMS != 0 && fn < (uint32_t)CN::ITEMS;
    const uint32_t* wait_prev = phase ? local + 8 + (phase - 1) : nullptr;
    if (have) {
        if constexpr (!PRE) O::load_w(L, O::item((int)first), fw, wave, lane);
        uint32_t item = chain_one_item<C>(L, pair, phase, first, fw, wait_prev, (uint32_t)CPREV::ITEMS, slot, smem, table, lds_words, tb, fcw, fc_part);
        if (item == 0xFFFFFFFFu) return false;
        while (item < (uint32_t)C::ITEMS) {                                        // more items than resident workgroups (never under the usual placement)
            ch_wait_vm<0>();                                                       // the previous item's stores are in the L2
            ch_bar();
            if (tid == 0) ch_local_add(local + 8 + phase, 1u);
            typename O::W fw2;
            O::load_w(L, O::item((int)item), fw2, wave, lane);
            item = chain_one_item<C>(L, pair, phase, item, fw2, nullptr, 0u, slot, smem, table, lds_words, 32, fcw, fc_part);
        }
    }
    // the next layer's fragments: behind the last item's stores (counted wait: the stores are acknowledged, the fragments fly on)
    if (have_n) ON::load_w(Ln, ON::item((int)fn), fwn, wave, lane);
    if (have) {
        if (have_n) ON::wait_all_but_w(wave);
        else ch_wait_vm<0>();
        ch_bar();
        CHT(tb + 4);
        if (tid == 0) ch_local_add(local + 8 + phase, 1u);                         // done: the item's stores are in the L2
        CHT(tb + 5);
    }
    return true;
}

// C0, C1, C2 (or ChainNone): the layers of the chain.  Grid: 256 workgroups of 512 threads (one per CU), CH_LDS_BYTES of dynamic LDS.
// a: the chain's layers (by value: in device memory every layer began with a dependent scalar round trip to memory, 1.5 us); sync: this launch's counter area (all zero at entry);
// next_sync: the area of the NEXT chain launch of the stream, zeroed here by workgroup 0 (stream order: nobody uses it now); batch: pairs of this call
template <class C0, class C1, class C2>
__global__ __launch_bounds__(CH_NT) void tail_chain_kernel(const ChainArgs a, uint32_t* __restrict__ sync, uint32_t* __restrict__ next_sync, int batch) {
    const ChainArgs* args = &a;
    extern __shared__ __attribute__((aligned(16))) uint8_t ch_smem[];
    uint32_t* lds_words = reinterpret_cast<uint32_t*>(ch_smem + CH_LDS_BYTES - 64);
    uint32_t* table = reinterpret_cast<uint32_t*>(ch_smem + CH_TABLE_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu;      // HW_REG_XCC_ID[3:0]: this workgroup's XCD
    CHT(0);
    const int B = batch;
    uint32_t* const claim = sync;
    const int pref = (int)(xcc % (uint32_t)B);

    // the claim of the preferred pair (agent scope: the one word every XCD must agree on) and, in the same round trip, this workgroup's first item of every
    // layer from the pair's counter of THIS XCD (an XCD that loses the claim drew from a counter nobody reads)
    if (tid == 0) {
        uint32_t* slot = sync + 32 + 32 * pref + xcc;
        const uint32_t d = ch_local_add(slot, 1u | (1u << 10) | (1u << 20));
        uint32_t expected = 0;
        __hip_atomic_compare_exchange_strong(claim + pref, &expected, xcc + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_words[2] = d;
        lds_words[8 + pref] = expected ? expected : xcc + 1;               // owner + 1 of the preferred pair
    }
    if (blockIdx.x == 0 && tid < CH_SYNC_WORDS / 4) {                      // the next launch's counters (write-through: its owner may be any XCD)
        const u32x4 z = {0u, 0u, 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(z, __builtin_amdgcn_make_buffer_rsrc((void*)next_sync, 0, CH_SYNC_WORDS * 4, 0x00020000), (uint32_t)(tid * 16), 0, 16 /* sc1 */);
    }
    ch_bar();
    CHT(1);
    bool ok = true;
    // pairs in the order pref, pref + 1, ...: the preferred one first; then (one more round trip for the whole scan) the pairs this XCD owns too - a workgroup of
    // it picked one up - and those NOBODY has claimed - an XCD without workgroups under some other placement - which any workgroup that is done picks up
    for (int k = 0; k < B && ok; k++) {
        const int pair = pref + k < B ? pref + k : pref + k - B;
        if (k == 1) {
            ch_bar();
            if (tid == 0) {
                uint32_t c[CH_MAX_PAIRS];
#pragma unroll
                for (int q = 0; q < CH_MAX_PAIRS; q++) c[q] = q < B && q != pref ? __hip_atomic_load(claim + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u << 30;
#pragma unroll
                for (int q = 0; q < CH_MAX_PAIRS; q++) if (q != pref) lds_words[8 + q] = c[q];
            }
            ch_bar();
        }
        uint32_t cur = lds_words[8 + pair];
        if (cur == 0) {
            ch_bar();
            if (tid == 0) {
                uint32_t expected = 0;
                __hip_atomic_compare_exchange_strong(claim + pair, &expected, xcc + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lds_words[3] = expected ? expected : xcc + 1;
            }
            ch_bar();
            cur = lds_words[3];
        }
        if (cur != xcc + 1) continue;

        // ---- the whole chain of a pair this XCD owns
        uint32_t* local = sync + 32 + 32 * pair;
        uint32_t* slot = local + xcc;
        if (k > 0) {                                                       // (the preferred pair's items were drawn with the claim)
            ch_bar();
            if (tid == 0) lds_words[2] = ch_local_add(slot, 1u | (1u << 10) | (1u << 20));
            ch_bar();
        }
        const uint32_t d = lds_words[2];
        const uint32_t f0 = d & 1023u, f1 = (d >> 10) & 1023u, f2 = d >> 20;
        ch_bar();
        CHT(2);
        typename ChainOps<C0>::W w0;
        typename ChainOps<C1>::W w1;
        typename ChainOps<C2>::W w2;
        typename ChainOps<ChainNone>::W wn;
        ok = chain_phase<C0, ChainNone, C1, false>(args->L[0], args->L[1], pair, 0, f0, w0, f1, w1, local, slot, ch_smem, table, lds_words);
        if constexpr (C2::ITEMS != 0) {
            if (ok) ok = chain_phase<C1, C0, C2, true>(args->L[1], args->L[2], pair, 1, f1, w1, f2, w2, local, slot, ch_smem, table, lds_words);
            if (ok) ok = chain_phase<C2, C1, ChainNone, true>(args->L[2], args->L[2], pair, 2, f2, w2, 0u, wn, local, slot, ch_smem, table, lds_words, args->fcw, args->fc_part);
        } else {
            if (ok) ok = chain_phase<C1, C0, C2, true>(args->L[1], args->L[2], pair, 1, f1, w1, f2, w2, local, slot, ch_smem, table, lds_words, args->fcw, args->fc_part);
        }
    }
    CHT(30);
#ifdef HNET_CHAIN_TRACE
    if (tid == 0) g_chain_trace[blockIdx.x * 32 + 31] = xcc + 1 + ((unsigned long long)(lds_words[8 + pref] == xcc + 1) << 8);
#endif
    if (tid == 0 && !ok) atomicOr(args->flag, (uint32_t)CH_FLAG_TIMEOUT);
}

// the layers (sizes fixed by the network: kConvs)
typedef ChainCfg<128, 5, 14, 20, 128, 8, 2, false> ChainL12;       // block_1_2
typedef ChainCfg<128, 3, 7, 10, 256, 8, 1, true> ChainL13;         // block_1_3
typedef ChainCfg<64, 5, 28, 40, 128, 16, 4, false> ChainL22;       // block_2_2
typedef ChainCfg<64, 3, 28, 40, 128, 16, 4, false> ChainLx4;       // block_3_3, block_4_4
typedef ChainCfg<128, 3, 14, 20, 256, 16, 2, false> ChainLx5;      // block_2_3, block_3_4, block_4_5
typedef ChainCfg<256, 3, 7, 10, 256, 8, 1, true> ChainLx6;         // block_2_4, block_3_5, block_4_6

}  // namespace hnet
