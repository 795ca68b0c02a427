// tools/pc_gemm_proto.hip — feasibility prototype: wave-specialised split-bf16 GEMM (one producer wave feeding an LDS
// ring with global_load_lds, four consumer waves doing only ds_read + MFMA), 128x128 tile, one workgroup per CU.
// Timing only (operands are random, the result is not checked).  Shape: the heads GEMM, M 8192 x N 512 x K 5120.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/pc_gemm_proto.hip -o tools/pc_gemm_proto.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128;

// MODE 0: specialised (producer wave(s) + four consumer waves).  MODE 1: consumers only (stale LDS): the MFMA + ds_read
// bound.  MODE 2: producers only: the DMA bound.
// BK 32: LDS rows of 64 B, chunk swizzle (row>>2)&3.  BK 16: rows of 32 B, chunk swizzle (row>>2)&1.
template <int MODE, int BK, int NSTAGE, int NPROD, int PF = 0, int SWP = 0>
__global__ __launch_bounds__(256 + 64 * NPROD) void pc_gemm(const uint16_t* __restrict__ A, size_t a_plane, const uint16_t* __restrict__ W,
                                                            size_t w_plane, float* __restrict__ out, int M, int N, int K) {
    constexpr int TILE_A = BM * BK, TILE_B = BN * BK, STAGE = 3 * (TILE_A + TILE_B);
    constexpr int CH = BK / 8;                                 // 16-byte chunks per row
    constexpr int RPI = 64 / CH;                               // rows per DMA instruction
    constexpr int NI = BM / RPI;                               // instructions per (operand, plane)
    constexpr int NPF = PF ? (BN * 3 * BK * 2 / 64 + 63) / 64 : 0;  // L2 prefetch touches of the W tile: one lane per 64-byte piece
    constexpr int PER_TILE = 6 * NI / NPROD + NPF;             // vector-memory instructions per producer wave and K-tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint16_t* smem = reinterpret_cast<uint16_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int n_iter = K / BK;
    auto swz = [](int row, int chunk) { return (chunk ^ ((row >> 2) & (CH - 1))) * 8; };
    if (wave >= 4) {
        const int pw = wave - 4;
        const int lrow = lane / CH, lphys = lane % CH;
        const uint16_t* asrc[NI];
        const uint16_t* wsrc[NI];
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int r = i * RPI + lrow;
            asrc[i] = A + (size_t)((m0 + r) >> 5) * K + (lphys ^ ((r >> 2) & (CH - 1))) * 8;   // 32 MC samples share a feature row
            wsrc[i] = W + (size_t)(n0 + r) * K + (lphys ^ ((r >> 2) & (CH - 1))) * 8;
        }
        auto issue = [&](int it, int stage) {
            uint16_t* sbase = smem + stage * STAGE;
#pragma unroll
            for (int pl = 0; pl < 3; pl++)
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    if (NPROD == 1 || (i & 1) == pw)
                        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(asrc[i] + pl * a_plane + it * BK),
                                                         (void __attribute__((address_space(3)))*)(sbase + pl * TILE_A + i * RPI * BK), 16, 0, 0);
                    if (NPROD == 1 || (i & 1) == pw)
                        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(wsrc[i] + pl * w_plane + it * BK),
                                                         (void __attribute__((address_space(3)))*)(sbase + 3 * TILE_A + pl * TILE_B + i * RPI * BK), 16, 0, 0);
                }
        };
        // L2 prefetch: touch every 64-byte piece of the W tile PF K-tiles ahead (result discarded)
        auto prefetch = [&](int it) {
#pragma unroll
            for (int j = 0; j < NPF; j++) {
                const int piece = j * 64 + lane;                         // (plane, row, 64-byte half of the BK*2-byte row segment)
                constexpr int PPR = BK * 2 / 64 > 0 ? BK * 2 / 64 : 1;   // pieces per row
                const int pl = piece / (BN * PPR), rem = piece % (BN * PPR), row = rem / PPR, pc = rem % PPR;
                const int itc = it < n_iter ? it : n_iter - 1;
                const uint16_t* src = W + (size_t)(pl < 3 ? pl : 2) * w_plane + (size_t)(n0 + row) * K + itc * BK + pc * 32;
                // DMA of 4 bytes per lane into a scratch area behind the ring: no VGPR result that could land late
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                                 (void __attribute__((address_space(3)))*)(smem + NSTAGE * STAGE), 4, 0, 0);
            }
        };
        if (MODE != 1)
            for (int t = 0; t < NSTAGE - 1 && t < n_iter; t++) { if (PF) prefetch(t + PF); issue(t, t); }
        for (int it = 0; it < n_iter; it++) {
            if (MODE != 1) {
                // tiles it+1 .. it+NSTAGE-2 may still be in flight
                if (it + NSTAGE - 2 < n_iter) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE * (NSTAGE - 2)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (MODE != 2) __builtin_amdgcn_s_barrier();
            if (MODE != 1 && it + NSTAGE - 1 < n_iter) { if (PF) prefetch(it + NSTAGE - 1 + PF); issue(it + NSTAGE - 1, (it + NSTAGE - 1) % NSTAGE); }
        }
        return;
    }
    if (MODE == 2) return;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, fh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
    if constexpr (SWP && BK == 32) {
        // software-pipelined: the barrier that publishes tile it+1 sits between the two k16 steps of tile it, so that the
        // fragment reads of the next tile are in flight while the second half of this tile's MFMAs executes
        auto rd = [&](const uint16_t* As, const uint16_t* Bs, int step, bf16x8 (&af)[2][3], bf16x8 (&bf)[2][3]) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int r = wm * 64 + i * 32 + frow;
#pragma unroll
                for (int pl = 0; pl < 3; pl++) af[i][pl] = *reinterpret_cast<const bf16x8*>(&As[pl * TILE_A + r * BK + swz(r, 2 * step + fh)]);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int r = wn * 64 + j * 32 + frow;
#pragma unroll
                for (int pl = 0; pl < 3; pl++) bf[j][pl] = *reinterpret_cast<const bf16x8*>(&Bs[pl * TILE_B + r * BK + swz(r, 2 * step + fh)]);
            }
        };
        auto mm = [&](bf16x8 (&af)[2][3], bf16x8 (&bf)[2][3]) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                }
        };
        bf16x8 a0[2][3], b0[2][3], a1[2][3], b1[2][3];
        __builtin_amdgcn_s_barrier();                                    // tile 0 landed
        rd(smem, smem + 3 * TILE_A, 0, a0, b0);
        for (int it = 0; it < n_iter; it++) {
            const uint16_t* As = smem + (it % NSTAGE) * STAGE;
            rd(As, As + 3 * TILE_A, 1, a1, b1);
            mm(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (it + 1 < n_iter) {
                __builtin_amdgcn_s_barrier();                            // tile it+1 landed
                const uint16_t* An = smem + ((it + 1) % NSTAGE) * STAGE;
                rd(An, An + 3 * TILE_A, 0, a0, b0);
            }
            mm(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
    for (int it = 0; it < n_iter; it++) {
            __builtin_amdgcn_s_barrier();
            const uint16_t* As = smem + (it % NSTAGE) * STAGE;
            const uint16_t* Bs = As + 3 * TILE_A;
    #pragma unroll
            for (int step = 0; step < BK / 16; step++) {
                bf16x8 af[2][3], bf[2][3];
    #pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int r = wm * 64 + i * 32 + frow;
    #pragma unroll
                    for (int pl = 0; pl < 3; pl++) af[i][pl] = *reinterpret_cast<const bf16x8*>(&As[pl * TILE_A + r * BK + swz(r, 2 * step + fh)]);
                }
    #pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int r = wn * 64 + j * 32 + frow;
    #pragma unroll
                    for (int pl = 0; pl < 3; pl++) bf[j][pl] = *reinterpret_cast<const bf16x8*>(&Bs[pl * TILE_B + r * BK + swz(r, 2 * step + fh)]);
                }
    #pragma unroll
                for (int i = 0; i < 2; i++)
    #pragma unroll
                    for (int j = 0; j < 2; j++) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                    }
            }
        }
    }
    const int col = lane & 31, rbase = 4 * fh;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rbase, n = n0 + wn * 64 + j * 32 + col;
                out[(size_t)m * N + n] = acc[i][j][r];
            }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int BK, int NSTAGE, int NPROD, int PF = 0, int SWP = 0>
static int run(const char* name, const uint16_t* A, size_t ap, const uint16_t* W, size_t wp, float* out, int M, int N, int K) {
    constexpr int LDS_BYTES = NSTAGE * 3 * (BM + BN) * BK * 2 + 256;
    auto kern = pc_gemm<MODE, BK, NSTAGE, NPROD, PF, SWP>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    dim3 grid(M / BM, N / BN), block(256 + 64 * NPROD);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, grid, block, LDS_BYTES, 0, A, ap, W, wp, out, M, N, K);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kern, grid, block, LDS_BYTES, 0, A, ap, W, wp, out, M, N, K);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    const double flop = 2.0 * M * N * (double)K;
    std::printf("%-52s %.4f ms   %.1f TFLOP/s fp32-equivalent  (%.0f TFLOP/s bf16 MFMA)\n", name, ms, flop / (ms * 1e-3) / 1e12, 6 * flop / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    const int M = 8192, N = 512, K = 5120;
    const size_t ap = (size_t)(M / 32) * K, wp = (size_t)N * K;
    uint16_t *A, *W; float* out;
    CK(hipMalloc(&A, 3 * ap * 2)); CK(hipMalloc(&W, 3 * wp * 2)); CK(hipMalloc(&out, (size_t)M * N * 4));
    std::vector<uint16_t> h(3 * (ap > wp ? ap : wp));
    uint32_t s = 1;
    for (size_t i = 0; i < h.size(); i++) { s = s * 1664525u + 1013904223u; h[i] = (uint16_t)(0x3C00 + ((s >> 16) & 0x3FF)); }
    CK(hipMemcpy(A, h.data(), 3 * ap * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), 3 * wp * 2, hipMemcpyHostToDevice));
#define RUN(MODE, BK, NS, NP, name) if (run<MODE, BK, NS, NP>(name, A, ap, W, wp, out, M, N, K)) return 1;
    RUN(0, 32, 3, 1, "BK32 x 3 stages, 1 producer")
    if (run<0, 32, 3, 1, 4>("BK32 x 3 stages, 1 producer, L2 prefetch +4", A, ap, W, wp, out, M, N, K)) return 1;
    if (run<0, 32, 3, 1, 8>("BK32 x 3 stages, 1 producer, L2 prefetch +8", A, ap, W, wp, out, M, N, K)) return 1;
    if (run<0, 32, 3, 1, 16>("BK32 x 3 stages, 1 producer, L2 prefetch +16", A, ap, W, wp, out, M, N, K)) return 1;
    if (run<0, 32, 3, 1, 0, 1>("BK32 x 3, 1 producer, pipelined consumers", A, ap, W, wp, out, M, N, K)) return 1;
    if (run<0, 32, 3, 1, 16, 1>("BK32 x 3, 1 producer, L2 prefetch +16, pipelined", A, ap, W, wp, out, M, N, K)) return 1;
    if (run<1, 32, 3, 1, 0, 1>("BK32 pipelined consumers only", A, ap, W, wp, out, M, N, K)) return 1;
    RUN(0, 32, 3, 2, "BK32 x 3 stages, 2 producers")
    if (run<0, 32, 3, 2, 0, 1>("BK32 x 3, 2 producers, pipelined consumers", A, ap, W, wp, out, M, N, K)) return 1;
    RUN(0, 16, 5, 2, "BK16 x 5 stages, 2 producers")
    RUN(0, 16, 6, 2, "BK16 x 6 stages, 2 producers")
    RUN(0, 16, 4, 1, "BK16 x 4 stages, 1 producer")
    RUN(1, 32, 3, 1, "BK32 consumers only (MFMA + ds_read bound)")
    RUN(1, 16, 5, 2, "BK16 consumers only (MFMA + ds_read bound)")
    RUN(2, 32, 3, 1, "BK32 producer only (DMA bound)")
    RUN(2, 16, 5, 2, "BK16 producer only (DMA bound)")
    return 0;
}
