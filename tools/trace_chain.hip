// tools/trace_chain.hip — where a one-XCD tail chain (csrc/chain_lat.h) spends its time: the block-3/4 chain (64 -> 128 -> 256 -> 256) alone, timed with HIP events
// and with in-kernel 100-MHz stamps per phase of every owner workgroup.   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DHNET_CHAIN_TRACE tools/trace_chain.hip -o tools/trace_chain.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../cuahn_vio_amd/csrc/chain_lat.h"
using namespace hnet;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class C> static int layer(ChainLayer& L, const uint16_t* in, size_t in_plane, int B) {
    uint16_t* w; float* bias;
    CK(hipMalloc(&w, C::WFRAG_HALVES * 2));
    std::vector<uint16_t> h(C::WFRAG_HALVES);
    uint32_t s = 777;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x2000 + ((s >> 16) & 0x3FF)); }       // small positive fp16 weights
    CK(hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&bias, C::COUT * 4)); CK(hipMemset(bias, 0, C::COUT * 4));
    L.in = in; L.in_plane = in_plane; L.wfrag = w; L.bias = bias;
    const size_t on = (size_t)B * C::HO * C::WO * C::COUT;
    if (C::OUT32) { CK(hipMalloc(&L.out32, on * 4)); L.out16 = nullptr; L.out_plane = 0; }
    else { CK(hipMalloc(&L.out16, 2 * on * 2)); L.out_plane = on; L.out32 = nullptr; }
    return 0;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1, reps = 50;
    typedef ChainLx4 C0; typedef ChainLx5 C1; typedef ChainLx6 C2;
    ChainArgs a = {};
    const size_t n_in = (size_t)B * C0::HI * C0::WI * C0::CIN;
    uint16_t* in; CK(hipMalloc(&in, 2 * n_in * 2));
    { std::vector<uint16_t> h(2 * n_in); uint32_t s = 5; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3000 + ((s >> 16) & 0x3FF)); } CK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice)); }
    if (layer<C0>(a.L[0], in, n_in, B)) return 1;
    if (layer<C1>(a.L[1], a.L[0].out16, a.L[0].out_plane, B)) return 1;
    if (layer<C2>(a.L[2], a.L[1].out16, a.L[1].out_plane, B)) return 1;
    uint32_t* sync; CK(hipMalloc(&sync, 2 * CH_SYNC_WORDS * 4)); CK(hipMemset(sync, 0, 2 * CH_SYNC_WORDS * 4));
    CK(hipMalloc(&a.flag, 4)); CK(hipMemset(a.flag, 0, 4));
#ifdef HNET_CHAIN_TRACE
    unsigned long long* tr; CK(hipMalloc(&tr, 256 * 32 * 8)); CK(hipMemset(tr, 0, 256 * 32 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_chain_trace), &tr, sizeof(tr)));
#endif
    auto kern = tail_chain_kernel<C0, C1, C2>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BYTES));
    hipStream_t st; CK(hipStreamCreate(&st));
    int nl = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; i++) { hipLaunchKernelGGL(kern, dim3(256), dim3(CH_NT), CH_LDS_BYTES, st, a, sync + (nl & 1) * CH_SYNC_WORDS, sync + ((nl + 1) & 1) * CH_SYNC_WORDS, B); nl++; }
    CK(hipStreamSynchronize(st));
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; i++) { hipLaunchKernelGGL(kern, dim3(256), dim3(CH_NT), CH_LDS_BYTES, st, a, sync + (nl & 1) * CH_SYNC_WORDS, sync + ((nl + 1) & 1) * CH_SYNC_WORDS, B); nl++; }
    hipEventRecord(e1, st);
    CK(hipStreamSynchronize(st));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint32_t flag; CK(hipMemcpy(&flag, a.flag, 4, hipMemcpyDeviceToHost));
    std::printf("chain x_4 -> x_5 -> x_6, batch %d: %.2f us per launch back to back (flag %u)\n", B, 1e3 * ms / reps, flag);
#ifndef HNET_CHAIN_TRACE
    return 0;       // (built without the stamps: the event timing of the kernel as the library runs it)
#else
    // one more launch, alone, for the stamps
    CK(hipMemset(tr, 0, 256 * 32 * 8));
    { hipLaunchKernelGGL(kern, dim3(256), dim3(CH_NT), CH_LDS_BYTES, st, a, sync + (nl & 1) * CH_SYNC_WORDS, sync + ((nl + 1) & 1) * CH_SYNC_WORDS, B); nl++; }
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h(256 * 32);
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    int xcd_count[8] = {};
    for (int b = 0; b < 256; b++) { if (h[b * 32]) t0 = std::min(t0, h[b * 32]); xcd_count[((h[b * 32 + 31] & 0xFF) - 1) & 7]++; }
    std::printf("workgroups per XCD:"); for (int x = 0; x < 8; x++) std::printf(" %d", xcd_count[x]); std::printf("\n");
    static const char* names[32] = {"entry", "claim known", "items drawn",
        "p0 weights issued", "p0 wait done", "p0 region staged", "p0 MFMA done", "p0 stored", "p0 signalled",
        "p1 weights issued", "p1 wait done", "p1 region staged", "p1 MFMA done", "p1 stored", "p1 signalled",
        "p2 weights issued", "p2 wait done", "p2 region staged", "p2 MFMA done", "p2 stored", "p2 signalled",
        "p0 enter", "p0 w issued", "p0 table", "p1 enter", "p1 w issued", "p1 table", "p2 enter", "p2 w issued", "p2 table"};
    for (int sl = 0; sl < 31; sl++) {
        std::vector<double> v;
        for (int b = 0; b < 256; b++) if ((h[b * 32 + 31] >> 8) && h[b * 32 + sl]) v.push_back((double)(h[b * 32 + sl] - t0) * 0.01);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        std::printf("  %-20s n=%3zu  min %6.2f  median %6.2f  max %6.2f us\n", sl == 30 ? "exit" : (names[sl] ? names[sl] : "?"), v.size(), v.front(), v[v.size() / 2], v.back());
    }
    { std::vector<double> v; for (int b = 0; b < 256; b++) if (!(h[b * 32 + 31] >> 8) && h[b * 32 + 30]) v.push_back((double)(h[b * 32 + 30] - t0) * 0.01);
      if (!v.empty()) { std::sort(v.begin(), v.end()); std::printf("  outsiders' exit       n=%3zu  min %6.2f  median %6.2f  max %6.2f us\n", v.size(), v.front(), v[v.size() / 2], v.back()); } }
    return 0;
#endif
}
