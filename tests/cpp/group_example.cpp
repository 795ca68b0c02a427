// group_example.cpp — a C++ caller of hnet_group (include/hnet.h, round 6): independent steps (a server's batches; BASELINE config 3 / config 5's per-GPU share)
// issued round-robin on the N contexts of a group, a consumer stream joined behind them, every step compared with a single context's result bit for bit;
// --time: steps per second on 1 .. N members.      group_example <weights.hnw> <n_ctx> <batch> [--time]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "hnet.h"

#define CK(x) do { if ((x) != hipSuccess) { std::printf("HIP error at %s\n", #x); return 1; } } while (0)
#define HK(x) do { const int rc_ = (x); if (rc_ != HNET_OK) { std::printf("%s: %s\n", #x, hnet_status_string(rc_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 4) { std::printf("usage: %s weights.hnw n_ctx batch [--time]\n", argv[0]); return 2; }
    const int n_ctx = std::atoi(argv[2]), B = std::atoi(argv[3]), steps = 12;
    const bool timing = argc > 4 && !std::strcmp(argv[4], "--time");
    hnet_config cfg;
    hnet_default_config(&cfg);
    cfg.use_prior = 1; cfg.blocks_to_run = 3; cfg.mc_samples = 16; cfg.max_batch = B; cfg.mc_seed = 5;
    const size_t npix = (size_t)HNET_IMG_ROWS * HNET_IMG_COLS;
    std::vector<uint8_t> img(2 * B * npix);
    uint32_t s = 99;
    for (auto& v : img) { s = s * 1664525u + 1013904223u; v = (uint8_t)(s >> 24); }
    std::vector<float> prior((size_t)B * 8);
    for (size_t i = 0; i < prior.size(); i++) prior[i] = (float)((int)(i % 7) - 3);
    uint8_t *d_prev, *d_curr; float *d_prior, *d_ref, *d_out;
    CK(hipMalloc(&d_prev, B * npix)); CK(hipMalloc(&d_curr, B * npix)); CK(hipMalloc(&d_prior, prior.size() * 4));
    CK(hipMalloc(&d_ref, (size_t)steps * B * 72 * 4)); CK(hipMalloc(&d_out, (size_t)steps * B * 72 * 4));
    CK(hipMemcpy(d_prev, img.data(), B * npix, hipMemcpyHostToDevice)); CK(hipMemcpy(d_curr, img.data() + B * npix, B * npix, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_prior, prior.data(), prior.size() * 4, hipMemcpyHostToDevice));

    // reference: one context, the same steps
    hnet_ctx* one = nullptr;
    HK(hnet_create(&cfg, argv[1], &one));
    for (int i = 0; i < steps; i++) HK(hnet_infer_batch_packed_device(one, d_prev, d_curr, HNET_PIX_U8, d_prior, B, 1000u * i, d_ref + (size_t)i * B * 72, nullptr, nullptr));
    HK(hnet_synchronize(one, nullptr));
    hnet_destroy(one);

    hnet_group* g = nullptr;
    HK(hnet_create_group(&cfg, argv[1], n_ctx, &g));
    for (int i = 0; i < steps; i++) {
        int member = -1;
        HK(hnet_group_infer_batch_packed_device(g, d_prev, d_curr, HNET_PIX_U8, d_prior, B, 1000u * i, d_out + (size_t)i * B * 72, nullptr, &member));
        if (member != i % n_ctx) { std::printf("step %d ran on member %d\n", i, member); return 1; }
    }
    // the caller's own stream (a collective, a copy to the host ...) ordered behind every member's steps
    hipStream_t consumer;
    CK(hipStreamCreateWithFlags(&consumer, hipStreamNonBlocking));
    HK(hnet_group_join(g, consumer));
    std::vector<float> out((size_t)steps * B * 72), ref(out.size());
    CK(hipMemcpyAsync(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost, consumer));
    CK(hipStreamSynchronize(consumer));
    CK(hipMemcpy(ref.data(), d_ref, ref.size() * 4, hipMemcpyDeviceToHost));
    int flags = 0;
    HK(hnet_group_overflow_flag(g, &flags));
    if (flags || std::memcmp(out.data(), ref.data(), out.size() * 4)) { std::printf("group results differ from the single context's (flags %d)\n", flags); return 1; }
    std::printf("GROUP_OK members=%d batch=%d steps=%d: every step bit-identical to a single context\n", hnet_group_size(g), B, steps);
    if (timing) {
        for (int it = 0; it < 2; it++) {      // (first pass: warm-up)
            const int n = 200;
            HK(hnet_group_synchronize(g));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; i++) HK(hnet_group_infer_batch_packed_device(g, d_prev, d_curr, HNET_PIX_U8, d_prior, B, 1000u * i, d_out + (size_t)(i % steps) * B * 72, nullptr, nullptr));
            HK(hnet_group_synchronize(g));
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (it) std::printf("GROUP_TIME members=%d batch=%d: %.1f k pairs/s (%.3f ms per step)\n", n_ctx, B, 1e-3 * n * B / dt, 1e3 * dt / n);
        }
    }
    hnet_destroy_group(g);
    return 0;
}
