// rccl_gather_example.cpp — the C++ side of the multi-GPU split (north_star: "host code stays C++ ... RCCL gather of the
// per-sample homographies / covariances over xGMI").  The reference has no collective to cite: pytorch::HomographyNet picks ONE device
// (HomographyNet.cpp:10-16).  What a C++ integrator adds around the C ABI of include/hnet.h:
//
//   one hnet context per GPU (weights replicated, 26 MB) -> the pairs of a batch split contiguously over the GPUs -> every GPU runs
//   hnet_infer_batch_packed_device on its shard, on its own stream: the ensemble kernel writes the packed [nb, 72] records (mean | cov) itself
//   (round 4; rounds 2 - 3 packed with two strided device copies) -> ONE ncclAllGather of nb x 288 bytes per GPU (latency bound; xGMI bandwidth is irrelevant) -> every GPU
//   holds all pairs' outputs in pair order.
//
// One process drives all visible GPUs here (ncclCommInitAll + group calls); a one-process-per-GPU deployment replaces ncclCommInitAll
// by ncclGetUniqueId / ncclCommInitRank and keeps the rest.  On a 1-GPU box this is a 1-rank communicator: the same calls execute.
//   usage: rccl_gather_example <weights.hnw> [pairs_per_gpu] [--time K]
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hnet.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 3; } } while (0)
#define HNET_CHECK(x) do { int s_ = (x); if (s_ != HNET_OK) { std::fprintf(stderr, "%s: %s\n", #x, hnet_status_string(s_)); return 4; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s weights.hnw [pairs_per_gpu]\n", argv[0]); return 1; }
    const int nb = argc > 2 ? std::atoi(argv[2]) : 8;
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    if (ndev < 1 || nb < 1) return 1;
    const size_t npix = (size_t)HNET_IMG_ROWS * HNET_IMG_COLS;

    std::vector<int> devs(ndev);
    for (int d = 0; d < ndev; d++) devs[d] = d;
    std::vector<ncclComm_t> comm(ndev);
    NCCL_OK(ncclCommInitAll(comm.data(), ndev, devs.data()));

    struct PerDev { hnet_ctx* ctx; hipStream_t s; uint8_t *prev, *curr; float *mean, *cov, *out, *all; };
    std::vector<PerDev> g(ndev);
    // synthetic frames: pair p of the whole batch is a smooth pattern and the same pattern shifted by (p % 5) pixels
    std::vector<uint8_t> hp(nb * npix), hc(nb * npix);
    for (int d = 0; d < ndev; d++) {
        HIP_OK(hipSetDevice(d));
        hnet_config cfg;
        hnet_default_config(&cfg);
        cfg.device_id = d;
        cfg.mc_samples = 16;
        cfg.max_batch = nb;
        HNET_CHECK(hnet_create(&cfg, argv[1], &g[d].ctx));
        HIP_OK(hipStreamCreateWithFlags(&g[d].s, hipStreamNonBlocking));
        for (int b = 0; b < nb; b++) {
            const int p = d * nb + b;                                  // global pair index: contiguous shards
            for (int v = 0; v < HNET_IMG_ROWS; v++)
                for (int u = 0; u < HNET_IMG_COLS; u++) {
                    hp[b * npix + (size_t)v * HNET_IMG_COLS + u] = (uint8_t)(128 + 60 * ((u / 16 + v / 12 + p) % 2) + ((u * 7 + v * 3) % 23));
                    const int us = u + p % 5;
                    hc[b * npix + (size_t)v * HNET_IMG_COLS + u] = (uint8_t)(128 + 60 * ((us / 16 + v / 12 + p) % 2) + ((us * 7 + v * 3) % 23));
                }
        }
        HIP_OK(hipMalloc((void**)&g[d].prev, nb * npix));
        HIP_OK(hipMalloc((void**)&g[d].curr, nb * npix));
        HIP_OK(hipMalloc((void**)&g[d].mean, (size_t)nb * 8 * 4));
        HIP_OK(hipMalloc((void**)&g[d].cov, (size_t)nb * 64 * 4));
        HIP_OK(hipMalloc((void**)&g[d].out, (size_t)nb * 72 * 4));
        HIP_OK(hipMalloc((void**)&g[d].all, (size_t)ndev * nb * 72 * 4));
        HIP_OK(hipMemcpy(g[d].prev, hp.data(), nb * npix, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(g[d].curr, hc.data(), nb * npix, hipMemcpyHostToDevice));
    }

    // one step: forward of every shard into its packed record buffer, all-gather - everything of a GPU on its one stream.  (A streaming caller
    // double-buffers `out` and issues the gather on a second stream behind an event, under the next step's forward: cuahn_vio_amd/dist.py OverlappedGather.)
    for (int d = 0; d < ndev; d++) {
        HIP_OK(hipSetDevice(d));
        HNET_CHECK(hnet_infer_batch_packed_device(g[d].ctx, g[d].prev, g[d].curr, HNET_PIX_U8, nullptr, nb, (uint64_t)d * nb, g[d].out, nullptr, g[d].s));
    }
    NCCL_OK(ncclGroupStart());
    for (int d = 0; d < ndev; d++) NCCL_OK(ncclAllGather(g[d].out, g[d].all, (size_t)nb * 72, ncclFloat, comm[d], g[d].s));
    NCCL_OK(ncclGroupEnd());

    // every GPU must hold every shard's packed outputs, in pair order, bit for bit; no output may be non-finite
    std::vector<std::vector<float>> own(ndev, std::vector<float>((size_t)nb * 72));
    for (int d = 0; d < ndev; d++) {
        HIP_OK(hipSetDevice(d));
        HIP_OK(hipStreamSynchronize(g[d].s));
        int flags = 0;
        HNET_CHECK(hnet_overflow_flag(g[d].ctx, g[d].s, &flags));
        if (flags) { std::fprintf(stderr, "device %d: non-finite outputs (overflow flag %d)\n", d, flags); return 5; }
        HIP_OK(hipMemcpy(own[d].data(), g[d].out, (size_t)nb * 72 * 4, hipMemcpyDeviceToHost));
    }
    std::vector<float> all((size_t)ndev * nb * 72);
    for (int d = 0; d < ndev; d++) {
        HIP_OK(hipSetDevice(d));
        HIP_OK(hipMemcpy(all.data(), g[d].all, all.size() * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < ndev; r++)
            if (std::memcmp(all.data() + (size_t)r * nb * 72, own[r].data(), (size_t)nb * 72 * 4) != 0) {
                std::fprintf(stderr, "device %d: shard %d of the gathered outputs differs from its source\n", d, r);
                return 6;
            }
    }
    // ---- optional: what the gather COSTS when a C++ deployment issues it.  `--time K`: K steps after 20, ms per step
    //   (a) forward only   (b) forward + ncclAllGather on the same stream   (c) double-buffered records, the gather on a second stream behind an event, under the
    //   next step's forward - and BASELINE config 4's shape: one pair, the MC samples of this rank (hnet_infer_mc_partial_device into ONE [2][1][n][8] array),
    //   ncclAllGather of it, hnet_mc_finish_gathered_device on the gathered buffer in place; beside it the same pair on a context that draws 1/8 of the samples
    //   (what a rank of an 8-GPU split computes) - so that "sharding saves x us of heads, the collective costs y us" is a measured statement.
    if (argc > 4 && std::strcmp(argv[3], "--time") == 0) {
        const int K = std::atoi(argv[4]);
        struct Tim { hipStream_t side; hipEvent_t done[2], gathered[2]; float* out2; };
        std::vector<Tim> t(ndev);
        for (int d = 0; d < ndev; d++) {
            HIP_OK(hipSetDevice(d));
            HIP_OK(hipStreamCreateWithFlags(&t[d].side, hipStreamNonBlocking));
            for (int k = 0; k < 2; k++) { HIP_OK(hipEventCreateWithFlags(&t[d].done[k], hipEventDisableTiming)); HIP_OK(hipEventCreateWithFlags(&t[d].gathered[k], hipEventDisableTiming)); }
            HIP_OK(hipMalloc((void**)&t[d].out2, (size_t)nb * 72 * 4));
        }
        auto sync_all = [&]() { for (int d = 0; d < ndev; d++) { (void)hipSetDevice(d); (void)hipStreamSynchronize(g[d].s); (void)hipStreamSynchronize(t[d].side); } };
        for (int mode = 0; mode < 3; mode++) {
            double ms = 0;
            for (int pass = 0; pass < 2; pass++) {             // pass 0 = warm-up
                const int n = pass ? K : 20;
                sync_all();
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < n; i++) {
                    const int k = i & 1;
                    for (int d = 0; d < ndev; d++) {
                        HIP_OK(hipSetDevice(d));
                        float* o = (mode == 2 && k) ? t[d].out2 : g[d].out;
                        if (mode == 2 && i >= 2) HIP_OK(hipStreamWaitEvent(g[d].s, t[d].gathered[k], 0));      // the gather that last read this record buffer
                        HNET_CHECK(hnet_infer_batch_packed_device(g[d].ctx, g[d].prev, g[d].curr, HNET_PIX_U8, nullptr, nb, (uint64_t)(i * ndev + d) * nb, o, nullptr, g[d].s));
                        if (mode == 2) { HIP_OK(hipEventRecord(t[d].done[k], g[d].s)); HIP_OK(hipStreamWaitEvent(t[d].side, t[d].done[k], 0)); }
                    }
                    if (mode) {
                        NCCL_OK(ncclGroupStart());
                        for (int d = 0; d < ndev; d++)
                            NCCL_OK(ncclAllGather((mode == 2 && k) ? t[d].out2 : g[d].out, g[d].all, (size_t)nb * 72, ncclFloat, comm[d], mode == 2 ? t[d].side : g[d].s));
                        NCCL_OK(ncclGroupEnd());
                        if (mode == 2) for (int d = 0; d < ndev; d++) { HIP_OK(hipSetDevice(d)); HIP_OK(hipEventRecord(t[d].gathered[k], t[d].side)); }
                    }
                }
                sync_all();
                ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n;
            }
            std::printf("RCCL_GATHER_TIME ranks=%d pairs_per_gpu=%d mode=%s ms_per_step=%.4f\n", ndev, nb,
                        mode == 0 ? "forward_only" : mode == 1 ? "gather_same_stream" : "gather_side_stream", ms);
        }
        // config 4: one pair, N = 32
        {
            const int N = 32;
            HIP_OK(hipSetDevice(0));
            hnet_ctx *cf = nullptr, *cs = nullptr;
            hnet_config cfg;
            hnet_default_config(&cfg);
            cfg.mc_samples = N; cfg.max_batch = 1;
            HNET_CHECK(hnet_create(&cfg, argv[1], &cf));
            cfg.mc_sample_begin = 0; cfg.mc_sample_end = N / 8;            // the samples of rank 0 of an 8-GPU split
            HNET_CHECK(hnet_create(&cfg, argv[1], &cs));
            float *both = nullptr, *gath = nullptr, *h1 = nullptr, *o72 = nullptr;
            HIP_OK(hipMalloc((void**)&both, (size_t)2 * N * 8 * 4)); HIP_OK(hipMalloc((void**)&gath, (size_t)ndev * 2 * N * 8 * 4));
            HIP_OK(hipMalloc((void**)&h1, 9 * 4)); HIP_OK(hipMalloc((void**)&o72, 72 * 4));
            auto time_it = [&](int what) -> double {
                double ms = 0;
                for (int pass = 0; pass < 2; pass++) {
                    const int n = pass ? K : 20;
                    (void)hipStreamSynchronize(g[0].s);
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int i = 0; i < n; i++) {
                        if (what == 0) (void)hnet_infer_batch_packed_device(cf, g[0].prev, g[0].curr, HNET_PIX_U8, nullptr, 1, (uint64_t)i, o72, nullptr, g[0].s);
                        else {
                            hnet_ctx* c = what == 1 ? cf : cs;
                            const int nl = what == 1 ? N : N / 8;
                            (void)hnet_infer_mc_partial_device(c, g[0].prev, g[0].curr, HNET_PIX_U8, nullptr, 1, (uint64_t)i, both, both + (size_t)nl * 8, h1, g[0].s);
                            if (what != 3) {
                                (void)ncclAllGather(both, gath, (size_t)2 * nl * 8, ncclFloat, comm[0], g[0].s);      // (rank 0's call of the collective; a 1-rank communicator completes it alone)
                                (void)hnet_mc_finish_gathered_device(c, gath, 1, nl, h1, 1, o72, g[0].s);
                            }
                        }
                    }
                    (void)hipStreamSynchronize(g[0].s);
                    ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n;
                }
                return ms;
            };
            if (ndev == 1) {
                const double a = time_it(0), b = time_it(1), c2 = time_it(2), d3 = time_it(3);
                std::printf("RCCL_CONFIG4_TIME one_pair_N32 ms: plain_forward=%.4f  partial+allgather+finish(all 32 samples)=%.4f  partial(4 of 32 samples)+allgather+finish=%.4f  "
                            "partial(4 of 32 samples) alone=%.4f\n", a, b, c2, d3);
            }
            hnet_destroy(cs); hnet_destroy(cf);
            (void)hipFree(both); (void)hipFree(gath); (void)hipFree(h1); (void)hipFree(o72);
        }
        for (int d = 0; d < ndev; d++) {
            (void)hipSetDevice(d);
            (void)hipFree(t[d].out2); (void)hipStreamDestroy(t[d].side);
            for (int k = 0; k < 2; k++) { (void)hipEventDestroy(t[d].done[k]); (void)hipEventDestroy(t[d].gathered[k]); }
        }
    }
    int ver = 0;
    NCCL_OK(ncclGetVersion(&ver));
    std::printf("RCCL_GATHER_OK ranks=%d pairs_per_gpu=%d rccl_version=%d  pair0 mean = %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f\n", ndev, nb, ver,
                all[0], all[1], all[2], all[3], all[4], all[5], all[6], all[7]);
    for (int d = 0; d < ndev; d++) {
        HIP_OK(hipSetDevice(d));
        hnet_destroy(g[d].ctx);
        (void)hipFree(g[d].prev); (void)hipFree(g[d].curr); (void)hipFree(g[d].mean); (void)hipFree(g[d].cov); (void)hipFree(g[d].out); (void)hipFree(g[d].all);
        (void)hipStreamDestroy(g[d].s);
        ncclCommDestroy(comm[d]);
    }
    return 0;
}
