#!/usr/bin/env python3
"""bench.py — HomographyNet hot-path benchmark on MI355X (contract: see the task statement / DESIGN.md §measurement).

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one forward of the full 4-block HomographyNet (MC-dropout N=32) over one batch of synthetic
320x224 frame pairs per GPU, inputs resident in HBM, outputs [B,8]+[B,64] left in HBM; with N>1 every rank
processes its own shard of the pairs (weak scaling) and the per-pair outputs are all-gathered over RCCL
(288 B per pair, the only exchange the path has).  `value` = frame-pair homography predictions per second over
all GPUs.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frame pairs per GPU per step")
    ap.add_argument("--mc", type=int, default=32, help="MC-dropout samples N")
    ap.add_argument("--variant", default="full", choices=["full", "prior3", "prior2", "prior1"])
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16x3"],
                    help="fp32: exact fp32 MFMA; bf16x3: fp32-grade split-bf16 MFMA (both pass the same parity tests)")
    ap.add_argument("--mode", default="pairs", choices=["pairs", "mc", "stream"],
                    help="pairs (default, the headline metric): frame pairs sharded over the GPUs.  mc (BASELINE config 4): "
                         "the SAME pairs on every rank, the N MC-dropout samples sharded over the ranks, one all-gather of the "
                         "per-sample head outputs, two-pass ensemble on every rank.  stream (BASELINE config 5, PCIe-inclusive, "
                         "never the headline value): every step's pairs start in pinned HOST memory; H2D of step i+1 on a copy "
                         "stream overlaps the forward of step i, outputs are copied back to pinned host memory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    return ap.parse_args()


def _calibrated_rate(one, set_threads, budget_s):
    """pairs/s of `one(i)` after picking the thread count that is fastest on this host (more threads than the small
    per-layer loops can feed only add fork/join cost: 256 threads ran 60x slower than 8 on the GPU box)"""
    avail = os.cpu_count() or 1
    best, cores = None, 1
    for th in [t for t in (4, 8, 16, 32, 64) if t <= avail] or [1]:
        set_threads(th)
        one(0)
        t0 = time.perf_counter()
        one(1)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, th
    set_threads(cores)
    one(0)
    t0 = time.perf_counter()
    for i in range(2):
        one(i)
    per = (time.perf_counter() - t0) / 2
    n = int(max(4, min(5000, budget_s / max(per, 1e-4))))
    t0 = time.perf_counter()
    for i in range(n):
        one(i)
    dt = time.perf_counter() - t0
    return n, dt, cores


def cpu_baseline(blob, state, prev, curr, prior, variant, n_mc, budget_s):
    """CPU baselines on this box's host cores, on a bounded sample, one pair at a time like the reference:
    (i) `value`: the forward on libtorch's CPU operators (oracle/torch_cpu.py — our restatement of the computation the
        reference's TorchScript file runs; the reference's own .pt/.py cannot travel to this box), kind "port";
    (ii) `c_port`: the oracle's plain-fp32 C build with OpenMP (oracle/liboracle_f32.so)."""
    import torch
    from oracle import pyoracle, torch_cpu
    btr = {"full": 3, "prior3": 3, "prior2": 2, "prior1": 1}[variant]
    net = torch_cpu.TorchCpuNet(state)

    def one_t(i):
        j = i % prev.shape[0]
        net.forward(prev[j], curr[j], None if variant == "full" else prior[j], btr, n_mc, 0.05, 1, i)

    n, dt, cores = _calibrated_rate(one_t, torch.set_num_threads, budget_s / 2)
    res = {"value": round(n / dt, 2), "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": f"{n} frame pairs, {variant} model, N={n_mc}, one pair at a time (the reference is batch-1), libtorch {torch.__version__} "
                     f"CPU operators (oracle/torch_cpu.py) on {cores} threads of {os.cpu_count()} host CPUs, {dt:.1f} s",
           "ms_per_pair": round(1e3 * dt / n, 3)}
    # the same on ONE thread (SURVEY.md §8d asks for both; the reference measured 27.7 ms per pair single-threaded)
    torch.set_num_threads(1)
    one_t(0)
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < 1.5:
        one_t(n1)
        n1 += 1
    res["single_thread"] = {"value": round(n1 / (time.perf_counter() - t0), 2), "unit": "pairs/s", "cores": 1}
    torch.set_num_threads(cores)
    orc = pyoracle.Oracle(blob, f32=True)

    def one_c(i):
        j = i % prev.shape[0]
        orc.forward(prev[j], curr[j], None if variant == "full" else prior[j], btr, n_mc, 0.05, 1, i)

    n, dt, cores = _calibrated_rate(one_c, orc.lib.oracle_set_threads, budget_s / 2)
    res["c_port"] = {"value": round(n / dt, 2), "unit": "pairs/s", "cores": cores,
                     "sample": f"{n} frame pairs, oracle/liboracle_f32.so with OpenMP on {cores} threads, {dt:.1f} s"}
    return res


def measured_traffic(kernel_substr, batch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE,
    collected separately with tools/profile_round.sh at batch 256); None when no matching profile is committed."""
    import csv
    import glob
    if batch != 256:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.csv")))
    if not files:
        return None
    with open(files[-1]) as f:
        rows = list(csv.DictReader(l for l in f if not l.startswith("#")))
    for r in rows:
        if kernel_substr in r["kernel"]:
            return (float(r["FETCH_SIZE_KiB_full_batch_launch"]) + float(r["WRITE_SIZE_KiB_full_batch_launch"])) * 1024.0
    return None


# stage name -> substring of the HIP kernel name in the rocprof tables
KERNEL_OF_STAGE = {"block_4_0+4_1": "block4_fused_kernel", "heads_fc1": "HeadLoaderS3, 128", "block_3_1": "conv_patch_s2_kernel<5>",
                   "block_2_2": "ConvLoaderS3<64, 5, 2, 32>"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    import torch.distributed as dist

    from cuahn_vio_amd import dist as hdist
    from cuahn_vio_amd import synth, weights
    from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # HNET_BENCH_SHARED_GPU=1: rehearsal of the N > 1 code path on a box with ONE GPU (all ranks on cuda:0, gloo for the
    # gathers; RCCL refuses two ranks on one device).  The numbers of such a run mean nothing.
    shared = os.environ.get("HNET_BENCH_SHARED_GPU") == "1"
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    B, n_mc = args.batch, args.mc
    blob = weights.pack_state_dict(weights.synthetic_state(0))
    n_distinct = min(B, 32)
    prev_h, curr_h, prior_h, _ = synth.make_batch(1000 + rank * n_distinct, n_distinct)
    reps = (B + n_distinct - 1) // n_distinct
    prev = torch.from_numpy(np.tile(prev_h, (reps, 1, 1))[:B]).to(dev)
    curr = torch.from_numpy(np.tile(curr_h, (reps, 1, 1))[:B]).to(dev)
    prior = torch.from_numpy(np.tile(prior_h, (reps, 1))[:B]).to(dev)
    d_prior = prior.data_ptr() if args.variant != "full" else None
    out = torch.zeros(B, 72, device=dev)            # [mean8 | cov64] per pair
    mean, cov = torch.zeros(B, 8, device=dev), torch.zeros(B, 64, device=dev)
    gathered = torch.zeros(world * B, 72, device=dev) if world > 1 else None

    prec = {"fp32": 0, "bf16x3": 2}[args.precision]
    mc_mode = args.mode == "mc"
    shard = hdist.shard_range(n_mc, world, rank) if mc_mode else None
    if mc_mode and n_mc % world:
        raise SystemExit("--mode mc needs N divisible by the number of GPUs")
    eng = HnetEngine(blob, variant=args.variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=1, max_batch=B, device_id=local_rank,
                     precision=prec, mc_shard=shard)
    # everything of a step — the forward's kernels, the packing of the outputs and the RCCL gather — is enqueued on ONE
    # ordinary (non-default) stream: ordered by the stream, no legacy-default-stream semantics involved
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    sp = stream.cuda_stream
    if mc_mode:
        n_loc = shard[1] - shard[0]
        ms_loc, lv_loc = torch.zeros(B, n_loc, 8, device=dev), torch.zeros(B, n_loc, 8, device=dev)
        h1 = torch.zeros(B, 9, device=dev)

    stream_mode = args.mode == "stream"
    if stream_mode:
        host_prev = [torch.from_numpy(np.tile(prev_h, (reps, 1, 1))[:B]).pin_memory() for _ in range(2)]
        host_curr = [torch.from_numpy(np.tile(curr_h, (reps, 1, 1))[:B]).pin_memory() for _ in range(2)]
        host_prior = [torch.from_numpy(np.tile(prior_h, (reps, 1))[:B]).pin_memory() for _ in range(2)]
        host_out = [torch.zeros(B, 72).pin_memory() for _ in range(2)]
        dbuf = [(torch.empty_like(prev), torch.empty_like(curr), torch.empty_like(prior)) for _ in range(2)]
        copy_stream, comp_stream = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        ev_ready = [torch.cuda.Event() for _ in range(2)]
        ev_free = [torch.cuda.Event() for _ in range(2)]

        def upload(i):
            k = i % 2
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ev_free[k])                      # the forward that last read this buffer has finished
                dbuf[k][0].copy_(host_prev[k], non_blocking=True)
                dbuf[k][1].copy_(host_curr[k], non_blocking=True)
                dbuf[k][2].copy_(host_prior[k], non_blocking=True)
                ev_ready[k].record(copy_stream)

        for k in range(2):
            ev_free[k].record(comp_stream)
        upload(0)

    def step(i):
        if stream_mode:
            k = i % 2
            skip = os.environ.get("HNET_STREAM_SKIP", "")              # experiments: "copy" / "compute"
            comp_stream.wait_event(ev_ready[k])
            if skip != "compute":
                eng.infer_batch_device(dbuf[k][0].data_ptr(), dbuf[k][1].data_ptr(), PIX_U8, dbuf[k][2].data_ptr() if args.variant != "full" else None,
                                       B, (rank * 1000003 + i) * B, mean.data_ptr(), cov.data_ptr(), None, comp_stream)
            ev_free[k].record(comp_stream)
            with torch.cuda.stream(comp_stream):
                hdist.pack_outputs(mean, cov, out)
                host_out[k].copy_(out, non_blocking=True)
            # the next step's pairs cross PCIe while this step computes (enqueued AFTER the forward: the runtime may hold the
            # calling thread until an H2D copy has been handed to the DMA engine, which must not delay the kernel launches)
            if skip != "copy":
                upload(i + 1)
            return
        if mc_mode:   # trunk replicated, heads for this rank's samples, gather, finish in the reference's two-pass order
            eng.infer_mc_partial_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, B, i * B, ms_loc.data_ptr(),
                                        lv_loc.data_ptr(), h1.data_ptr(), sp)
            if world > 1:
                ms_all, lv_all, _ = hdist.gather_mc_samples(ms_loc, lv_loc, h1)
            else:
                ms_all, lv_all = ms_loc, lv_loc
            eng.mc_finish_device(ms_all.data_ptr(), lv_all.data_ptr(), n_mc, h1.data_ptr(), B, mean.data_ptr(), cov.data_ptr(), sp)
            return
        eng.infer_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, B, (rank * 1000003 + i) * B,
                               mean.data_ptr(), cov.data_ptr(), None, sp)
        if world > 1:
            hdist.gather_outputs(mean, cov, out, gathered)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cpu" if shared else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1e3 * dt / args.steps
    value = (1 if mc_mode else world) * B * args.steps / dt     # mc mode: every rank works on the same B pairs

    res = {
        "metric": "homography preds/sec (frame pairs/s), full 4-block HomographyNet @ 320x224",
        "value": round(value, 1), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if mc_mode else "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "f32 as 3 x bf16 planes (six bf16 MFMAs per product, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"{args.variant} HomographyNet forward, 320x224 u8 frame pairs, MC-dropout N={n_mc} p=0.05, "
                               f"{B} pairs/GPU/step, inputs+outputs resident in HBM",
                   "batch_per_gpu": B, "mc_samples": n_mc, "variant": args.variant, "precision": args.precision,
                   "parallelism": (f"MC-dropout samples sharded {n_mc}/{world} per GPU, trunk replicated, RCCL all_gather of [B,N/R,16]"
                                   if mc_mode else
                                   (f"pairs sharded over {world} GPU(s), RCCL all_gather of [B,72] outputs" if world > 1 else "single GPU")),
                   "weights": "synthetic seed 0 (trained checkpoint not shipped with the reference)"},
        "mc_preds_per_s": round(value * n_mc, 1),
    }

    if stream_mode:
        res["config"]["workload"] += "; STREAMED: inputs start in pinned host memory, H2D overlapped with compute, outputs back to host"
        res["config"]["parallelism"] = "host -> device streaming, double-buffered (PCIe-inclusive; not the headline metric)"
    if rank == 0 and (mc_mode or stream_mode):
        print(json.dumps(res), flush=True)
    if rank == 0 and not mc_mode and not stream_mode:
        # ---- roofline of the dominant kernel: per-launch HIP events on the stream the kernels run on
        stages = eng.stages()
        ms = eng.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, B, 0, mean.data_ptr(), cov.data_ptr(), 5)
        k = int(np.argmax(ms))
        total_flops = sum(f for _, f in stages) * B
        fl = stages[k][1] * B
        ms = [float(x) for x in ms]
        ach = fl / (ms[k] * 1e-3) / 1e12
        res["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_FP32_MATRIX_TFLOPS, 4),
                           "traffic": measured_traffic(KERNEL_OF_STAGE.get(stages[k][0], "\0"), B) if args.precision == "bf16x3" else None,
                           "kernel": stages[k][0], "kernel_ms": round(float(ms[k]), 4),
                           "flops_per_launch": fl}
        res["forward"] = {"gflop_per_pair": round(total_flops / B / 1e9, 4),
                          "tflops_whole_forward": round(total_flops / (ms_per_step * 1e-3) / 1e12, 2),
                          "frac_of_fp32_mfma_peak": round(total_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4),
                          "stage_ms": {n: round(float(m), 4) for (n, _), m in zip(stages, ms)}}
        if not args.no_latency:
            e1 = HnetEngine(blob, variant=args.variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=1, max_batch=1, device_id=local_rank,
                            precision=prec)
            e1.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, 1, 0, mean.data_ptr(), cov.data_ptr(), 20)
            per, _tot = e1.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, 1, 0, mean.data_ptr(), cov.data_ptr(), 200)
            res["latency_batch1_ms"] = {"p50": round(float(np.percentile(per, 50)), 4), "p95": round(float(np.percentile(per, 95)), 4),
                                        "definition": "device time of one pair, inputs/outputs resident (the reference's 'pure network inference')"}
            e1.close()
            # end to end through the reference's class surface: load_current_img (71 KB H2D) + network_inference (forward + 288 B D2H),
            # host wall clock per frame, like VioManager.cpp:188,236 drives it
            import contextlib
            import io
            from cuahn_vio_amd.homography_net import HomographyNet
            with contextlib.redirect_stdout(io.StringIO()):
                net = HomographyNet("bench.hnw", use_prior=args.variant != "full", blocks_to_run={"full": 3, "prior3": 3, "prior2": 2, "prior1": 1}[args.variant],
                                    mc_samples=n_mc, dropout_p=0.05, mc_seed=1, device_id=local_rank, weights_blob=blob, precision=prec)
                e2e = []
                for i in range(220):
                    t0 = time.perf_counter()
                    net.load_current_img(curr_h[i % n_distinct], float(i))
                    net.network_inference(prior_h[i % n_distinct].astype(np.float64), 0)
                    if i >= 20:
                        e2e.append(1e3 * (time.perf_counter() - t0))
            res["latency_batch1_ms"]["end_to_end_p50"] = round(float(np.percentile(e2e, 50)), 4)
            res["latency_batch1_ms"]["end_to_end_p95"] = round(float(np.percentile(e2e, 95)), 4)
            res["latency_batch1_ms"]["end_to_end_definition"] = ("host wall clock of load_current_img + network_inference per frame through the "
                                                                 "HomographyNet class surface (u8 image H2D, forward, outputs D2H), 200 frames after 20")
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(blob, weights.synthetic_state(0), prev_h, curr_h, prior_h, args.variant, n_mc, args.cpu_seconds)
        print(json.dumps(res), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
