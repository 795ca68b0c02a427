#!/usr/bin/env python3
"""bench.py — HomographyNet hot-path benchmark on MI355X (contract: see the task statement / DESIGN.md §measurement).

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the driver launches the ranks (torch.distributed.run sets RANK / WORLD_SIZE) or,
when WORLD_SIZE is unset, this script starts them itself as child processes BEFORE anything touches the GPU and relays
rank 0's JSON line.

A "step" is one forward of the full 4-block HomographyNet (MC-dropout N=32) over one batch of synthetic 320x224 frame pairs
per GPU, inputs resident in HBM, outputs [B,8]+[B,64] left in HBM; with N>1 every rank processes its own shard of the pairs
(weak scaling) and the per-pair outputs are all-gathered over RCCL (288 B per pair, the only exchange the path has).
`value` = frame-pair homography predictions per second over all GPUs.  After the timed loop (outside it) pairs of the LAST
step are checked against the CPU oracle; a mismatch makes the run fail.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters / matrix cores (dense; never the 2:1-sparsity figures)
PEAK_FP32_MFMA_TFLOPS = 157.3     # v_mfma_f32_32x32x2_f32 / 16x16x4_f32
PEAK_BF16_MFMA_TFLOPS = 2500.0    # v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16
PEAK_HBM_GBS = 8000.0
TOL_PX = 1e-4                     # |HIP - oracle| gate of the self check in the default arithmetic (tests/conftest.py TOL_PX_VS_ORACLE; north_star's figure)
TOL_PX_REFERENCE_MODES = 1.5e-4   # exact fp32 MFMA / split-bf16: their own fp32 accumulation noise against the double-accumulating oracle + margin (conftest.tol_px_vs_oracle)

# precision -> (hnet_config.precision, bf16/fp32 MFMAs issued per multiply-accumulate, peak of the instruction issued, label)
PRECISIONS = {
    "fp32": (0, 1, PEAK_FP32_MFMA_TFLOPS, "f32"),
    "bf16x3": (2, 6, PEAK_BF16_MFMA_TFLOPS, "f32 as 3 x bf16 planes (six bf16 MFMAs per product, fp32 accumulate)"),
    "bf16": (1, 1, PEAK_BF16_MFMA_TFLOPS, "bf16 operands, fp32 accumulate (REPORTED mode: ~1e-2 px, outside the parity gate)"),
    "f16x2": (3, 3, PEAK_BF16_MFMA_TFLOPS, "f32 as 2 x fp16 planes (three fp16 MFMAs per product, fp32 accumulate)"),   # fp16 MFMA peak = bf16 MFMA peak
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (default 100 = 0.14 s at batch 256: the chip needs ~30 ms under load to settle its clock)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="frame pairs per GPU per step")
    ap.add_argument("--mc", type=int, default=32, help="MC-dropout samples N")
    ap.add_argument("--variant", default="full", choices=["full", "prior3", "prior2", "prior1"])
    ap.add_argument("--precision", default="f16x2", choices=list(PRECISIONS),
                    help="fp32: exact fp32 MFMA; bf16x3: fp32-grade split-bf16 MFMA (six MFMAs per product), passes the same parity tests; "
                         "bf16: plain bf16 operands (BASELINE config 2's 'bf16'), reported with its error, not gated; "
                         "f16x2 (default, the library's default): fp32-grade on the fp16 matrix cores with three MFMAs per product, same parity gates")
    ap.add_argument("--mode", default="pairs", choices=["pairs", "mc", "stream"],
                    help="pairs (default, the headline metric): frame pairs sharded over the GPUs.  mc (BASELINE config 4): "
                         "the SAME pairs on every rank, the N MC-dropout samples sharded over the ranks, one all-gather of the "
                         "per-sample head outputs, two-pass ensemble on every rank.  stream (BASELINE config 5, PCIe-inclusive, "
                         "never the headline value): every step's pairs start in pinned HOST memory; H2D of step i+1 on a copy "
                         "stream overlaps the forward of step i, outputs are copied back to pinned host memory")
    ap.add_argument("--replay", default=None, metavar="SEQ",
                    help="with --mode stream: the pairs are rendered along the committed UZH-FPV trajectory fixture "
                         "tests/golden/replay_<SEQ>.npz (tools/make_replay_fixture.py), priors from the EKF mean propagation")
    ap.add_argument("--pairs-total", type=int, default=None, metavar="P",
                    help="BASELINE config 5's shape: P pairs per step ACROSS the ranks (batch per GPU = P / N, strong scaling) instead of "
                         "--batch pairs per GPU")
    ap.add_argument("--force-collective", action="store_true",
                    help="create the RCCL process group and run the per-step all-gather even with one rank (a 1-rank nccl communicator: the "
                         "collective code path on a single MI355X)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the additional measurements of the default run (sustained window, other arithmetic modes, configs 3 / 5, "
                         "reference launch default latency)")
    ap.add_argument("--sustain-seconds", type=float, default=1.6, help="length of the sustained window of the default run")
    ap.add_argument("--contexts", type=int, default=None, metavar="NC",
                    help="independent steps issued round-robin on NC contexts, each with its own HIP stream and buffers (--mode pairs, one GPU): the launch chain of "
                         "one step runs under the kernels of the others - what a server with independent batches does.  The contexts are an hnet_group (include/hnet.h).  "
                         "Default: 4 in --mode pairs (round 6, three fresh boxes, profiles/r06_ctx_sweep.log: + 8 %% at 256 pairs per step, + 43 %% at 64, + 60 %% at 32; the "
                         "one-context figure of rounds 1 - 4 is reported beside it as `single_context`), 1 in the other modes")
    ap.add_argument("--distinct", type=int, default=None, metavar="D",
                    help="distinct frame pairs in a step's batch (default: all of them; rounds 1 - 5 timed 32 distinct pairs tiled to the batch size)")
    ap.add_argument("--honour-env", action="store_true",
                    help="let HNET_S3_TILE / HNET_FUSE_* / HNET_GRAPH / HNET_WARP_EXACT / HNET_PRECISION of the environment select kernels (A/B experiments).  Default: "
                         "the timed engines are built from this command line only and the variables are IGNORED; whatever HNET_* is set is listed in `env_overrides`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the last step (profiling passes)")
    ap.add_argument("--cpu-seconds", type=float, default=14.0, help="wall-clock budget of the whole cpu_baseline leg")
    ap.add_argument("--config", type=int, default=None, choices=[4, 5],
                    help="BASELINE.json presets for the multi-GPU configurations: 4 = --mode mc --mc 32 --batch 1 (one frame pair, the N = 32 MC-dropout "
                         "samples sharded over the GPUs, RCCL gather of the per-sample head outputs); 5 = --pairs-total 256 --mode stream --replay "
                         "indoor_forward_7 --variant prior3 --mc 16 (UZH-FPV replay, 256 streamed pairs per step across the GPUs, PCIe inclusive, RCCL "
                         "gather of the [B, 72] outputs on a side stream).  Use with --gpus N")
    ap.add_argument("--dry-run", action="store_true",
                    help="rank start-up, process-group creation and the max-over-ranks reduction only (gloo, no GPU): the CPU "
                         "test of the multi-rank launch path")
    a = ap.parse_args(argv)
    if a.config == 4:
        a.mode, a.mc, a.batch = "mc", 32, 1
    elif a.config == 5:
        a.pairs_total, a.mode, a.replay, a.variant, a.mc = 256, "stream", "indoor_forward_7", "prior3", 16
    return a


# ------------------------------------------------------------------------------------------------ multi-rank launch
def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes (torch.distributed.run) before this
    process has imported torch or touched HIP, relay their output, exit with their status.  Never an exec."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    env["HNET_BENCH_SPAWNED"] = "1"
    proc = subprocess.Popen(cmd, env=env)
    return proc.wait()


_INPUT_CACHE = {}      # (replay, rank, distinct pairs) -> host arrays: the default run builds several engines on the same inputs


def time_budget_rate(one, budget_s, n_max=5000):
    """runs one(i) until `budget_s` of wall clock is used (at least 2 calls after one warm-up); returns (calls, seconds)"""
    one(0)
    t0 = time.perf_counter()
    n = 0
    while n < 2 or (time.perf_counter() - t0 < budget_s and n < n_max):
        one(n + 1)
        n += 1
    return n, time.perf_counter() - t0


def cpu_baseline_cpp(state, prev, curr, prior, variant, n_mc, budget_s, avail):
    """the reference's own call sequence in C++ - torch::jit::load + module.forward on one pair at a time (HomographyNet.cpp:89,183-186) -
    on a TorchScript trace of our restatement (oracle/libtorch/): returns the result dict or None when the harness cannot be built"""
    try:
        from oracle import libtorch as lt
        lt.build()
        model = lt.model_path(state, variant, n_mc)
    except Exception as e:                                   # no g++ / torch headers on this box: the Python form below is timed instead
        print(f"bench.py: libtorch C++ harness unavailable ({e}); timing oracle/torch_cpu.py", file=sys.stderr)
        return None
    masks = lt.keep_masks(n_mc, 0.05, 1, 0)
    pr = None if variant == "full" else prior[0]
    best, cores = None, 1
    for th in [t for t in (8, 16, 32) if t <= avail] or [1]:
        r = lt.run(model, prev[0], curr[0], pr, masks, threads=th, seconds=0.3)
        if best is None or r["ms_per_forward"] < best:
            best, cores = r["ms_per_forward"], th
    r = lt.run(model, prev[0], curr[0], pr, masks, threads=cores, seconds=budget_s * 0.35)
    r1 = lt.run(model, prev[0], curr[0], pr, masks, threads=1, seconds=budget_s * 0.08)
    import torch
    return {"value": round(1e3 / r["ms_per_forward"], 2), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{r['forwards']} forwards of one frame pair, {variant} model, N={n_mc}, one pair at a time (the reference is batch-1): "
                      f"torch::jit::load + module.forward in C++ (oracle/libtorch/libtorch_harness.cpp, libtorch {torch.__version__}) of the "
                      f"TorchScript trace of our restatement oracle/torch_cpu.py, {cores} threads of {avail} host CPUs, {r['seconds']:.1f} s",
            "ms_per_pair": round(r["ms_per_forward"], 3),
            "single_thread": {"value": round(1e3 / r1["ms_per_forward"], 2), "unit": "pairs/s", "cores": 1},
            # SURVEY 8(d) also asks for "all host cores": measured ONCE on the 256-CPU host of the round-4 box (gpurun_out/r04_v3_bench.json) and not repeated in
            # every run - a batch-1 forward with 256 intra-op threads takes ~20 s (0.05 pairs/s: oversubscription of libtorch's pool), the leg took 72 s
            "all_host_threads": {"value": 0.05, "unit": "pairs/s", "cores": 256, "measured": "round 4, once (profiles/r04_cpu_all_threads.json); not re-run",
                                 "note": "a batch-1 forward of this size stops scaling at 8 - 32 intra-op threads; 256 threads = 20 s per forward"}}


def cpu_baseline(blob, state, prev, curr, prior, variant, n_mc, budget_s):
    """CPU baselines on this box's host cores, on a bounded sample (the whole leg stays inside `budget_s`), one pair at a
    time like the reference:
    (i) `value`: the forward on libtorch's CPU operators - the C++ TorchScript harness of oracle/libtorch/ (the reference's own
        call sequence; its .pt/.py cannot travel to this box, so the traced module is our restatement), or, when that cannot be
        built here, the same restatement called from Python (oracle/torch_cpu.py); kind "port";
    (ii) `c_port`: the oracle's plain-fp32 C build with OpenMP (oracle/liboracle_f32.so)."""
    import torch
    from oracle import pyoracle, torch_cpu
    t_leg = time.perf_counter()
    btr = {"full": 3, "prior3": 3, "prior2": 2, "prior1": 1}[variant]
    avail = os.cpu_count() or 1
    res = cpu_baseline_cpp(state, prev, curr, prior, variant, n_mc, budget_s, avail)
    if res is None:
        res = cpu_baseline_py(state, prev, curr, prior, variant, n_mc, budget_s, avail, btr)
    orc = pyoracle.Oracle(blob, f32=True)
    c_threads = min(16, avail)
    orc.lib.oracle_set_threads(c_threads)

    def one_c(i):
        j = i % prev.shape[0]
        orc.forward(prev[j], curr[j], None if variant == "full" else prior[j], btr, n_mc, 0.05, 1, i)

    left = max(1.0, budget_s - (time.perf_counter() - t_leg) - 0.5)
    n, dt = time_budget_rate(one_c, min(left, budget_s * 0.25))
    res["c_port"] = {"value": round(n / dt, 2), "unit": "pairs/s", "cores": c_threads,
                     "sample": f"{n} frame pairs, oracle/liboracle_f32.so with OpenMP on {c_threads} threads, {dt:.1f} s"}
    res["leg_seconds"] = round(time.perf_counter() - t_leg, 1)
    return res


def cpu_baseline_py(state, prev, curr, prior, variant, n_mc, budget_s, avail, btr):
    import torch
    from oracle import torch_cpu
    net = torch_cpu.TorchCpuNet(state)

    def one_t(i):
        j = i % prev.shape[0]
        net.forward(prev[j], curr[j], None if variant == "full" else prior[j], btr, n_mc, 0.05, 1, i)

    # thread count: more threads than the small per-layer loops can feed only add fork/join cost (256 threads ran 60x slower
    # than 8 on the GPU box): try a few, two calls each
    best, cores = None, 1
    for th in [t for t in (8, 16, 32) if t <= avail] or [1]:
        torch.set_num_threads(th)
        one_t(0)
        t0 = time.perf_counter()
        one_t(1)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, th
    torch.set_num_threads(cores)
    n, dt = time_budget_rate(one_t, budget_s * 0.5)
    res = {"value": round(n / dt, 2), "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": f"{n} frame pairs, {variant} model, N={n_mc}, one pair at a time (the reference is batch-1), libtorch {torch.__version__} "
                     f"CPU operators (oracle/torch_cpu.py) on {cores} threads of {avail} host CPUs, {dt:.1f} s",
           "ms_per_pair": round(1e3 * dt / n, 3)}
    # the same on ONE thread (SURVEY.md §8d asks for both; the reference measured 27.7 ms per pair single-threaded)
    torch.set_num_threads(1)
    n1, dt1 = time_budget_rate(one_t, budget_s * 0.1)
    res["single_thread"] = {"value": round(n1 / dt1, 2), "unit": "pairs/s", "cores": 1}
    torch.set_num_threads(cores)
    return res


def committed_traffic(kernel_substr, batch):
    """HBM bytes per launch of the dominant kernel from the newest COMMITTED rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE,
    collected separately with tools/profile_round.sh at batch 256; gfx950 correction of MI355X_MICROARCH.md §HBM, re-calibrated for
    these kernels' access shapes with tools/traffic_calib.hip: bytes = 2 x FETCH_SIZE + WRITE_SIZE).  Not measured in this run:
    returned with its source so that a reader can see which build it belongs to; (None, None) when no matching profile is committed."""
    import csv
    import glob
    if batch != 256:
        return None, None
    import re
    # natural order of the round/version tags (r02_v10 after r02_v9)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.csv")),
                   key=lambda p: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(p))])
    if not files:
        return None, None
    # the figure belongs to ONE build: the CSV carries the digest of the kernel sources that were profiled (tools/csrc_digest.py, written on
    # the GPU box by tools/profile_round.sh).  After any change under csrc/ the committed figure is dropped, not silently re-quoted.
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from csrc_digest import csrc_digest
    with open(files[-1]) as f:
        lines = f.readlines()
    have = [l.split(":", 1)[1].split()[0] for l in lines if l.startswith("# csrc_digest:")]
    if not have or have[0] != csrc_digest(ROOT):
        return None, "none: " + os.path.basename(files[-1]) + " was collected on other kernel sources (csrc digest " + (have[0] if have else "absent") + ")"
    rows = list(csv.DictReader(l for l in lines if not l.startswith("#")))
    for r in rows:
        if kernel_substr in r["kernel"]:
            return ((2.0 * float(r["FETCH_SIZE_KiB_full_batch_launch"]) + float(r["WRITE_SIZE_KiB_full_batch_launch"])) * 1024.0,
                    "profiles/" + os.path.basename(files[-1]))
    return None, None


# stage name -> substring of the HIP kernel name in the rocprof tables
KERNEL_OF_STAGE = {"block_4_0+4_1": "block4_fused_kernel", "heads_fc1": "igemm_heads_pipe_kernel", "block_3_1": "conv_patch_s2_kernel<5",
                   "block_2_2": "igemm_s3_pipe_kernel<hnet::ConvLoaderS3<64, 5, 2, 32>", "block_3_0+3_1": "block3_fused_kernel",
                   "block_4_2+4_3": "block42_fused_kernel", "block_1_2": "igemm_s3_region_kernel<hnet::RegionCfg<128, 5, 1"}
# stages whose contraction does not run on the bf16 matrix cores in the split-bf16 mode: the small FCs (fp32 FMAs).  (Round 1 also ran
# the Cin = 2 first layers of blocks 1 and 2 on the fp32 MFMA; they are bf16x3 kernels since r02_v2, conv_first.h conv7_c2_s2_s3_kernel.)
FP32_STAGES = ("fc_dlt_b1", "fc_dlt_b2", "fc_dlt_b3", "heads_fc2")


def committed_config4_cost():
    """(ms, source) of all-gather + finish on a 1-rank RCCL communicator from the committed C++ measurement (tests/cpp/rccl_gather_example.cpp --time on a gpurun
    box): the line RCCL_CONFIG4_TIME of the newest profiles/r*_rccl_gather_cost.log; (None, None) when absent"""
    import glob
    import re
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rccl_gather_cost.log")), reverse=True):
        for line in open(fn):
            m = re.search(r"RCCL_CONFIG4_TIME.*partial\(4 of 32 samples\)\+allgather\+finish=([0-9.]+)\s+partial\(4 of 32 samples\) alone=([0-9.]+)", line)
            if m:
                return round(float(m.group(1)) - float(m.group(2)), 4), os.path.relpath(fn, ROOT)
    return None, None


def verify_last_step(blob, prev_h, curr_h, prior_h, variant, n_mc, seq_of_slot, mean, cov, slots):
    """the oracle (test infrastructure, CPU) on `slots` of the last step's batch; returns (n, max px err, max cov rel err)"""
    import numpy as np
    from oracle import pyoracle
    btr = {"full": 3, "prior3": 3, "prior2": 2, "prior1": 1}[variant]
    orc = pyoracle.Oracle(blob)
    worst, worst_c = 0.0, 0.0
    for b in slots:
        j = b % prev_h.shape[0]
        o = orc.forward(prev_h[j], curr_h[j], None if variant == "full" else prior_h[j], btr, n_mc, 0.05, 1, seq_of_slot(b))
        worst = max(worst, float(np.abs(mean[b] - o["mean"]).max()))
        worst_c = max(worst_c, float(np.abs(cov[b].reshape(8, 8) - o["cov"]).max() / np.abs(o["cov"]).max()))
    return len(slots), worst, worst_c


def dry_run(args, rank, world):
    """the multi-rank plumbing without a GPU: process group (gloo), the resolved configuration of the presets, the double-buffered gather of the packed
    [B, 72] outputs (cuahn_vio_amd.dist.OverlappedGather on CPU tensors: same order of operations as on the GPU, synchronous), barrier, max-over-ranks
    of a timer, one JSON line"""
    import torch
    import torch.distributed as dist
    if args.pairs_total is not None:
        if args.pairs_total % world:
            raise SystemExit("--pairs-total must be divisible by the number of GPUs")
        args.batch = args.pairs_total // world
    if args.mode == "mc" and args.mc % world:
        raise SystemExit("--mode mc needs N divisible by the number of GPUs")
    if world > 1:
        dist.init_process_group("gloo")
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    dt = time.perf_counter() - t0
    backend, ranks, gather_ok = None, 1, None
    if world > 1:
        from cuahn_vio_amd import dist as hdist
        B = min(args.batch, 64)
        if args.mode == "mc":       # config 4: the per-sample head outputs of this rank's sample range
            n_loc = args.mc // world
            ms = torch.full((B, n_loc, 8), float(rank)) + torch.arange(n_loc).view(1, n_loc, 1)
            ms_all, lv_all, _ = hdist.gather_mc_samples(ms, -ms, None)
            want = torch.cat([torch.full((B, n_loc, 8), float(r)) + torch.arange(n_loc).view(1, n_loc, 1) for r in range(world)], 1)
            gather_ok = bool(torch.equal(ms_all, want) and torch.equal(lv_all, -want))
        else:                       # pairs / stream (config 5): three slabs through the double-buffered gather (small steps are grouped, as in run())
            og = hdist.OverlappedGather(B, "cpu", group_steps=max(1, 128 // args.batch))
            fill = lambda i, r: torch.full((B, 72), float(1000 * i + r)) + torch.arange(B).view(B, 1)     # noqa: E731
            gather_ok = True
            for i in range(3 * og.G):
                og.acquire(i)
                og.buffer(i).copy_(fill(i, rank))
                og.submit(i)
                if i % og.G == og.G - 1:          # the slab has been gathered: every step of it, every rank's rows, in rank order
                    for j in range(i - og.G + 1, i + 1):
                        gather_ok = gather_ok and bool(torch.equal(og.result(j), torch.stack([fill(j, r) for r in range(world)], 0)))
        dist.barrier()
        t = torch.tensor([dt, 0.0 if gather_ok else 1.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, gather_ok = float(t[0]), float(t[1]) == 0.0
        backend, ranks = dist.get_backend(), dist.get_world_size()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": ranks, "backend": backend, "steps": args.steps,
                          "warmup": args.warmup, "max_over_ranks_s": round(dt, 4), "value": None,
                          "resolved": {"config": args.config, "mode": args.mode, "batch_per_gpu": args.batch, "pairs_total": args.pairs_total,
                                       "mc": args.mc, "mc_per_gpu": args.mc // world if args.mode == "mc" else args.mc, "variant": args.variant,
                                       "replay": args.replay, "gathers": world > 1 or args.force_collective,
                                       # what run() would use: steps per all-gather of the packed outputs (small steps are grouped), contexts / HIP streams per GPU
                                       "gather_group_steps": (max(1, 128 // args.batch) if args.mode != "mc" else None),
                                       "contexts": int(args.contexts) if args.contexts else (4 if args.mode == "pairs" else 1)},
                          "gather_checked": gather_ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as `python bench.py --gpus N`: no launcher set the rank environment.  Start the ranks ourselves, as children,
        # before this process initialises the GPU (nothing below this line has run yet: torch is not imported).
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, rank, world)
    if args.pairs_total is not None:
        if args.pairs_total % world:
            raise SystemExit("--pairs-total must be divisible by the number of GPUs")
        args.batch = args.pairs_total // world
    # the HIP runtime multiplexes its streams over GPU_MAX_HW_QUEUES (default 4) hardware queues; with 4, the copy stream of the stream
    # mode regularly shared a queue with the compute stream on the pool's boxes and H2D did not overlap the forward at all
    # (3.6 instead of 3.0 ms per step).  Read by the runtime when HIP initialises, i.e. it has to be set before torch is imported.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")     # the process group's RCCL stream on a high-priority hardware queue: the gather runs UNDER the next forward (dist.py OverlappedGather)
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # HNET_BENCH_SHARED_GPU=1: rehearsal of the N > 1 code path on a box with ONE GPU (all ranks on cuda:0, gloo for the
    # gathers; RCCL refuses two ranks on one device).  The numbers of such a run mean nothing.
    shared = os.environ.get("HNET_BENCH_SHARED_GPU") == "1"
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    collective = world > 1 or args.force_collective
    if collective:
        if world == 1 and "MASTER_ADDR" not in os.environ:      # a 1-rank communicator needs no launcher
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    ctx = {"rank": rank, "local_rank": local_rank, "world": world, "dev": dev, "shared": shared, "collective": collective}
    res, ok = run(args, ctx, primary=True)
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(f"bench.py: the last step's outputs do not match the oracle (max {res.get('max_px_err')} px)")


def child_run(argv, timeout_s=180):
    """one configuration in a CHILD process of this script (never an exec): the streamed configuration overlaps H2D copies with the forward on
    two HIP streams, and which hardware queues those streams get depends on every stream the process created before - after the other
    sub-runs of the default line the copy and the compute stream serialised (2.9 ms per step where the same command alone takes 1.6 - 1.7;
    GPU_MAX_HW_QUEUES=2 restores the overlap in process).  A fresh process is what `python bench.py --mode stream ...` is for a user."""
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.pop("HNET_BENCH_SPAWNED", None)
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout_s)
        line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception as e:                                   # noqa: BLE001 - reported, the caller falls back to the in-process measurement
        return {"error": f"{type(e).__name__}: {e}"}
    keep = ("value", "unit", "ms_per_step", "steps", "verified_pairs", "max_px_err", "max_cov_rel_err")
    out = {k: d[k] for k in keep if k in d}
    out["passed"] = bool(d.get("verify", {}).get("passed", r.returncode == 0)) if isinstance(d.get("verify"), dict) else r.returncode == 0
    out["workload"] = d["config"]["workload"]
    out["precision"] = d["config"]["precision"]
    out["process"] = "child process of the default run (same command as `python bench.py " + " ".join(argv) + "`)"
    return out


def sub_run(base, ctx, **over):
    """one more configuration measured in the same process (the default run's `modes` / `configs` entries): the same step, timing and oracle
    check as the headline, nothing else.  Returns the reduced result dict."""
    import copy
    a = copy.copy(base)
    a.steps, a.warmup = 20, 3
    for k, v in over.items():
        setattr(a, k, v)
    r, ok = run(a, ctx, primary=False)
    keep = ("value", "unit", "ms_per_step", "steps", "verified_pairs", "max_px_err", "max_cov_rel_err", "latency_batch1_ms", "mc_sharding", "stage_ms", "stage_kernels")
    out = {k: r[k] for k in keep if k in r}
    out["contexts"] = r["config"].get("contexts", 1)
    out["passed"] = bool(ok)
    out["workload"] = r["config"]["workload"]
    out["precision"] = r["config"]["precision"]
    return out


def run(args, ctx, primary):
    """builds the engine and the inputs of one configuration, times it by the contract (W warm-up steps, K timed steps between
    barrier + synchronize, max over ranks) and checks the last step against the oracle.  primary = the headline configuration: it also
    carries the roofline, the stage profile, the latency figures, the CPU baseline and (default run) the extra measurements, and prints
    the JSON line."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from cuahn_vio_amd import dist as hdist
    from cuahn_vio_amd import homography_net as _hn
    from cuahn_vio_amd import synth, weights
    from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine

    # the timed engines come from this command line only: HNET_S3_TILE / HNET_FUSE_* / HNET_GRAPH / HNET_WARP_EXACT / HNET_PRECISION of the environment are ignored
    # (and listed in `env_overrides`) unless --honour-env
    honour_env = bool(getattr(args, "honour_env", False))
    _hn.IGNORE_ENV = not honour_env
    rank, local_rank, world, dev, shared, collective = (ctx[k] for k in ("rank", "local_rank", "world", "dev", "shared", "collective"))
    backend = None
    if collective:
        backend = dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")

    B, n_mc = args.batch, args.mc
    prec, mfma_per_mac, peak_tf, dtype_label = PRECISIONS[args.precision]
    blob = weights.pack_state_dict(weights.synthetic_state(0))
    replay = None
    if args.replay:
        from cuahn_vio_amd import replay as hreplay
        replay = hreplay.load_fixture(args.replay)
    # Round 6: every slot of the timed batch is a DISTINCT pair (36.7 MB of images at 256 pairs; rounds 1 - 5 tiled 32 distinct pairs, 4.6 MB, which stayed
    # cache-resident for the four warp + pool launches).  --distinct 32 reproduces the old input; the default line carries that figure once (`tiled_32_distinct`).
    n_distinct = B if not getattr(args, "distinct", None) else max(1, min(B, int(args.distinct)))
    key = (args.replay, rank, n_distinct)
    if key not in _INPUT_CACHE:
        if replay is not None:      # consecutive frames of the trajectory; rank r starts further along it
            _INPUT_CACHE[key] = hreplay.render_pairs(replay, first=rank * n_distinct, count=n_distinct)
        else:
            _INPUT_CACHE[key] = synth.make_batch(1000 + rank * n_distinct, n_distinct)[:3]
    prev_h, curr_h, prior_h = _INPUT_CACHE[key]
    reps = (B + n_distinct - 1) // n_distinct
    prev = torch.from_numpy(np.tile(prev_h, (reps, 1, 1))[:B]).to(dev)
    curr = torch.from_numpy(np.tile(curr_h, (reps, 1, 1))[:B]).to(dev)
    prior = torch.from_numpy(np.tile(prior_h, (reps, 1))[:B]).to(dev)
    d_prior = prior.data_ptr() if args.variant != "full" else None
    # [mean8 | cov64] per pair: written directly by the ensemble kernel (hnet_infer_batch_packed_device); with a collective, two such buffers and the
    # all-gather of step i on a side stream under the forward of step i + 1 (cuahn_vio_amd.dist.OverlappedGather)
    # (small steps are grouped: one collective per >= 128 pairs - at 32 pairs per GPU a step is 0.35 ms and the host-side cost of a Python collective call
    # per step, ~40 us, is visible; a C++ caller's ncclAllGather costs a few us and needs no grouping)
    og = hdist.OverlappedGather(B, dev, group_steps=max(1, 128 // B)) if collective and args.mode != "mc" else None
    out = torch.zeros(B, 72, device=dev)
    out_of_step = (lambda i: og.buffer(i)) if og is not None else (lambda i: out)
    mean, cov = torch.zeros(B, 8, device=dev), torch.zeros(B, 64, device=dev)       # (--mode mc: the un-packed finish entry point)

    mc_mode = args.mode == "mc"
    shard = hdist.shard_range(n_mc, world, rank) if mc_mode else None
    if mc_mode and n_mc % world:
        raise SystemExit("--mode mc needs N divisible by the number of GPUs")
    # --contexts NC: step i runs on context i % NC (own stream, own activation buffers, own output record; the inputs are read-only and shared)
    NC = getattr(args, "contexts", None)
    if NC is None:
        NC = 4 if args.mode == "pairs" else 1
    NC = max(1, int(NC))
    if NC > 1 and (mc_mode or args.mode == "stream"):
        raise SystemExit("--contexts: --mode pairs only")
    group = None
    if NC > 1:
        # round 6: the contexts are an hnet_group (include/hnet.h) - what a C++ caller gets: the library creates the member streams first, in a fixed order,
        # each on its own priority level / hardware queue, and issues the steps round-robin
        from cuahn_vio_amd.homography_net import HnetGroup
        group = HnetGroup(blob, NC, variant=args.variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=1, max_batch=B, device_id=local_rank, precision=prec)
        eng = group.members[0]
    else:
        eng = HnetEngine(blob, variant=args.variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=1, max_batch=B, device_id=local_rank,
                         precision=prec, mc_shard=shard)
    # everything of a step — the forward's kernels, the packing of the outputs and the RCCL gather — is enqueued on ONE
    # ordinary (non-default) stream: ordered by the stream, no legacy-default-stream semantics involved
    stream = torch.cuda.ExternalStream(group.stream(0), device=dev) if group is not None else torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    sp = stream.cuda_stream
    # (round 6: with a collective the steps of one gathered slab may run on different contexts / streams - OverlappedGather.submit waits for every producer
    # stream, acquire makes every stream wait for the slab's previous gather - so grouped small steps (config 5 at 8 GPUs: 32 pairs per GPU, four steps per
    # all-gather) keep their contexts)
    engs, streams, outs = [eng], [stream], [out]
    for k in range(1, NC):
        engs.append(group.members[k])
        streams.append(torch.cuda.ExternalStream(group.stream(k), device=dev))
        outs.append(torch.zeros(B, 72, device=dev))
    if NC > 1 and og is None:
        out_of_step = lambda i: outs[i % NC]
    if mc_mode:
        n_loc = shard[1] - shard[0]
        # one [2, B, n_loc, 8] array per rank (mean_s | logvar_s) and the buffer ONE all-gather fills: the finish reads it in place (round 5: no stack / permute /
        # .contiguous() launches between the forward and the ensemble)
        both = torch.zeros(2, B, n_loc, 8, device=dev)
        ms_loc, lv_loc = both[0], both[1]
        mc_gathered = torch.zeros(world, 2, B, n_loc, 8, device=dev)
        h1 = torch.zeros(B, 9, device=dev)

    stream_mode = args.mode == "stream"
    if stream_mode:
        host_prev = [torch.from_numpy(np.tile(prev_h, (reps, 1, 1))[:B]).pin_memory() for _ in range(2)]
        host_curr = [torch.from_numpy(np.tile(curr_h, (reps, 1, 1))[:B]).pin_memory() for _ in range(2)]
        host_prior = [torch.from_numpy(np.tile(prior_h, (reps, 1))[:B]).pin_memory() for _ in range(2)]
        host_out = [torch.zeros(B, 72).pin_memory() for _ in range(2)]
        dbuf = [(torch.empty_like(prev), torch.empty_like(curr), torch.empty_like(prior)) for _ in range(2)]
        # the copy stream gets its own priority level: the HIP runtime multiplexes streams of one priority over GPU_MAX_HW_QUEUES (4)
        # hardware queues, and a copy stream that lands on the compute stream's queue serialises with it (3.6 ms per step instead
        # of 3.0 on this pool: the copies and the forward did not overlap at all); queues of different priority are never shared
        copy_stream, comp_stream = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev)
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
        # The H2D copies are handed to the DMA engine by a second host thread: hipMemcpyAsync from pinned memory holds its calling thread for
        # most of the transfer on this pool, and with one thread the kernel launches of the next step queued up behind it (2.41 ms per step;
        # 2.13 ms when the upload was at least issued before the launches; measured with HNET_STREAM_THREAD=0 / HNET_STREAM_ORDER=before).
        import queue
        import threading
        use_thread = not (honour_env and os.environ.get("HNET_STREAM_THREAD", "1") == "0")
        up_q = queue.Queue()
        up_done = [threading.Event() for _ in range(2)]
        up_done[0].set()

        def uploader():
            torch.cuda.set_device(dev)
            while True:
                j = up_q.get()
                if j is None:
                    up_q.task_done()
                    return
                upload(j)
                up_done[j % 2].set()
                up_q.task_done()

        if use_thread:
            up_thread = threading.Thread(target=uploader, daemon=True)
            up_thread.start()
            up_q.put(1)

    def seq0_of_step(i):
        return i * B if mc_mode else (rank * 1000003 + i) * B

    def step(i):
        if stream_mode:
            k = i % 2
            skip = os.environ.get("HNET_STREAM_SKIP", "") if honour_env else ""              # experiments: "copy" / "compute"
            early = honour_env and os.environ.get("HNET_STREAM_ORDER", "") == "before"     # experiment: hand the next upload to the DMA engine first
            if use_thread:
                up_done[k].wait()                                       # upload(i) has been enqueued (its event recorded) by the uploader
                up_done[k].clear()
            elif early and skip != "copy":
                upload(i + 1)
            comp_stream.wait_event(ev_ready[k])
            if og is not None:
                og.acquire(i, comp_stream)
            if skip != "compute":
                eng.infer_batch_packed_device(dbuf[k][0].data_ptr(), dbuf[k][1].data_ptr(), PIX_U8, dbuf[k][2].data_ptr() if args.variant != "full" else None,
                                              B, seq0_of_step(i), out_of_step(i).data_ptr(), None, comp_stream)
            ev_free[k].record(comp_stream)
            if og is not None:                                              # BASELINE config 5: the streamed step gathers too, on the side stream
                og.submit(i, comp_stream)
            with torch.cuda.stream(comp_stream):
                host_out[k].copy_(out_of_step(i), non_blocking=True)
            # the next step's pairs cross PCIe while this step computes (enqueued AFTER the forward: the runtime may hold the
            # calling thread until an H2D copy has been handed to the DMA engine, which must not delay the kernel launches)
            if use_thread:
                up_q.put(i + 2)                                         # buffer k is free once this step's forward is done (ev_free[k] above)
            elif skip != "copy" and not early:
                upload(i + 1)
            return
        if mc_mode:   # trunk replicated, heads for this rank's samples, gather, finish in the reference's two-pass order
            eng.infer_mc_partial_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, B, seq0_of_step(i), ms_loc.data_ptr(),
                                        lv_loc.data_ptr(), h1.data_ptr(), sp)
            if collective:
                hdist.gather_mc_block(both, mc_gathered)
                eng.mc_finish_gathered_device(mc_gathered.data_ptr(), world, n_loc, h1.data_ptr(), B, out.data_ptr(), sp)
            else:
                eng.mc_finish_packed_device(ms_loc.data_ptr(), lv_loc.data_ptr(), n_mc, h1.data_ptr(), B, out.data_ptr(), sp)
            return
        if og is not None:
            og.acquire(i, streams[i % NC])
        if group is not None:       # hnet_group_infer_batch_packed_device: member i % NC (the steps are issued in order from 0), on its own stream
            m = group.infer_batch_packed_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, B, seq0_of_step(i), out_of_step(i).data_ptr(), None)
            assert m == i % NC, "step index and group member out of step"
        else:
            eng.infer_batch_packed_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, B, seq0_of_step(i), out_of_step(i).data_ptr(), None, sp)
        if og is not None:
            og.submit(i, streams[i % NC], all_streams=streams)

    def sync():
        if stream_mode and use_thread:
            up_q.join()                                                 # every queued upload has been enqueued on the copy stream
        torch.cuda.synchronize(dev)
        if collective:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    if og is not None and args.warmup:
        og.flush(args.warmup - 1, comp_stream if stream_mode else streams[(args.warmup - 1) % NC], all_streams=None if stream_mode else streams)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if og is not None:      # a step count that is not a multiple of the group: the last, partly filled slab is gathered INSIDE the timed region
        og.flush(args.warmup + args.steps - 1, comp_stream if stream_mode else streams[(args.warmup + args.steps - 1) % NC], all_streams=None if stream_mode else streams)
    sync()
    dt = time.perf_counter() - t0
    if collective:
        t = torch.tensor([dt], device="cpu" if shared else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1e3 * dt / args.steps
    value = (1 if mc_mode else world) * B * args.steps / dt     # mc mode: every rank works on the same B pairs
    last = args.warmup + args.steps - 1                         # index of the last executed step (the one the oracle check looks at)
    extras = primary and not args.no_extras and not stream_mode and not mc_mode
    sustained = None
    if extras and args.sustain_seconds > 0:
        # the timed region above is short (K steps, 0.14 s by default; 34 ms with --steps 20: the chip is still settling its clock).  The same step back to back for >= 1.5 s, outside `value`:
        n_sus = max(args.steps, int(np.ceil(args.sustain_seconds * 1e3 / ms_per_step)))
        t0 = time.perf_counter()
        for i in range(n_sus):
            step(last + 1 + i)
        sync()
        dts = time.perf_counter() - t0
        if collective:
            t = torch.tensor([dts], device="cpu" if shared else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        last += n_sus
        sustained = {"value": round(world * B * n_sus / dts, 1), "unit": "pairs/s", "ms_per_step": round(1e3 * dts / n_sus, 4), "steps": n_sus,
                     "seconds": round(dts, 3), "vs_timed_region": round((world * B * n_sus / dts) / value, 4),
                     "definition": "the same step enqueued back to back right after the timed region, barrier + synchronize on both sides, max over ranks"}

    res = {
        "metric": "homography preds/sec (frame pairs/s), full 4-block HomographyNet @ 320x224",
        "value": round(value, 1), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong" if (mc_mode or args.pairs_total is not None) else "weak", "vs_baseline": None,
        "dtype": dtype_label, "data": "synthetic",
        "config": {"workload": f"{args.variant} HomographyNet forward, 320x224 u8 frame pairs, MC-dropout N={n_mc} p=0.05, "
                               f"{B} pairs/GPU/step ({n_distinct} distinct), inputs+outputs resident in HBM",
                   "batch_per_gpu": B, "mc_samples": n_mc, "variant": args.variant, "precision": args.precision, "contexts": NC,
                   "parallelism": (f"MC-dropout samples sharded {n_mc}/{world} per GPU, trunk replicated, RCCL all_gather of [B,N/R,16]"
                                   if mc_mode else
                                   (f"pairs sharded over {world} GPU(s), RCCL all_gather of [B,72] outputs" if collective else "single GPU")) +
                                  (f"; per GPU the independent steps alternate between the {NC} contexts / HIP streams of an hnet_group" if NC > 1 else ""),
                   "weights": "synthetic seed 0 (trained checkpoint not shipped with the reference)"},
        "mc_preds_per_s": round(value * n_mc, 1),
        "rccl_ranks": dist.get_world_size() if collective else 1, "backend": backend,
    }
    if NC > 1:
        res["config"]["workload"] += f"; independent steps round-robin on {NC} contexts / HIP streams"
    if replay is not None:
        res["config"]["workload"] += f"; frames rendered along UZH-FPV {replay['name']} (tests/golden/replay_{args.replay}.npz), priors from the EKF mean propagation"
        res["config"]["sequence"] = replay["name"]
    if stream_mode:
        res["config"]["workload"] += "; STREAMED: inputs start in pinned host memory, H2D overlapped with compute, outputs back to host"
        res["config"]["parallelism"] = ("host -> device streaming, double-buffered (PCIe-inclusive; not the headline metric); ONE compute context per GPU (the copy stream "
                                        "holds the second hardware queue)" + (f"; RCCL all_gather of the packed outputs every {og.G} step(s) on a side stream" if og is not None else ""))

    # ---- self check, outside the timed region: pairs of the LAST step against the CPU oracle (every rank checks its own shard)
    ok = True
    if sustained is not None:
        res["sustained"] = sustained
    if not args.no_verify:
        slots = sorted({0, 1, B // 2, B - 1} & set(range(B)))
        s0 = seq0_of_step(last)
        if mc_mode:
            mean_np, cov_np = out[:, :8].cpu().numpy(), out[:, 8:].cpu().numpy()
        else:
            fin = out_of_step(last)                                       # the packed [B, 72] record of the last executed step
            mean_np, cov_np = fin[:, :8].cpu().numpy(), fin[:, 8:].cpu().numpy()
            if og is not None:                                            # ... and what the side-stream gather delivered for it: this rank's rows, bit for bit
                og.submit(last, streams[last % NC], flush=True, all_streams=streams)          # (a partly filled group of small steps)
                g = og.result(last)
                if not torch.equal(g[rank], fin):
                    raise SystemExit("bench.py: the gathered outputs differ from the local ones")
        n_v, err_px, err_cov = verify_last_step(blob, prev_h, curr_h, prior_h, args.variant, n_mc, lambda b: s0 + b, mean_np, cov_np, slots)
        gated = args.precision != "bf16"        # plain bf16 is a reported mode: its error is printed, not gated
        gate_px = TOL_PX if args.precision == "f16x2" else TOL_PX_REFERENCE_MODES
        ok = (err_px < gate_px and err_cov < 1e-4) or not gated
        if collective:
            v = torch.tensor([err_px, err_cov, 0.0 if ok else 1.0], device="cpu" if shared else dev, dtype=torch.float64)
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            err_px, err_cov, ok = float(v[0]), float(v[1]), float(v[2]) == 0.0
        res["verified_pairs"] = n_v * world
        res["max_px_err"] = float(f"{err_px:.3e}")
        res["max_cov_rel_err"] = float(f"{err_cov:.3e}")
        res["verify"] = {"against": "oracle/ (CPU restatement, double accumulation)", "slots_per_rank": slots, "step": last,
                         "gate_px": gate_px if gated else None, "passed": bool(ok)}

    if rank == 0 and mc_mode and world == 1 and not primary:
        # BASELINE config 4 stated, not discovered: what sharding the N samples over 8 GPUs saves per rank - the same pair on a context that draws
        # N / 8 of the samples - against what the collective + the finish add (1-rank RCCL, C++ caller: tests/cpp/rccl_gather_example.cpp --time, committed log)
        def t_partial(e, nl, n_it=300):
            bb = torch.zeros(2, B, nl, 8, device=dev)
            for i in range(n_it + 30):
                if i == 30:
                    torch.cuda.synchronize(dev)
                    t0_ = time.perf_counter()
                e.infer_mc_partial_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, B, i * B, bb[0].data_ptr(), bb[1].data_ptr(), h1.data_ptr(), sp)
            torch.cuda.synchronize(dev)
            return 1e3 * (time.perf_counter() - t0_) / n_it
        if n_mc % 8 == 0:
            e8 = HnetEngine(blob, variant=args.variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=1, max_batch=B, device_id=local_rank, precision=prec,
                            mc_shard=(0, n_mc // 8))
            t_all, t_8th = t_partial(eng, n_mc), t_partial(e8, n_mc // 8)
            e8.close()
            coll = committed_config4_cost()
            res["mc_sharding"] = {"forward_all_samples_ms": round(t_all, 4), "forward_one_eighth_of_the_samples_ms": round(t_8th, 4),
                                  "saved_per_rank_ms": round(t_all - t_8th, 4), "allgather_plus_finish_ms": coll[0], "allgather_source": coll[1],
                                  "allgather_measured_in_this_run": False,      # a committed C++ measurement (another box, possibly another build): context for the statement, not a live figure
                                  "statement": ("one pair per step: sharding the N = %d MC samples over 8 GPUs removes %.0f us of heads per rank and adds the collective + finish "
                                                "(%s us on a 1-rank RCCL communicator, more over xGMI): for ONE pair the 8-GPU form is %s than one GPU - the sharding pays "
                                                "for batches of pairs, where the heads are 10 %% of a step, or for N >> 32"
                                                % (n_mc, 1e3 * (t_all - t_8th), "n/a" if coll[0] is None else "%.0f" % (1e3 * coll[0]),
                                                   "n/a" if coll[0] is None else ("slower" if coll[0] > t_all - t_8th else "faster")))}
    if rank == 0 and not primary and getattr(args, "stage_probe", False) and not mc_mode and not stream_mode:
        # per-launch times of THIS configuration (the mid-size batches, where a fixed chain of dependent launches dominates)
        eng.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, B, 0, mean.data_ptr(), cov.data_ptr(), 5)
        ms_st = eng.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, B, 0, mean.data_ptr(), cov.data_ptr(), 20)
        res["stage_ms"] = {n: round(float(m_), 4) for (n, _f), m_ in zip(eng.stages(), ms_st)}
        res["stage_kernels"] = int(sum(eng.stage_kernels()))
    if rank == 0 and not primary and getattr(args, "latency_probe", False) and not mc_mode and not stream_mode:
        # BASELINE config 2 per arithmetic mode ("single frame pair ... bf16"): the batch-1 device latency of THIS configuration (graph replay, 200 after 20)
        e1 = HnetEngine(blob, variant=args.variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=1, max_batch=1, device_id=local_rank, precision=prec)
        e1.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, 1, 0, mean.data_ptr(), cov.data_ptr(), 20)
        per, _tot = e1.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, 1, 0, mean.data_ptr(), cov.data_ptr(), 200)
        res["latency_batch1_ms"] = {"p50": round(float(np.percentile(per, 50)), 4), "p95": round(float(np.percentile(per, 95)), 4)}
        e1.close()
    if rank == 0 and primary and not mc_mode and not stream_mode:
        # ---- roofline of the dominant kernel: per-launch HIP events on the stream the kernels run on.
        # executed FLOP = 2 x MACs x (MFMAs per MAC): three fp16 MFMAs stand behind every MAC of the default mode, six bf16 ones in split-bf16;
        # peak = dense peak of the instruction actually issued.  The fp32-equivalent rate (2 x MACs / time) is a separate field.
        stages = eng.stages()
        # the oracle check above kept the host busy and the GPU idle for seconds: bring the clocks back to the state of the timed
        # region first (one discarded pass), then average 10 launches per stage
        eng.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, B, 0, mean.data_ptr(), cov.data_ptr(), 5)
        ms = [float(x) for x in eng.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, B, 0, mean.data_ptr(),
                                                         cov.data_ptr(), 10)]
        k = int(np.argmax(ms))

        def issued(name, flops):     # (executed flops, peak of the issuing instruction)
            if name in FP32_STAGES or args.precision == "fp32":
                return flops, PEAK_FP32_MFMA_TFLOPS
            return flops * mfma_per_mac, peak_tf

        alg = stages[k][1] * B
        exe, pk = issued(stages[k][0], alg)
        ach = exe / (ms[k] * 1e-3) / 1e12
        # committed PMC pass of the same kernel instantiation: the template arguments name the arithmetic mode (planes = 3 / 2)
        np_arg = {"bf16x3": 3, "f16x2": 2}.get(args.precision)
        ksub = KERNEL_OF_STAGE.get(stages[k][0], "\0")
        if ksub == "block4_fused_kernel" and np_arg:
            ksub = "block4_fused_kernel<%d, 256, %d," % (8 if np_arg == 2 else 7, np_arg)
        traffic, traffic_src = committed_traffic(ksub, B) if np_arg else (None, None)
        ach_alg = alg / (ms[k] * 1e-3) / 1e12
        # achieved = ALGORITHMIC flops per launch / the kernel's duration.  `frac` is that against the dense peak of the instruction issued;
        # `frac_issued` counts every MFMA the fp32-grade operand split issues per multiply-accumulate (3 in the default mode) - the matrix pipe's occupancy
        res["roofline"] = {"bound": "mfma", "achieved": round(ach_alg, 2), "peak": pk, "unit": "TFLOP/s", "frac": round(ach_alg / pk, 4),
                           "achieved_issued": round(ach, 2), "frac_issued": round(ach / pk, 4),
                           "traffic": traffic, "traffic_source": traffic_src,
                           "kernel": stages[k][0], "kernel_ms": round(ms[k], 4),
                           "executed_flops_per_launch": exe, "algorithmic_flops_per_launch": alg,
                           "mfma_per_mac": mfma_per_mac if pk == peak_tf else 1,
                           "fp32_equivalent_tflops": round(alg / (ms[k] * 1e-3) / 1e12, 2),
                           "note": ("achieved / frac count ALGORITHMIC flops (2 x MACs); the default mode issues three fp16 MFMAs per multiply-accumulate "
                                    "(two fp16 planes per value): achieved_issued / frac_issued count those"
                                    if args.precision == "f16x2" else None),
                           "peak_of": ("dense fp16 MFMA (v_mfma_f32_16x16x32_f16; same rate as bf16)" if args.precision == "f16x2" else
                                       "dense bf16 MFMA (v_mfma_f32_16x16x32_bf16)") if pk == PEAK_BF16_MFMA_TFLOPS else "fp32 MFMA (v_mfma_f32_32x32x2_f32)"}
        alg_total = sum(f for _, f in stages) * B
        exe_bf16 = sum(issued(n, f)[0] for n, f in stages if issued(n, f)[1] == PEAK_BF16_MFMA_TFLOPS) * B
        exe_fp32 = sum(issued(n, f)[0] for n, f in stages if issued(n, f)[1] == PEAK_FP32_MFMA_TFLOPS) * B
        t_step = ms_per_step * 1e-3
        res["forward"] = {"gflop_per_pair_algorithmic": round(alg_total / B / 1e9, 4),
                          "fp32_equivalent_tflops": round(alg_total / t_step / 1e12, 2),
                          "executed_tflops_bf16_mfma": round(exe_bf16 / t_step / 1e12, 1),
                          "executed_tflops_fp32_mfma": round(exe_fp32 / t_step / 1e12, 2),
                          # time the step would take with every MFMA issued at its instruction's dense peak / the measured step
                          "frac_of_mfma_peak": round((exe_bf16 / (PEAK_BF16_MFMA_TFLOPS * 1e12) + exe_fp32 / (PEAK_FP32_MFMA_TFLOPS * 1e12)) / t_step, 4),
                          "stage_ms": {n: round(m, 4) for (n, _), m in zip(stages, ms)}}
        if not args.no_latency:
            e1 = HnetEngine(blob, variant=args.variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=1, max_batch=1, device_id=local_rank,
                            precision=prec)
            e1.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, 1, 0, mean.data_ptr(), cov.data_ptr(), 20)
            per, _tot = e1.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, 1, 0, mean.data_ptr(), cov.data_ptr(), 200)
            res["latency_batch1_ms"] = {"p50": round(float(np.percentile(per, 50)), 4), "p95": round(float(np.percentile(per, 95)), 4),
                                        "definition": "device time of one pair, inputs/outputs resident (the reference's 'pure network inference')"}
            # per launch of the latency path (HIP events on the context stream) and the floor SURVEY 8(d) asks for
            e1.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, 1, 0, mean.data_ptr(), cov.data_ptr(), 5)
            l_ms = e1.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, 1, 0, mean.data_ptr(), cov.data_ptr(), 50)
            l_st = e1.stages()
            l_k = e1.stage_kernels()
            rows, fl_total = latency_floor([n for n, _ in l_st], [float(f) for _, f in l_st], [1e3 * float(m) for m in l_ms], n_mc, mfma_per_mac, peak_tf, l_k)
            res["latency_batch1_ms"]["per_launch_us"] = rows
            res["latency_batch1_ms"]["launches"] = len(rows)
            res["latency_batch1_ms"]["kernels"] = int(sum(l_k))
            res["latency_batch1_ms"]["sum_of_launches_us"] = round(float(sum(r["us"] for r in rows)), 1)
            res["latency_batch1_ms"]["floor_us"] = round(fl_total, 1)
            res["latency_batch1_ms"]["floor_definition"] = ("sum over the launches of max(compulsory bytes / 8 TB/s, issued FLOP / dense peak, 1.45 us kernel boundary) + 1.45 us "
                                                            "for every further kernel of a launch (a split-K layer's reduce kernel; `kernels` = hnet_stage_kernels), + 1.0 us per in-launch layer hand-off of a one-XCD tail chain; "
                                                            "per-launch times are event to event (they include the boundary in front of the launch)")
            res["latency_batch1_ms"]["p50_over_floor"] = round(1e3 * res["latency_batch1_ms"]["p50"] / fl_total, 2)
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
        if extras and world == 1 and not args.no_latency:
            res["latency_batch1_ms"]["reference_launch_default"] = reference_launch_default_latency(blob, prev_h, curr_h, prior_h, local_rank, prec)
        if extras and world == 1:
            # the other arithmetic modes and BASELINE's other single-GPU configurations, 20 timed steps each, same process, same oracle gate
            # (a second of idle before each: they follow a 1.5 s sustained window, and the power-limited fp32 MFMA reads 49 k pairs/s on the hot
            # chip where a stand-alone run gives 61 k)
            if NC > 1:      # the figure comparable with rounds 1 - 4: the same steps on ONE context / stream
                res["single_context"] = sub_run(args, ctx, contexts=1, no_extras=True)
            if n_distinct > 32:   # rounds 1 - 5 timed 32 distinct pairs tiled to the batch (images cache-resident for the warp + pool launches): kept beside the headline once
                res["tiled_32_distinct"] = sub_run(args, ctx, distinct=32, no_extras=True)
            res["modes"] = {}
            for pm in ("bf16x3", "fp32", "bf16"):
                if pm != args.precision:
                    time.sleep(1.0)
                    res["modes"][pm] = sub_run(args, ctx, precision=pm, no_extras=True, latency_probe=True)      # + BASELINE config 2 in that mode (batch-1 latency)

            def both_ways(**kw):
                # Mid-size batches leave the chip waiting on their own launch chain (25 dependent launches of 8 - 20 us each): a deployment with INDEPENDENT
                # steps - config 3's batches, config 5's per-GPU share of a streamed sequence - issues them round-robin on a few contexts / HIP streams so that one
                # step's chain runs under the others' kernels.  `value` = that with the FOUR contexts of an hnet_group (round 6: member streams created first, each on its own
                # priority level - monotone 1 -> 4 contexts and within 4 % over three fresh boxes, profiles/r06_ctx_sweep.log); the one-context figure of rounds 1 - 4 stays beside it.
                one = sub_run(args, ctx, contexts=1, stage_probe=True, **kw)
                # (200 steps after 40: a 32 - 64-pair step is 0.2 - 0.35 ms, and the chip needs ~ 30 ms under load to settle its clock - 60 steps read 3 - 5 % low)
                # (a child process, like the streamed configuration: which hardware queues two streams get depends on every stream the process created before;
                # after the other sub-runs the in-process figure was 125 k pairs/s where the same command alone gives 174 - 177 k)
                pip = child_run(["--variant", kw["variant"], "--batch", str(kw["batch"]), "--mc", str(kw["mc"]), "--contexts", "4", "--steps", "200", "--warmup", "40",
                                 "--precision", str(args.precision), "--no-extras", "--no-cpu-baseline", "--no-latency"])
                if "error" in pip:
                    err = pip["error"]
                    pip = sub_run(args, ctx, contexts=4, steps=200, warmup=40, **kw)
                    pip["process"] = "in process (child run failed: " + err + ")"
                pip["contexts"] = 4
                pip["single_context"] = {k: one[k] for k in ("value", "ms_per_step", "steps", "max_px_err", "passed", "stage_ms", "stage_kernels") if k in one}
                pip["passed"] = bool(pip["passed"] and one["passed"])
                return pip

            res["configs"] = {
                "config3_prior3_b64_n16": both_ways(variant="prior3", batch=64, mc=16, no_extras=True),
                "config4_mc_n32_one_pair": sub_run(args, ctx, mode="mc", batch=1, no_extras=True),
                "config5_replay_stream_prior3_b256": None,

                "config5_shape_32_pairs_per_gpu": both_ways(variant="prior3", batch=32, mc=16, no_extras=True),
            }
            # (40 steps after 10: the first H2D copies out of freshly pinned buffers run at a fraction of the link rate)
            stream_argv = ["--mode", "stream", "--replay", "indoor_forward_7", "--variant", "prior3", "--mc", "16", "--batch", str(args.batch),
                           "--steps", "40", "--warmup", "10", "--precision", str(args.precision), "--no-extras", "--no-cpu-baseline", "--no-latency"]
            c5 = child_run(stream_argv)
            if "error" in c5:                                  # the same measurement in process (see child_run for what that costs)
                err = c5["error"]
                c5 = sub_run(args, ctx, mode="stream", replay="indoor_forward_7", variant="prior3", mc=16, no_extras=True, steps=40, warmup=10)
                c5["process"] = "in process (child run failed: " + err + ")"
            res["configs"]["config5_replay_stream_prior3_b256"] = c5
            torch.cuda.set_stream(stream)
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(blob, weights.synthetic_state(0), prev_h, curr_h, prior_h, args.variant, n_mc, args.cpu_seconds)
        par = parity_from_table(ROOT)
        if par is not None:
            res["parity"] = par
    if rank == 0 and primary:
        # The metric is "preds/sec + p50 per-pair latency": the latency figures as TOP-LEVEL scalars, and the line's last kilobyte a compact `summary` that repeats what a
        # reader of a truncated record needs (the driver keeps the END of stdout, and of nested objects only the names).
        lat = res.get("latency_batch1_ms") or {}
        rld = lat.get("reference_launch_default") or {}
        sc = res.get("single_context") or {}
        res["latency_p50_ms"] = lat.get("p50")
        res["latency_p95_ms"] = lat.get("p95")
        res["latency_floor_us"] = lat.get("floor_us")
        res["latency_end_to_end_p50_ms"] = lat.get("end_to_end_p50")
        res["latency_launch_default_p50_ms"] = rld.get("p50")
        res["latency_launch_default_end_to_end_p50_ms"] = rld.get("end_to_end_p50")
        res["single_context_value"] = sc.get("value", res["value"] if NC == 1 else None)
        res["env_overrides"] = {"seen": _hn.env_overrides(), "honoured": honour_env}
        rf = res.get("roofline") or {}
        cfgs = res.get("configs") or {}
        md = res.get("modes") or {}
        res["summary"] = {
            "value_pairs_per_s": res["value"], "contexts": NC, "single_context_value": res["single_context_value"], "ms_per_step": res["ms_per_step"],
            "distinct_pairs_per_step": n_distinct, "tiled_32_distinct_value": (res.get("tiled_32_distinct") or {}).get("value"),
            "latency_p50_ms": res["latency_p50_ms"], "latency_p95_ms": res["latency_p95_ms"], "latency_floor_us": res["latency_floor_us"],
            "latency_end_to_end_p50_ms": res["latency_end_to_end_p50_ms"], "latency_launch_default_p50_ms": res["latency_launch_default_p50_ms"],
            "latency_launches": lat.get("launches"), "latency_kernels": lat.get("kernels"),
            "latency_by_mode_p50_ms": {k: (v.get("latency_batch1_ms") or {}).get("p50") for k, v in md.items()},
            "value_by_mode": {k: v.get("value") for k, v in md.items()},
            "configs_value": {k: (v or {}).get("value") for k, v in cfgs.items()},
            "configs_single_context": {k: ((v or {}).get("single_context") or {}).get("value") for k, v in cfgs.items() if (v or {}).get("single_context")},
            "roofline_kernel": rf.get("kernel"), "roofline_kernel_ms": rf.get("kernel_ms"), "roofline_frac_algorithmic": rf.get("frac"),
            "roofline_frac_issued": rf.get("frac_issued"), "cpu_baseline_pairs_per_s": (res.get("cpu_baseline") or {}).get("value"),
            "max_px_err_vs_oracle": res.get("max_px_err"), "verify_passed": (res.get("verify") or {}).get("passed"),
            "env_overrides": res["env_overrides"]["seen"], "env_honoured": honour_env}
        print(json.dumps(res), flush=True)
    if stream_mode and use_thread:
        up_q.put(None)
        up_thread.join()
    torch.cuda.synchronize(dev)
    torch.cuda.set_stream(torch.cuda.default_stream(dev))      # (the current stream may be a member stream of the group, which dies with it)
    if group is not None:
        group.close()
    else:
        eng.close()
    return res, ok


def parity_from_table(root):
    """the parity statement of the bench line: per comparison the maximum over the golden cases of the newest committed parity table
    (profiles/r*_parity_table.csv, written by tests/test_gpu_parity.py::test_forward_golden under HNET_PARITY_TABLE), default arithmetic.  Says AGAINST WHICH
    evaluation of the reference each figure is: north_star's "< 1e-4 px vs the TorchScript reference" is met against the reference evaluated in double;
    the reference's own fp32 run is up to reference_fp32_vs_fp64_max_px from that."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_parity_table.csv")))
    if not files:
        return None
    rows = [r for r in csv.DictReader(open(files[-1])) if r["precision"] == "f16x2"]
    if not rows:
        return None
    out = {"source": "profiles/" + os.path.basename(files[-1]), "cases": len(rows), "precision": "f16x2",
           "vs_reference_fp64_max_px": max(float(r["abs_err_vs_ref_fp64_px"]) for r in rows),
           "vs_reference_fp32_max_px": max(float(r["abs_err_vs_ref_fp32_px"]) for r in rows),
           "vs_oracle_max_px": max(float(r["abs_err_vs_oracle_px"]) for r in rows),
           "gate_px": {"vs_reference_fp64": 1e-4, "vs_oracle": 1e-4, "vs_reference_fp32": "max(3e-4, the case's own |ref fp32 - ref fp64| + 1e-4)"}}
    try:
        import numpy as np
        worst = 0.0
        for fn in glob.glob(os.path.join(root, "tests", "golden", "*.npz")):
            g = np.load(fn)
            if "mean" in g and "mean64" in g:
                worst = max(worst, float(np.abs(g["mean"] - g["mean64"]).max()))
        out["reference_fp32_vs_fp64_max_px"] = float(f"{worst:.3e}")
    except Exception:                                            # noqa: BLE001 - the goldens are optional for the bench
        pass
    out["statement"] = ("the reference model evaluated in float64 is the yardstick of north_star's 1e-4 px; the reference's own float32 run differs from it by up to "
                        "reference_fp32_vs_fp64_max_px on these cases, so the distance to the float32 goldens is reported, not gated at 1e-4")
    return out


def latency_floor(stage_names, stage_flops, stage_us, n_mc, mfma_per_mac, peak_tf, stage_kernels=None):
    """SURVEY.md section 8(d), batch 1: "report as us, plus % of the sum of per-kernel roofline times".  Per launch of the latency path:
    floor = max(compulsory bytes / 8 TB/s, issued FLOP / the instruction's dense peak, 1.45 us) - 1.45 us is what MI355X_MICROARCH.md prices a dependent
    kernel boundary at.  Bytes = weights (4 B per element in the two-plane fp16 form, fp32 for the small FCs) + the layer's input and output activations
    of ONE pair; the launches are named by hnet_stage_name, so a fused launch carries the bytes of the layers it fuses."""
    from cuahn_vio_amd.weights import CONV_LAYERS
    geo, blk_hw = {}, {1: (28, 40), 2: (56, 80), 3: (112, 160), 4: (224, 320)}
    h = w = 0
    for name, cin, cout, k, s_ in CONV_LAYERS:
        if name.endswith("_1") and name[6] in "12" or name.endswith("_0"):
            h, w = blk_hw[int(name[6])]
        pd = (k - 1) // 2
        ho, wo = (h + 2 * pd - k) // s_ + 1, (w + 2 * pd - k) // s_ + 1
        geo[name] = 4.0 * (cout * cin * k * k + h * w * cin + ho * wo * cout)
        h, w = ho, wo
    rows, total_floor = [], 0.0
    if stage_kernels is None:
        stage_kernels = [1] * len(stage_names)
    for n, fl, us, nk in zip(stage_names, stage_flops, stage_us, stage_kernels):
        by = 0.0
        for part in n.replace("fc_dlt+", "fc_dlt_bX+").replace("prior_dlt+", "").split("+"):
            part = part.strip()
            if part.startswith("block_") and part in geo:
                by += geo[part]
            elif len(part) == 3 and part[0] in "1234" and part[1] == "_":      # "block_4_0+4_1", "block_2_2+2_3+2_4": the further layers of a fused launch / a tail chain
                by += geo.get("block_" + part, 0.0)
            elif part.startswith("prep_b"):
                k = 8 >> (int(part[-1]) - 1)
                by += 2 * 224 * 320 + 4.0 * 2 * (224 // k) * (320 // k)        # two u8 frames in, the pooled two-channel map out
            elif part.startswith("fc_dlt"):
                by += 4.0 * (8 * 5120 + 5120)
            elif part == "heads_fc1":
                by += 4.0 * 512 * 5120 + 4.0 * 5120 + 4.0 * n_mc * 512
            elif part.startswith("heads_fc2"):
                by += 4.0 * (16 * 256 + n_mc * 512)
            elif part == "errmap":
                by += 2 * 224 * 320 + 224 * 320
        mf = 1 if (n.startswith("fc_dlt") and "prep" not in n) or n.startswith("heads_fc2") else mfma_per_mac
        t_b, t_f = by / 8e12 * 1e6, fl * mf / ((peak_tf if mf > 1 else 157.3) * 1e12) * 1e6
        fl_us = max(t_b, t_f, 1.45) + 1.45 * (max(1, int(nk)) - 1)       # every further kernel of the launch is one more dependent boundary
        # a one-XCD tail chain (csrc/chain_lat.h, "block_4_4+4_5+4_6"): its layers are separated by in-launch hand-offs inside one XCD, which the price list of
        # MI355X_MICROARCH.md puts at 0.8 - 1.3 us each (same-XCD flag hand-off): 1.0 us per layer after the first
        if n.startswith("block_") and n.count("+") >= 1 and n.split("+")[0][-1] in "234" and not n.endswith(("4_1", "3_1", "4_3")):
            fl_us += 1.0 * n.count("+")
        total_floor += fl_us
        rows.append({"launch": n, "us": round(float(us), 2), "kernels": int(nk), "floor_us": round(float(fl_us), 2),
                     "bound": "boundary" if max(t_b, t_f) <= 1.45 else ("hbm" if t_b >= t_f else "mfma")})
    return rows, float(total_floor)


def reference_launch_default_latency(blob, prev_h, curr_h, prior_h, device_id, prec):
    """The configuration the reference actually deploys and the only one it recorded a time for (32.8 ms mean network inference on its
    laptop GPU, cuahn_ros/ov_data/uzh_fpv/traj_timing.txt column 4): the 3-block model with the EKF prior, N = 16, the "_showError"
    variant with show_img = true (cuahn/launch/uzhfpv.launch:56-58,67), one pair per call - driven through the HomographyNet class
    surface like VioManager.cpp:188,236-241: load_current_img (71 680 B H2D) + network_inference (forward, photometric error map,
    288 B + 71 680 B D2H).  device = hnet_last_timing().device_ms of each call (events around the forward, the error map and the
    copies back); end_to_end = host wall clock per frame."""
    import contextlib
    import io
    import numpy as np
    from cuahn_vio_amd.homography_net import HomographyNet
    with contextlib.redirect_stdout(io.StringIO()):
        net = HomographyNet("bench_showError.hnw", use_prior=True, blocks_to_run=3, mc_samples=16, dropout_p=0.05, mc_seed=1,
                            device_id=device_id, weights_blob=blob, precision=prec)
        dev_ms, e2e = [], []
        n = prev_h.shape[0]
        for i in range(220):
            t0 = time.perf_counter()
            net.load_current_img(curr_h[i % n], float(i))
            net.network_inference(prior_h[i % n].astype(np.float64), 0)
            if i >= 20:
                e2e.append(1e3 * (time.perf_counter() - t0))
                dev_ms.append(net._eng.last_timing()["device_ms"])
        assert net.last_error_map is not None and net.last_error_map.shape == (224, 320)
    pc = lambda v, q: round(float(np.percentile(v, q)), 4)
    return {"p50": pc(dev_ms, 50), "p95": pc(dev_ms, 95), "end_to_end_p50": pc(e2e, 50), "end_to_end_p95": pc(e2e, 95),
            "config": "prior-3 (3 blocks + EKF prior), N=16, p=0.05, error map computed and copied back (71 680 B), batch 1, 200 frames after 20",
            "reference_recorded_ms": 32.8,
            "reference_recorded_note": "mean of column 4 of the reference's own traj_timing.txt (its laptop GPU, libtorch CUDA); other hardware - context only"}


if __name__ == "__main__":
    main()
