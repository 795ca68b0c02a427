#!/usr/bin/env python3
"""In-process A/B of build-time-identical kernels under different context switches (cdna_hip_programming.md §5.4 rule 24:
perf deltas come from interleaved rounds in ONE process on ONE device).  Each variant is a dict of the environment variables that the Python
mirror maps onto hnet_config (cuahn_vio_amd/homography_net.py kernel_selection_from_env: HNET_S3_TILE = the low byte of hnet_config.variant,
HNET_FUSE_SMALL / HNET_FUSE_B3 / HNET_FUSE_B42 = 0, HNET_GRAPH, HNET_WARP_EXACT; the library itself reads no environment variable); variants are
timed round-robin with per-stage HIP events.

  python tools/ab_bench.py '{}' '{"HNET_S3_TILE":"30"}' [--rounds 5] [--batch 256] [--stages block_4_0+4_1,block_3_1]
"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--mc", type=int, default=32)
    ap.add_argument("--variant", default="full")
    ap.add_argument("--precision", type=int, default=3)
    ap.add_argument("--stages", default="")
    a = ap.parse_args()
    import torch
    from cuahn_vio_amd import synth, weights
    from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine
    dev = torch.device("cuda:0")
    blob = weights.pack_state_dict(weights.synthetic_state(0))
    B = a.batch
    ph, ch, prh, _ = synth.make_batch(1000, min(B, 32))
    reps = (B + ph.shape[0] - 1) // ph.shape[0]
    prev = torch.from_numpy(np.tile(ph, (reps, 1, 1))[:B]).to(dev)
    curr = torch.from_numpy(np.tile(ch, (reps, 1, 1))[:B]).to(dev)
    prior = torch.from_numpy(np.tile(prh, (reps, 1))[:B]).to(dev)
    mean, cov = torch.zeros(B, 8, device=dev), torch.zeros(B, 64, device=dev)
    d_prior = prior.data_ptr() if a.variant != "full" else None
    engs = []
    for v in a.variants:
        env = json.loads(v)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        # (AB_DROPOUT_P inside a variant's JSON: the MC-dropout probability of that engine - data-dependence of the heads GEMM, not a library switch)
        engs.append(HnetEngine(blob, variant=a.variant, mc_samples=a.mc, dropout_p=float(env.get("AB_DROPOUT_P", 0.05)), mc_seed=1, max_batch=B, precision=a.precision))
        for k, o in old.items():
            if o is None: del os.environ[k]
            else: os.environ[k] = o
    want = [s for s in a.stages.split(",") if s]
    acc = [[] for _ in engs]
    tot = [[] for _ in engs]
    for r in range(a.rounds + 1):
        for i, e in enumerate(engs):
            ms = e.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, B, 0, mean.data_ptr(), cov.data_ptr(), 3)
            _per, t = e.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, B, 0, mean.data_ptr(), cov.data_ptr(), 5)
            if r:           # round 0 = warm-up
                acc[i].append(ms)
                tot[i].append(t / 5)
    names = [n for n, _ in engs[0].stages()]
    for i, v in enumerate(a.variants):
        m = np.median(np.array(acc[i]), axis=0)
        mn = np.min(np.array(acc[i]), axis=0)
        nm = [n for n, _ in engs[i].stages()]
        sel = {n: (round(float(x), 4), round(float(y), 4)) for n, x, y in zip(nm, m, mn) if not want or n in want}
        print(json.dumps({"variant": json.loads(v), "forward_ms_median": round(float(np.median(tot[i])), 4), "forward_ms_min": round(float(np.min(tot[i])), 4),
                          "stage_ms_median_min": sel}), flush=True)
    for e in engs: e.close()

if __name__ == "__main__":
    main()
