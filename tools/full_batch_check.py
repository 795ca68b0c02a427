#!/usr/bin/env python3
"""Every pair of one benchmark-shaped batch against the CPU oracle (test infrastructure: oracle/ is the checker, never the product):
    python tools/full_batch_check.py [batch=256] [n_mc=32] [variant=full] [n_distinct=256] [precision=3]
bench.py and tests/test_gpu_bench_shapes.py check a handful of slots per run; this walks ALL slots of a batch of DISTINCT synthetic pairs
(different textures and homographies, corner offsets up to 12 px) once, prints the error distribution and the worst slot."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (before the C ABI: see tests/conftest.py)
from cuahn_vio_amd import synth, weights
from cuahn_vio_amd.homography_net import HnetEngine
from oracle.pyoracle import Oracle

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
variant = sys.argv[3] if len(sys.argv) > 3 else "full"
nd = int(sys.argv[4]) if len(sys.argv) > 4 else B
prec = int(sys.argv[5]) if len(sys.argv) > 5 else 3              # 3 = two fp16 planes (default), 2 = bf16x3, 0 = fp32 MFMA
seed = 0x5EED5EED12345678
state = weights.synthetic_state(0)
blob = weights.pack_state_dict(state)
prev, curr, prior, _ = synth.make_batch(70000, nd)
reps = (B + nd - 1) // nd
prev, curr, prior = np.tile(prev, (reps, 1, 1))[:B].copy(), np.tile(curr, (reps, 1, 1))[:B].copy(), np.tile(prior, (reps, 1))[:B].copy()
if variant == "full":
    prior = None
eng = HnetEngine(blob, variant=variant, mc_samples=N, dropout_p=0.05, mc_seed=seed, max_batch=B, precision=prec)
mean, cov = eng.infer_batch(prev, curr, prior, pair_seq0=4242)
orc = Oracle(blob)
t0 = time.time()
err = np.zeros(B); cre = np.zeros(B)
for b in range(B):
    o = orc.forward(prev[b], curr[b], None if prior is None else prior[b], 3, N, 0.05, seed, 4242 + b)
    err[b] = np.abs(mean[b] - o["mean"]).max()
    cre[b] = np.abs(cov[b] - o["cov"]).max() / np.abs(o["cov"]).max()
print(f"{variant} B={B} N={N}, {nd} distinct pairs, precision {prec}, HNET_WARP_EXACT={os.environ.get('HNET_WARP_EXACT', '0')}, oracle time {time.time() - t0:.1f} s")
gate = 1e-4 if prec == 3 else 1.5e-4        # tests/conftest.py tol_px_vs_oracle: the default mode at north_star's figure, the reference modes at their own fp32 noise + margin
print(f"|offset - oracle| px: max {err.max():.3e} (slot {int(err.argmax())}), p99 {np.percentile(err, 99):.3e}, median {np.median(err):.3e}; gate {gate:.1e}")
print(f"cov rel err: max {cre.max():.3e}, median {np.median(cre):.3e}; gate 2e-5")
print(f"max |offset| in the batch: {np.abs(mean).max():.2f} px; stages: {len(eng.stages())}")
sys.exit(0 if err.max() < gate and cre.max() < 2e-5 else 1)
