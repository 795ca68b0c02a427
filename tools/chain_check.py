#!/usr/bin/env python3
"""The one-XCD tail chains (csrc/chain_lat.h) against the per-layer launches (HNET_CHAIN=0): outputs, every layer's map, timing.
   python tools/chain_check.py [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cuahn_vio_amd import synth, weights
from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine
from oracle import pyoracle

dev = torch.device("cuda:0")
blob = weights.pack_state_dict(weights.synthetic_state(0))
orc = pyoracle.Oracle(blob)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def eng(chain, **kw):
    os.environ["HNET_CHAIN"] = "1" if chain else "0"
    try:
        return HnetEngine(blob, **kw)
    finally:
        os.environ.pop("HNET_CHAIN", None)


bad = 0
for variant, n_mc, batch in (("full", 32, 1), ("prior3", 16, 1), ("full", 16, 2), ("prior2", 8, 5), ("full", 16, 8), ("prior1", 8, 3)):
    ph, ch, prh, _ = synth.make_batch(300 + batch, batch)
    pr = None if variant == "full" else prh
    res = {}
    for chain in (True, False):
        e = eng(chain, variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=3, max_batch=8)
        mean, cov = e.infer_batch(ph, ch, pr, pair_seq0=5)
        mean2, cov2 = e.infer_batch(ph, ch, pr, pair_seq0=5)
        assert np.array_equal(mean, mean2) and np.array_equal(cov, cov2), "not reproducible"
        h1 = np.stack([e.debug_h_part1(b) for b in range(batch)])
        layers = {}
        btr = {"full": range(20), "prior3": range(3, 20), "prior2": range(7, 20), "prior1": range(13, 20)}[variant]
        for l in btr:
            if l == 13:
                continue
            try:
                layers[l] = np.stack([e.debug_layer_output(l, b) for b in range(batch)])
            except Exception as ex:      # noqa: BLE001
                layers[l] = None
        names = [n for n, _ in e.stages()]
        res[chain] = (mean, cov, h1, layers, names, e.overflow_flag())
        e.close()
    (m1, c1, h1, l1, n1, f1), (m0, c0, h0, l0, n0, f0) = res[True], res[False]
    worst_layer = 0.0
    for l in l1:
        if l1[l] is not None and l0[l] is not None:
            d = np.abs(l1[l] - l0[l]).max() / max(np.abs(l0[l]).max(), 1e-30)
            worst_layer = max(worst_layer, d)
            if d > 1e-5:
                print(f"   layer {l}: rel diff {d:.3e}")
    btrn = {"full": 3, "prior3": 3, "prior2": 2, "prior1": 1}[variant]
    o = orc.forward(ph[0], ch[0], None if pr is None else pr[0], btrn, n_mc, 0.05, 3, 5)
    d_paths = np.abs(m1 - m0).max()
    d_or1, d_or0 = np.abs(m1[0] - o["mean"]).max(), np.abs(m0[0] - o["mean"]).max()
    print(f"{variant} N={n_mc} batch={batch}: |chain - launches| = {d_paths:.3e} px, H rel {np.abs(h1 - h0).max() / np.abs(h0).max():.2e}, worst layer rel {worst_layer:.2e}; "
          f"vs oracle: chain {d_or1:.3e}, launches {d_or0:.3e}; flags {f1} {f0}; launches {len(n1)} vs {len(n0)}", flush=True)
    if d_paths > 5e-5 or d_or1 > 1e-4 or f1:
        bad += 1
        print("   stages:", n1)

# timing: batch-1 device latency (graph-free, back to back) and per-stage
for variant, n_mc in (("full", 32), ("prior3", 16)):
    ph, ch, prh, _ = synth.make_batch(40, 8)
    prev, curr, prior = torch.from_numpy(ph).to(dev), torch.from_numpy(ch).to(dev), torch.from_numpy(prh).to(dev)
    mean, cov = torch.zeros(8, 8, device=dev), torch.zeros(8, 64, device=dev)
    dp = prior.data_ptr() if variant != "full" else None
    for batch in (1, 8):
        row = []
        for chain in (True, False):
            e = eng(chain, variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=3, max_batch=8)
            e.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, dp, batch, 0, mean.data_ptr(), cov.data_ptr(), 30)
            per, _ = e.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, dp, batch, 0, mean.data_ptr(), cov.data_ptr(), reps)
            e.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, dp, batch, 0, mean.data_ptr(), cov.data_ptr(), 5)
            st = e.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, dp, batch, 0, mean.data_ptr(), cov.data_ptr(), 50)
            names = [n for n, _ in e.stages()]
            row.append((float(np.percentile(per, 50)), float(np.percentile(per, 95)), dict(zip(names, [round(1e3 * float(x), 2) for x in st]))))
            e.close()
        print(f"latency {variant} N={n_mc} batch={batch}: chain p50 {row[0][0]*1e3:.1f} us (p95 {row[0][1]*1e3:.1f}), launches p50 {row[1][0]*1e3:.1f} us (p95 {row[1][1]*1e3:.1f})")
        print("   chain stages us:", row[0][2])
        if batch == 1:
            print("   launch stages us:", row[1][2])
sys.exit(1 if bad else 0)
