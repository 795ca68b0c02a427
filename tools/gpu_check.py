#!/usr/bin/env python3
"""Quick GPU-side diagnostics (no asserts): per-operator and end-to-end error of the HIP path vs the oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cuahn_vio_amd import synth, weights  # noqa: E402
from cuahn_vio_amd.homography_net import HnetEngine  # noqa: E402
from cuahn_vio_amd.weights import CONV_LAYERS  # noqa: E402
from oracle import pyoracle  # noqa: E402


def main():
    state = weights.synthetic_state(0)
    blob = weights.pack_state_dict(state)
    t = time.time()
    eng = HnetEngine(blob, variant="full", mc_samples=16, dropout_p=0.05, mc_seed=5, max_batch=4, emit_error_map=True)
    print(f"create: {time.time() - t:.2f}s  warm-up {eng.last_timing()['host_ms']:.1f} ms")
    orc = pyoracle.Oracle(blob)
    i1, i2, off = synth.make_pair(1)
    f1, f2 = pyoracle.as_f32_image(i1), pyoracle.as_f32_image(i2)
    p4 = np.array([0, 0, 0, 223, 319, 223, 319, 0], np.float32)
    h = pyoracle.dlt(p4 + off.astype(np.float32))
    print("dlt   :", np.abs(eng.op_dlt(p4 + off.astype(np.float32))[0] - h).max())
    print("warp  :", np.abs(eng.op_warp(f2, h) - pyoracle.warp(f2, h)).max())
    for k in (1, 2, 4, 8):
        ref = pyoracle.avgpool(np.stack([f1, pyoracle.warp(f2, h)]), k)
        print(f"prep k={k}:", np.abs(eng.op_prep(f1, f2, h, k) - ref).max())
    rng = np.random.default_rng(0)
    dims = {1: (28, 40), 2: (56, 80), 3: (112, 160), 4: (224, 320)}
    cur = {}
    for li, (name, cin, cout, k, s) in enumerate(CONV_LAYERS):
        blk = int(name[6])
        hh, ww = cur.get(blk, dims[blk])
        x = rng.standard_normal((1, cin, hh, ww)).astype(np.float32)
        pre = "model_last_block_list.0." if blk == 4 else "model_part1."
        ref = pyoracle.conv_lrelu(x[0], state[pre + name + ".0.weight"], state[pre + name + ".0.bias"], s)
        got = eng.op_conv(li, x)[0]
        print(f"conv {name}: max err {np.abs(got - ref).max():.2e} (ref max {np.abs(ref).max():.2f})")
        cur[blk] = ref.shape[1:]
    mean, cov, err = eng.infer_batch(i1[None], i2[None], pair_seq0=3, want_err=True)
    o = orc.forward(i1, i2, n_mc=16, p=0.05, mc_seed=5, pair_seq=3, want_err=True, want_trace=True)
    print("forward mean hip   :", mean[0])
    print("forward mean oracle:", o["mean"])
    print("max |d mean| px:", np.abs(mean[0] - o["mean"]).max(), " cov rel:", np.abs(cov[0] - o["cov"]).max() / np.abs(o["cov"]).max())
    print("H_part1 diff:", np.abs(eng.debug_h_part1(0) - o["H_part1"]).max(), " err map max diff:", np.abs(err[0] - o["err"]).max())
    for li, (name, *_r) in enumerate(CONV_LAYERS):
        a = eng.debug_layer_output(li, 0).astype(np.float64).reshape(-1)
        st = o["layer_stats"][name]
        print(f"  {name}: L2 rel diff {abs(np.sqrt((a * a).sum()) - st[1]) / st[1]:.2e}")


if __name__ == "__main__":
    main()
