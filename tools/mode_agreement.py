#!/usr/bin/env python3
"""How far apart are the three fp32-grade arithmetic modes on many pairs?  Every pair of a few large synthetic batches (full model and
prior-3, MC-dropout on) is run through HNET_PREC_F16X2, HNET_PREC_BF16X3 and HNET_PREC_FP32; the largest corner-offset difference between
the modes is printed per configuration (the oracle checks of tests/ and bench.py cover a handful of pairs; this covers thousands).
  python tools/mode_agreement.py [--batches 8]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    from cuahn_vio_amd import synth, weights
    from cuahn_vio_amd.homography_net import HnetEngine
    blob = weights.pack_state_dict(weights.synthetic_state(0))
    for variant, n_mc in (("full", 32), ("prior3", 16)):
        engs = {p: HnetEngine(blob, variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=11, max_batch=a.batch, precision=p) for p in (3, 2, 0)}
        worst = {(3, 2): 0.0, (3, 0): 0.0, (2, 0): 0.0}
        worst_c = dict(worst)
        scale = 0.0
        for b in range(a.batches):
            prev, curr, prior, _ = synth.make_batch(5000 + 17 * b, a.batch)
            out = {p: e.infer_batch(prev, curr, None if variant == "full" else prior, pair_seq0=b * a.batch) for p, e in engs.items()}
            scale = max(scale, float(np.abs(out[0][0]).max()))
            for (x, y) in worst:
                worst[(x, y)] = max(worst[(x, y)], float(np.abs(out[x][0] - out[y][0]).max()))
                worst_c[(x, y)] = max(worst_c[(x, y)], float((np.abs(out[x][1] - out[y][1]).reshape(a.batch, -1).max(1) / np.abs(out[y][1]).reshape(a.batch, -1).max(1)).max()))
        assert all(e.precision() == p for p, e in engs.items())
        for e in engs.values():
            e.close()
        print(f"{variant} N={n_mc}: {a.batches * a.batch} pairs, offsets up to {scale:.1f} px; max |offset difference| px: "
              f"f16x2-bf16x3 {worst[(3, 2)]:.2e}, f16x2-fp32 {worst[(3, 0)]:.2e}, bf16x3-fp32 {worst[(2, 0)]:.2e}; "
              f"max relative covariance difference: {worst_c[(3, 2)]:.1e}, {worst_c[(3, 0)]:.1e}, {worst_c[(2, 0)]:.1e}")


if __name__ == "__main__":
    main()
