#!/usr/bin/env python3
"""diagnostics for the HNET_STREAMS=2 path: repeat a 48-pair batch, report which pairs / how much results move between runs,
under a few kernel switches (bisection of a suspected race)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child():
    import numpy as np
    from cuahn_vio_amd import synth, weights
    from cuahn_vio_amd.homography_net import HnetEngine
    blob = weights.pack_state_dict(weights.synthetic_state(0))
    prev, curr, prior, _ = synth.make_batch(4500, 8)
    prev, curr, prior = np.tile(prev, (6, 1, 1)), np.tile(curr, (6, 1, 1)), np.tile(prior, (6, 1))
    prec = int(os.environ.get("DBG_PREC", "2"))
    variant = os.environ.get("DBG_VARIANT", "prior3")
    eng = HnetEngine(blob, variant=variant, mc_samples=16, dropout_p=float(os.environ.get("DBG_P", "0.05")), mc_seed=5, max_batch=48, precision=prec)
    runs = [eng.infer_batch(prev, curr, prior if variant != "full" else None, pair_seq0=11) for _ in range(6)]
    m0 = runs[0][0]
    out = []
    for m, _c in runs[1:]:
        d = np.abs(m - m0).max(axis=1)
        out.append({"n_diff_pairs": int((d > 0).sum()), "max": float(d.max()), "pairs": np.nonzero(d > 0)[0][:12].tolist()})
    print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith(("HNET_", "DBG_"))}, "runs": out}))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        base = {"HNET_STREAMS": "2"}
        for extra in ({}, {"HNET_STREAMS": "1"}, {"DBG_PREC": "0"}, {"HNET_FUSE_B4": "0"}, {"HNET_PATCH": "0"}, {"HNET_S3_DMA": "0"},
                      {"DBG_P": "0.0"}, {"DBG_VARIANT": "prior1"}, {"HNET_PREP_TILED": "0"}, {"HNET_XCD_REMAP": "0"}):
            env = dict(os.environ); env.update(base); env.update(extra)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
            print(p.stdout.strip() or p.stderr[-400:], flush=True)
