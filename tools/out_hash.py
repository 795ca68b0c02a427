#!/usr/bin/env python3
"""sha256 of the outputs (mean | cov | H_part1 of slot 0 | error map) of fixed forwards - before / after a change that must not alter a single bit
   python tools/out_hash.py [precision=3]"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cuahn_vio_amd import synth, weights
from cuahn_vio_amd.homography_net import HnetEngine
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
blob = weights.pack_state_dict(weights.synthetic_state(0))
for variant, n_mc, batch in (("full", 32, 256), ("prior3", 16, 64), ("full", 16, 1), ("prior1", 8, 5)):
    prev, curr, prior, _ = synth.make_batch(300 + batch, min(batch, 32))
    reps = (batch + 31) // 32
    prev, curr, prior = (np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:batch].copy() for a in (prev, curr, prior))
    e = HnetEngine(blob, variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=9, max_batch=batch, emit_error_map=True, precision=prec)
    m, c, err = e.infer_batch(prev, curr, None if variant == "full" else prior, pair_seq0=17, want_err=True)
    h = hashlib.sha256(m.tobytes() + c.tobytes() + e.debug_h_part1(0).tobytes() + err.tobytes()).hexdigest()[:16]
    print(f"out_hash precision={prec} {variant} N={n_mc} batch={batch}: {h}")
    e.close()
