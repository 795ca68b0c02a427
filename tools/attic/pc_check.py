"""large-batch run of the forward; saves mean/cov so that two builds / env settings can be compared (tools/exp.sh)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (before the HIP library, see tests/conftest.py)
from cuahn_vio_amd import synth, weights
from cuahn_vio_amd.homography_net import HnetEngine
B = int(sys.argv[2]) if len(sys.argv) > 2 else 130
blob = weights.pack_state_dict(weights.synthetic_state(0))
prev, curr, prior, _ = synth.make_batch(500, 16)
reps = (B + 15) // 16
prev = np.tile(prev, (reps, 1, 1))[:B]; curr = np.tile(curr, (reps, 1, 1))[:B]
eng = HnetEngine(blob, variant="full", mc_samples=32, dropout_p=0.05, mc_seed=3, max_batch=B, precision=2)
mean, cov = eng.infer_batch(prev, curr, None, pair_seq0=7)[:2]
np.savez(sys.argv[1], mean=mean, cov=cov)
print("saved", sys.argv[1], mean.shape, float(np.abs(mean).max()))
