#!/usr/bin/env python3
"""batch-1 forwards back to back on one context (for rocprofv3 --kernel-trace: per-kernel durations and the gaps between them on the latency path).
   python tools/lat_trace.py [variant] [n_mc] [reps]      HNET_CHAIN=0: the per-layer launches"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cuahn_vio_amd import synth, weights
from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine
variant = sys.argv[1] if len(sys.argv) > 1 else "full"
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda:0")
blob = weights.pack_state_dict(weights.synthetic_state(0))
ph, ch, prh, _ = synth.make_batch(40, 1)
prev, curr, prior = torch.from_numpy(ph).to(dev), torch.from_numpy(ch).to(dev), torch.from_numpy(prh).to(dev)
mean, cov = torch.zeros(1, 8, device=dev), torch.zeros(1, 64, device=dev)
e = HnetEngine(blob, variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=3, max_batch=1)
dp = prior.data_ptr() if variant != "full" else None
per, _ = e.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, dp, 1, 0, mean.data_ptr(), cov.data_ptr(), 30)
per, _ = e.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, dp, 1, 0, mean.data_ptr(), cov.data_ptr(), reps)
print(f"{variant} N={n_mc}: p50 {np.percentile(per, 50) * 1e3:.1f} us over {reps} forwards, HNET_CHAIN={os.environ.get('HNET_CHAIN', '1')}")
e.close()
