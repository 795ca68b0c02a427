#!/usr/bin/env python3
"""A/B of the block-tail FC on the latency path: summed from the 32 partial sums the tail chain's last layer leaves (default, round 6) against recomputed by every
workgroup of the next warp + pool launch (HNET_CHAIN_FC=0 = include/hnet.h HNET_VARIANT_CHAIN_NO_FC).  Device p50 of back-to-back forwards, interleaved repetitions,
and the distance between the two forms' outputs.      python tools/chain_fc_ab.py [reps=3] [ENV_SWITCH=HNET_CHAIN_FC]
(any other 0 / 1 switch of cuahn_vio_amd/homography_net.py kernel_selection_from_env can be given as the second argument, e.g. HNET_POOL_FUSE)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cuahn_vio_amd import synth, weights  # noqa: E402
from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
SW = sys.argv[2] if len(sys.argv) > 2 else "HNET_CHAIN_FC"
dev = torch.device("cuda:0")
blob = weights.pack_state_dict(weights.synthetic_state(0))
for variant, n_mc, batch in (("full", 32, 1), ("prior3", 16, 1), ("full", 32, 8)):
    ph, ch, prh, _ = synth.make_batch(40, batch)
    prev, curr, prior = torch.from_numpy(ph).to(dev), torch.from_numpy(ch).to(dev), torch.from_numpy(prh).to(dev)
    mean, cov = torch.zeros(batch, 8, device=dev), torch.zeros(batch, 64, device=dev)
    dp = prior.data_ptr() if variant != "full" else None
    outs = {}
    for r in range(reps):
        for fc in ("1", "0"):
            os.environ[SW] = fc
            e = HnetEngine(blob, variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=3, max_batch=batch)
            e.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, dp, batch, 0, mean.data_ptr(), cov.data_ptr(), 30)
            per, _ = e.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, dp, batch, 0, mean.data_ptr(), cov.data_ptr(), 300)
            outs[fc] = mean.cpu().numpy().copy()
            n_launch = len(e.stages())
            e.close()
            print(f"{variant} N={n_mc} batch={batch} rep {r}: {SW}={fc}{(' (FC from the chain partials)' if fc == '1' else ' (FC recomputed in the prep launch)') if SW == 'HNET_CHAIN_FC' else ''}: p50 {np.percentile(per, 50) * 1e3:7.2f} us  p95 {np.percentile(per, 95) * 1e3:7.2f} us  ({n_launch} launches)", flush=True)
    print(f"   max |offset({SW}=1) - offset({SW}=0)| = {np.abs(outs['1'] - outs['0']).max():.2e} px")
os.environ.pop(SW, None)
