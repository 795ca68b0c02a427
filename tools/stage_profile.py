#!/usr/bin/env python3
"""per-stage device time at a given batch (HIP events after every launch): python tools/stage_profile.py [batch] [variant]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuahn_vio_amd import synth, weights
from cuahn_vio_amd.homography_net import HnetEngine, PIX_U8
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variant = sys.argv[2] if len(sys.argv) > 2 else "full"
blob = weights.pack_state_dict(weights.synthetic_state(0))
prev_h, curr_h, prior_h, _ = synth.make_batch(1, min(B, 8))
reps = (B + 7) // 8
dev = torch.device("cuda:0")
prev = torch.from_numpy(np.tile(prev_h, (reps, 1, 1))[:B]).to(dev); curr = torch.from_numpy(np.tile(curr_h, (reps, 1, 1))[:B]).to(dev)
prior = torch.from_numpy(np.tile(prior_h, (reps, 1))[:B]).to(dev)
mean = torch.zeros(B, 8, device=dev); cov = torch.zeros(B, 64, device=dev)
eng = HnetEngine(blob, variant=variant, mc_samples=32, dropout_p=0.05, mc_seed=1, max_batch=B)
dp = prior.data_ptr() if variant != "full" else None
eng.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, dp, B, 0, mean.data_ptr(), cov.data_ptr(), 3)
ms = eng.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, dp, B, 0, mean.data_ptr(), cov.data_ptr(), 20)
per, tot = eng.time_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, dp, B, 0, mean.data_ptr(), cov.data_ptr(), 50)
print(f"batch {B} {variant}: sum of stages {ms.sum()*1e3:.1f} us, back-to-back p50 {np.percentile(per,50)*1e3:.1f} us")
for (n, f), m in zip(eng.stages(), ms):
    print(f"  {n:18s} {m*1e3:8.1f} us   {f*B/(m*1e-3)/1e12 if m>0 else 0:6.2f} TF/s")
