#!/usr/bin/env python3
"""scratch experiment: does running two half batches on two HIP streams (two independent contexts, no shared buffers) beat one
full batch on one stream?  GPU_MAX_HW_QUEUES must give the two streams different hardware queues.
   python tools/dual_ctx_bench.py [total_batch=256] [n_ctx=2]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cuahn_vio_amd import synth, weights
from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
blob = weights.pack_state_dict(weights.synthetic_state(0))
ph, ch, prh, _ = synth.make_batch(1000, 32)
def mk(b):
    reps = (b + 31) // 32
    return (torch.from_numpy(np.tile(ph, (reps, 1, 1))[:b]).to(dev), torch.from_numpy(np.tile(ch, (reps, 1, 1))[:b]).to(dev),
            torch.zeros(b, 8, device=dev), torch.zeros(b, 64, device=dev))
def run(nc, steps=30):
    b = B // nc
    engs = [HnetEngine(blob, variant="full", mc_samples=32, dropout_p=0.05, mc_seed=1, max_batch=b) for _ in range(nc)]
    bufs = [mk(b) for _ in range(nc)]
    streams = [torch.cuda.Stream(dev, priority=-(i % 2)) for i in range(nc)]
    def step(i):
        for e, (p, c, m, cv), s in zip(engs, bufs, streams):
            e.infer_batch_device(p.data_ptr(), c.data_ptr(), PIX_U8, None, b, i * B, m.data_ptr(), cv.data_ptr(), None, s.cuda_stream)
    for i in range(5): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    for e in engs: e.close()
    return dt
for nc in (1, NC, 1, NC):
    dt = run(nc)
    print(f"{nc} context(s) x {B // nc} pairs: {dt * 1e3:.3f} ms per {B} pairs = {B / dt:.0f} pairs/s")
