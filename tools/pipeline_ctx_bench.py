#!/usr/bin/env python3
"""Independent steps of a mid-size batch issued round-robin on NC contexts / HIP streams (one context per stream, no shared buffers): does the launch chain of
one step run under the kernels of the other?  (tools/dual_ctx_bench.py split ONE batch instead.)
   python tools/pipeline_ctx_bench.py [batch=64] [variant=prior3] [mc=16]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cuahn_vio_amd import synth, weights
from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
variant = sys.argv[2] if len(sys.argv) > 2 else "prior3"
mc = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda:0")
blob = weights.pack_state_dict(weights.synthetic_state(0))
ph, ch, prh, _ = synth.make_batch(1000, 32)
def mk():
    reps = (b + 31) // 32
    return (torch.from_numpy(np.tile(ph, (reps, 1, 1))[:b]).to(dev), torch.from_numpy(np.tile(ch, (reps, 1, 1))[:b]).to(dev),
            torch.from_numpy(np.tile(prh, (reps, 1))[:b]).to(dev), torch.zeros(b, 8, device=dev), torch.zeros(b, 64, device=dev))
def run(nc, steps=60, prio=True):
    engs = [HnetEngine(blob, variant=variant, mc_samples=mc, dropout_p=0.05, mc_seed=1, max_batch=b) for _ in range(nc)]
    bufs = [mk() for _ in range(nc)]
    streams = [torch.cuda.Stream(dev, priority=(-(i % 2) if prio else 0)) for i in range(nc)]
    def step(i):
        k = i % nc
        p, c, pr, m, cv = bufs[k]
        engs[k].infer_batch_device(p.data_ptr(), c.data_ptr(), PIX_U8, pr.data_ptr() if variant != "full" else None, b, i * b, m.data_ptr(), cv.data_ptr(), None,
                                   streams[k].cuda_stream)
    for i in range(2 * nc + 4): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    for e in engs: e.close()
    return dt
for nc, prio in ((1, False), (2, False), (2, True), (3, False), (1, False), (2, False), (4, False)):
    dt = run(nc, prio=prio)
    print(f"{nc} context(s){' (prio)' if prio else ''}, steps of {b} pairs ({variant}, N={mc}): {dt * 1e3:.3f} ms per step = {b / dt:.0f} pairs/s", flush=True)
