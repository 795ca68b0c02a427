#!/usr/bin/env python3
"""Long soak of the one-XCD tail chains (csrc/chain_lat.h): tens of thousands of small-batch forwards, (a) back to back on one context with alternating batch
sizes, (b) on the four contexts of an hnet_group at once - chain launches of different contexts compete for the same CUs and XCDs -, every output compared bit
for bit with the first forward of its shape, the flag word (bit 1 = a bounded spin gave up) read at the end.   python tools/soak_chain.py [forwards=40000]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cuahn_vio_amd import synth, weights  # noqa: E402
from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine, HnetGroup  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
dev = torch.device("cuda:0")
blob = weights.pack_state_dict(weights.synthetic_state(0))
bad = 0

# (a) one context, batch sizes 1 / 8 / 3 / 5 in turn
ph, ch, prh, _ = synth.make_batch(900, 8)
prev, curr, prior = torch.from_numpy(ph).to(dev), torch.from_numpy(ch).to(dev), torch.from_numpy(prh).to(dev)
for variant, n_mc in (("full", 32), ("prior3", 16)):
    e = HnetEngine(blob, variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=3, max_batch=8)
    sizes = (1, 8, 3, 5)
    CH = 2000
    ref = {}
    t0 = time.time()
    done = 0
    while done < n:
        out = torch.zeros(CH, 8, 72, device=dev)
        for i in range(CH):
            b = sizes[i % 4]
            e.infer_batch_packed_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, prior.data_ptr() if variant != "full" else None, b, 5, out[i].data_ptr())
        e.synchronize()
        o = out.cpu().numpy()
        for i in range(CH):
            b = sizes[i % 4]
            if b not in ref:
                ref[b] = o[i, :b].copy()
            elif not np.array_equal(o[i, :b], ref[b]):
                bad += 1
        done += CH
    flag = e.overflow_flag()
    print(f"soak one context {variant} N={n_mc}: {done} forwards (batches 1 / 8 / 3 / 5 in turn), {bad} differ, flag word {flag}, {time.time() - t0:.1f} s", flush=True)
    bad += 1 if flag else 0
    e.close()

# (b) four contexts at once, batch 2 and batch 8
for batch in (2, 8):
    g = HnetGroup(blob, 4, variant="full", mc_samples=16, dropout_p=0.05, mc_seed=3, max_batch=8)
    CH = 2000
    ref = None
    done = 0
    nb = 0
    t0 = time.time()
    while done < n // 2:
        out = torch.zeros(CH, 8, 72, device=dev)
        torch.cuda.synchronize()
        for i in range(CH):
            g.infer_batch_packed_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, None, batch, 5, out[i].data_ptr())
        g.synchronize()
        o = out.cpu().numpy()
        if ref is None:
            ref = o[0, :batch].copy()
        nb += sum(1 for i in range(CH) if not np.array_equal(o[i, :batch], ref))
        done += CH
    flag = g.overflow_flag()
    print(f"soak four contexts at once, full N=16 batch={batch}: {done} forwards, {nb} differ, flag word {flag}, {time.time() - t0:.1f} s", flush=True)
    bad += nb + (1 if flag else 0)
    g.close()
sys.exit(1 if bad else 0)
