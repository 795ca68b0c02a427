#!/usr/bin/env python3
"""The same small-batch forward many times on one context (and through the graph-replayed hnet_infer of the class surface): every repetition must reproduce
the first one bit for bit.  Guards the fence-free split-K last arriver (igemm_s3.h s3_splitk_last_arriver: system-scope loads / stores + a relaxed
agent-scope ticket) and the keep bits drawn beside block 4's warp against a rare ordering bug.   python tools/determinism_stress.py [reps=3000]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cuahn_vio_amd import synth, weights
from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda:0")
blob = weights.pack_state_dict(weights.synthetic_state(0))
bad = 0
for variant, n_mc, batch in (("full", 32, 1), ("prior3", 16, 1), ("full", 16, 2), ("prior1", 8, 8)):
    ph, ch, prh, _ = synth.make_batch(40 + batch, batch)
    prev, curr, prior = torch.from_numpy(ph).to(dev), torch.from_numpy(ch).to(dev), torch.from_numpy(prh).to(dev)
    out = torch.zeros(reps, batch, 72, device=dev)
    e = HnetEngine(blob, variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=3, max_batch=batch)
    for i in range(reps):
        e.infer_batch_packed_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, prior.data_ptr() if variant != "full" else None, batch, 5, out[i].data_ptr())
    e.synchronize()
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    diff = [i for i in range(1, reps) if not np.array_equal(o[i], o[0])]
    bad += len(diff)
    print(f"determinism {variant} N={n_mc} batch={batch}: {reps} forwards, {len(diff)} differ from the first" + (f" (first at {diff[0]}, max |d| {np.abs(o[diff[0]] - o[0]).max():.3e})" if diff else ""), flush=True)
    e.close()
# the bench shape on two contexts / streams at once (the default of bench.py): the steps of one context run under the other's kernels and must not notice
big = int(os.environ.get("HNET_STRESS_BIG", "200"))
if big:
    batch, n_mc = 256, 32
    ph, ch, prh, _ = synth.make_batch(77, 32)
    tile = lambda a: torch.from_numpy(np.tile(a, (8,) + (1,) * (a.ndim - 1))).to(dev)
    prev, curr = tile(ph), tile(ch)
    engs = [HnetEngine(blob, variant="full", mc_samples=n_mc, dropout_p=0.05, mc_seed=3, max_batch=batch) for _ in range(2)]
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    outs = [torch.zeros(big, batch, 72, device=dev) for _ in range(2)]
    for i in range(big):
        for k in range(2):
            engs[k].infer_batch_packed_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, None, batch, 5, outs[k][i].data_ptr(), None, streams[k].cuda_stream)
    torch.cuda.synchronize()
    o = [x.cpu().numpy() for x in outs]
    diff = [(k, i) for k in range(2) for i in range(big) if not np.array_equal(o[k][i], o[0][0])]
    bad += len(diff)
    print(f"determinism full N={n_mc} batch={batch}, two contexts interleaved: {2 * big} forwards, {len(diff)} differ from the first", flush=True)
    for e in engs: e.close()
sys.exit(1 if bad else 0)
