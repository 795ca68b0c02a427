#!/usr/bin/env python3
"""Times the fused block-4 kernel with phases dropped (the -DHNET_B4_ABLATE build: `make -C cuahn_vio_amd/csrc ablate`), one
process per switch value because the launcher reads HNET_B4_DBG once.  Results of the ablated runs are WRONG by design; only
the stage time matters.   dbg bits: 1 = no phase-1 stores, 2 = no phase-2 MFMAs, 4 = no phase-1 MFMAs."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child():
    import numpy as np, torch
    from cuahn_vio_amd import synth, weights
    from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine
    dev = torch.device("cuda:0")
    blob = weights.pack_state_dict(weights.synthetic_state(0))
    B = 256
    ph, ch, _pr, _ = synth.make_batch(1000, 16)
    prev = torch.from_numpy(np.tile(ph, (16, 1, 1))).to(dev)
    curr = torch.from_numpy(np.tile(ch, (16, 1, 1))).to(dev)
    mean, cov = torch.zeros(B, 8, device=dev), torch.zeros(B, 64, device=dev)
    out = {}
    for cfg in (6,):     # (the shipped kernel: 7 x 32 tiles, 256 threads, LDS-DMA staging, fragment reuse)
        e = HnetEngine(blob, variant="full", mc_samples=32, dropout_p=0.05, mc_seed=1, max_batch=B, precision=2)
        names = [n for n, _ in e.stages()]
        k = names.index("block_4_0+4_1")
        e.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, None, B, 0, mean.data_ptr(), cov.data_ptr(), 2)
        ms = [e.profile_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, None, B, 0, mean.data_ptr(), cov.data_ptr(), 3)[k] for _ in range(4)]
        out[f"cfg{cfg}"] = round(float(np.median(ms)), 4)
        e.close()
    print(json.dumps({"dbg": int(os.environ.get("HNET_B4_DBG", "0")), **out}), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        lib = os.path.join(ROOT, "cuahn_vio_amd", "libhnet_hip_ablate.so")
        for dbg in (0, 1, 2, 4, 6, 7, 3, 5):
            env = dict(os.environ, HNET_LIB_PATH=lib, HNET_B4_DBG=str(dbg))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
            print(p.stdout.strip() or p.stderr[-600:], flush=True)
