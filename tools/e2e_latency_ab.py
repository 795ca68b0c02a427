"""End-to-end latency of one frame through the reference's class surface (load_current_img + network_inference, host wall clock; HomographyNet.cpp:178-188 is the
reference's own timer) with hnet_infer's graph in its two forms: kernels that read / write the pinned host block directly (default, round 6) against memcpy / memset
nodes (HNET_GRAPH_COPIES=1 = include/hnet.h HNET_VARIANT_GRAPH_COPIES).  Interleaved repetitions in one process.
    python tools/e2e_latency_ab.py [reps]"""
import contextlib
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuahn_vio_amd import synth, weights  # noqa: E402
from cuahn_vio_amd.homography_net import HomographyNet  # noqa: E402


def run(blob, name, copies, use_prior, n_mc, frames, prior):
    os.environ["HNET_GRAPH_COPIES"] = copies
    with contextlib.redirect_stdout(io.StringIO()):
        net = HomographyNet(name, use_prior=use_prior, blocks_to_run=3, mc_samples=n_mc, dropout_p=0.05, mc_seed=1, weights_blob=blob)
        e2e, dev = [], []
        for i in range(420):
            t0 = time.perf_counter()
            net.load_current_img(frames[i % len(frames)], float(i))
            net.network_inference(prior, 0)
            if i >= 20:
                e2e.append(1e3 * (time.perf_counter() - t0))
                dev.append(net._eng.last_timing()["device_ms"])
        net.close()
    return float(np.percentile(dev, 50)), float(np.percentile(e2e, 50)), float(np.percentile(e2e, 95))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    blob = weights.pack_state_dict(weights.synthetic_state(0))
    frames = [synth.make_pair(100 + i)[0] for i in range(8)]
    prior = np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.25, 2.0, -0.75])
    for label, name, use_prior, n_mc in (("full model, N = 32 (BASELINE config 2)", "ab.hnw", False, 32),
                                         ("prior-3, N = 16, error map copied back (the reference's launch default)", "ab_showError.hnw", True, 16)):
        print(label)
        for r in range(reps):
            for copies in ("0", "1"):
                d, e, e95 = run(blob, name, copies, use_prior, n_mc, frames, prior)
                print("   %-28s events around the graph p50 %.4f ms   end to end p50 %.4f  p95 %.4f ms" % ("memcpy nodes" if copies == "1" else "kernels on the pinned block", d, e, e95))


if __name__ == "__main__":
    main()
