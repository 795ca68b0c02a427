"""worker of tests/test_gpu_dist.py: N ranks sharing ONE GPU (gloo, host-staged gathers) run the sharded forward;
rank 0 compares with the unsharded forward of a single context and writes the verdict as JSON to argv[1]."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cuahn_vio_amd import dist as hdist  # noqa: E402
from cuahn_vio_amd import synth, weights  # noqa: E402
from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    blob = weights.pack_state_dict(weights.synthetic_state(0))
    B, n_mc = 8, 16      # divisible by the world sizes tested (equal shards: one all_gather_into_tensor)
    prev_h, curr_h, prior_h, _ = synth.make_batch(300, B)
    res = {}

    # ---- pairs sharded over the ranks (SURVEY.md §8e "partitioning (batched)")
    lo, hi = hdist.shard_range(B, world, rank)
    nb = hi - lo
    eng = HnetEngine(blob, variant="prior3", mc_samples=n_mc, dropout_p=0.05, mc_seed=5, max_batch=B, precision=2)
    prev, curr = torch.from_numpy(prev_h[lo:hi]).to(dev), torch.from_numpy(curr_h[lo:hi]).to(dev)
    prior = torch.from_numpy(prior_h[lo:hi]).to(dev)
    mean, cov = torch.zeros(nb, 8, device=dev), torch.zeros(nb, 64, device=dev)
    eng.infer_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, prior.data_ptr(), nb, 40 + lo, mean.data_ptr(), cov.data_ptr(), None,
                           torch.cuda.current_stream(dev))
    # no synchronize: the forward runs on torch's current stream, the gather below is ordered after it by the stream
    out, gathered = torch.zeros(nb, 72, device=dev), torch.zeros(world * nb, 72, device=dev)
    hdist.gather_outputs(mean, cov, out, gathered)
    if rank == 0:
        m_ref, c_ref = eng.infer_batch(prev_h, curr_h, prior_h, pair_seq0=40)[:2]
        g = gathered.cpu().numpy()
        res["pairs_mean_maxdiff"] = float(np.abs(g[:, :8] - m_ref).max())
        res["pairs_cov_reldiff"] = float(np.abs(g[:, 8:] - c_ref.reshape(B, 64)).max() / np.abs(c_ref).max())

    # ---- MC-dropout samples sharded over the ranks (BASELINE config 4)
    shard = hdist.shard_range(n_mc, world, rank)
    e2 = HnetEngine(blob, variant="full", mc_samples=n_mc, dropout_p=0.05, mc_seed=5, max_batch=B, precision=2, mc_shard=shard)
    n_loc = shard[1] - shard[0]
    pv, cv = torch.from_numpy(prev_h).to(dev), torch.from_numpy(curr_h).to(dev)
    ms, lv, h1 = torch.zeros(B, n_loc, 8, device=dev), torch.zeros(B, n_loc, 8, device=dev), torch.zeros(B, 9, device=dev)
    sp = torch.cuda.current_stream(dev)
    e2.infer_mc_partial_device(pv.data_ptr(), cv.data_ptr(), PIX_U8, None, B, 77, ms.data_ptr(), lv.data_ptr(), h1.data_ptr(), sp)
    ms_all, lv_all, _ = hdist.gather_mc_samples(ms, lv, h1)
    mean2, cov2 = torch.zeros(B, 8, device=dev), torch.zeros(B, 64, device=dev)
    e2.mc_finish_device(ms_all.data_ptr(), lv_all.data_ptr(), n_mc, h1.data_ptr(), B, mean2.data_ptr(), cov2.data_ptr(), sp)
    torch.cuda.synchronize(dev)
    if rank == 0:
        e1 = HnetEngine(blob, variant="full", mc_samples=n_mc, dropout_p=0.05, mc_seed=5, max_batch=B, precision=2)
        m1, c1 = e1.infer_batch(prev_h, curr_h, None, pair_seq0=77)[:2]
        res["mc_mean_maxdiff"] = float(np.abs(mean2.cpu().numpy() - m1).max())
        res["mc_cov_reldiff"] = float(np.abs(cov2.cpu().numpy().reshape(B, 8, 8) - c1).max() / np.abs(c1).max())
        if os.environ.get("HNET_DIST_DEBUG"):
            print("cov2[0] diag", np.diag(cov2.cpu().numpy().reshape(B, 8, 8)[0]), "\nc1[0] diag", np.diag(c1[0]), "\nlv_all[0,:,0]", lv_all[0, :, 0].cpu().numpy(),
                  "\nms_all[0,:,0]", ms_all[0, :, 0].cpu().numpy(), flush=True)
        with open(sys.argv[1], "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
