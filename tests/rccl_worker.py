"""worker of tests/test_gpu_rccl.py: the collective path of cuahn_vio_amd/dist.py on a REAL RCCL communicator.  One MI355X can host a
1-rank nccl process group, which is enough to execute init_process_group("nccl"), all_gather_into_tensor and the barrier / all_reduce
bench.py issues; with WORLD_SIZE > 1 (a multi-GPU node) the same file checks the gathered order across ranks.  Writes the verdict as
JSON to argv[1] (rank 0)."""
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
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    res = {"backend": dist.get_backend(), "world": dist.get_world_size(), "rccl_version": list(torch.cuda.nccl.version())}
    blob = weights.pack_state_dict(weights.synthetic_state(0))
    B, n_mc = 8 * world, 16
    prev_h, curr_h, prior_h, _ = synth.make_batch(300, B)
    lo, hi = hdist.shard_range(B, world, rank)
    nb = hi - lo
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)

    # ---- pairs sharded over the ranks: forward, pack [nb, 72], one all_gather_into_tensor
    eng = HnetEngine(blob, variant="prior3", mc_samples=n_mc, dropout_p=0.05, mc_seed=5, max_batch=nb, device_id=local)
    prev, curr = torch.from_numpy(prev_h[lo:hi]).to(dev), torch.from_numpy(curr_h[lo:hi]).to(dev)
    prior = torch.from_numpy(prior_h[lo:hi]).to(dev)
    mean, cov = torch.zeros(nb, 8, device=dev), torch.zeros(nb, 64, device=dev)
    eng.infer_batch_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, prior.data_ptr(), nb, 40 + lo, mean.data_ptr(), cov.data_ptr(), None, stream)
    out, gathered = torch.zeros(nb, 72, device=dev), torch.zeros(world * nb, 72, device=dev)
    hdist.gather_outputs(mean, cov, out, gathered)          # the nccl arm: dist.all_gather_into_tensor
    torch.cuda.synchronize(dev)
    g = gathered.cpu().numpy()
    res["own_shard_bitwise"] = bool(np.array_equal(g[lo:hi, :8], mean.cpu().numpy()) and np.array_equal(g[lo:hi, 8:], cov.cpu().numpy()))
    # every rank's shard against an unsharded forward of the whole batch on this rank (slot invariance at equal batch size is bitwise only
    # for equal batch sizes, so: a second context run shard by shard)
    ok_all = True
    for r in range(world):
        a, b = hdist.shard_range(B, world, r)
        pm, pc = torch.zeros(b - a, 8, device=dev), torch.zeros(b - a, 64, device=dev)
        tp, tc, tr = torch.from_numpy(prev_h[a:b]).to(dev), torch.from_numpy(curr_h[a:b]).to(dev), torch.from_numpy(prior_h[a:b]).to(dev)   # kept alive until the sync
        eng.infer_batch_device(tp.data_ptr(), tc.data_ptr(), PIX_U8, tr.data_ptr(), b - a, 40 + a, pm.data_ptr(), pc.data_ptr(), None, stream)
        torch.cuda.synchronize(dev)
        ok_all = ok_all and np.array_equal(g[a:b, :8], pm.cpu().numpy()) and np.array_equal(g[a:b, 8:], pc.cpu().numpy())
    res["all_shards_bitwise"] = bool(ok_all)
    eng.close()

    # ---- MC-dropout samples sharded over the ranks: gather_mc_samples (nccl arm), finish on the gathered samples == unsharded context
    s_lo, s_hi = hdist.shard_range(n_mc, world, rank)
    e_sh = HnetEngine(blob, variant="full", mc_samples=n_mc, dropout_p=0.05, mc_seed=5, max_batch=2, device_id=local, mc_shard=(s_lo, s_hi))
    p2, c2 = torch.from_numpy(prev_h[:2]).to(dev), torch.from_numpy(curr_h[:2]).to(dev)
    n_loc = s_hi - s_lo
    ms, lv, h1 = torch.zeros(2, n_loc, 8, device=dev), torch.zeros(2, n_loc, 8, device=dev), torch.zeros(2, 9, device=dev)
    e_sh.infer_mc_partial_device(p2.data_ptr(), c2.data_ptr(), PIX_U8, None, 2, 7, ms.data_ptr(), lv.data_ptr(), h1.data_ptr(), stream)
    ms_all, lv_all, _ = hdist.gather_mc_samples(ms, lv, h1)
    m_sh, c_sh = torch.zeros(2, 8, device=dev), torch.zeros(2, 64, device=dev)
    e_sh.mc_finish_device(ms_all.data_ptr(), lv_all.data_ptr(), n_mc, h1.data_ptr(), 2, m_sh.data_ptr(), c_sh.data_ptr(), stream)
    torch.cuda.synchronize(dev)
    e_sh.close()
    e_un = HnetEngine(blob, variant="full", mc_samples=n_mc, dropout_p=0.05, mc_seed=5, max_batch=2, device_id=local)
    m_un, c_un = torch.zeros(2, 8, device=dev), torch.zeros(2, 64, device=dev)
    e_un.infer_batch_device(p2.data_ptr(), c2.data_ptr(), PIX_U8, None, 2, 7, m_un.data_ptr(), c_un.data_ptr(), None, stream)
    torch.cuda.synchronize(dev)
    e_un.close()
    res["mc_sharded_equals_unsharded_bitwise"] = bool(torch.equal(m_sh, m_un) and torch.equal(c_sh, c_un))

    # ---- the two other collectives bench.py issues: barrier and the max-over-ranks all_reduce of the timer
    dist.barrier()
    t = torch.tensor([float(rank + 1)], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    res["all_reduce_max"] = float(t.item())
    if rank == 0:
        with open(sys.argv[1], "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
